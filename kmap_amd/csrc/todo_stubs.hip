// TEMPORARY: entry points declared in kmap_hip.h that are not implemented yet return KMAP_E_UNSUP.
#include "common.h"
#define STUB(name, ...) int name(__VA_ARGS__) { kmap_set_error(#name ": not implemented yet"); return KMAP_E_UNSUP; }
extern "C" {
STUB(kmap_scan_create, kmap_scan **)
STUB(kmap_scan_destroy, kmap_scan *)
STUB(kmap_scan_run_dev, kmap_scan *, const uint8_t *, int64_t, const int64_t *, int64_t, int, uint64_t, int, int, int64_t *, void *)
STUB(kmap_scan_fetch, kmap_scan *, int32_t *, int8_t *, int32_t *)
}
