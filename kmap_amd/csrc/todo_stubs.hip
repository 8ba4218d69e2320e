// TEMPORARY: entry points declared in kmap_hip.h that are not implemented yet return KMAP_E_UNSUP.
#include "common.h"
#define STUB(name, ...) int name(__VA_ARGS__) { kmap_set_error(#name ": not implemented yet"); return KMAP_E_UNSUP; }
extern "C" {
STUB(kmap_scan_create, kmap_scan **)
STUB(kmap_scan_destroy, kmap_scan *)
STUB(kmap_scan_run_dev, kmap_scan *, const uint8_t *, int64_t, const int64_t *, int64_t, int, uint64_t, int, int, int64_t *, void *)
STUB(kmap_scan_fetch, kmap_scan *, int32_t *, int8_t *, int32_t *)
STUB(kmap_knn_sums_u8_dev, const uint8_t *, int64_t, const int32_t *, int64_t, int, int64_t, int64_t, uint16_t *, int64_t, void *)
STUB(kmap_knn_smooth_f32, const float *, const int32_t *, int64_t, int, float *)
STUB(kmap_ld_prob_mat_f32, const float *, int64_t, float *)
STUB(kmap_cross_entropy_f32, const float *, const float *, int64_t, float *)
STUB(kmap_gradient_loss_f32, const float *, const float *, const float *, int64_t, float *)
STUB(kmap_embed_create, kmap_embed **, int64_t, int64_t, int64_t, int, float, int)
STUB(kmap_embed_destroy, kmap_embed *)
STUB(kmap_embed_set_prob_f32, kmap_embed *, const float *, int64_t)
STUB(kmap_embed_set_prob_lut, kmap_embed *, const uint16_t *, int64_t, const float *, int)
STUB(kmap_embed_set_coords, kmap_embed *, const float *, const float *)
STUB(kmap_embed_set_jitter, kmap_embed *, const float *, int)
STUB(kmap_embed_forces, kmap_embed *, float *, double *, void *)
STUB(kmap_embed_apply, kmap_embed *, const float *, const double *, void *)
STUB(kmap_embed_step, kmap_embed *, int, void *)
STUB(kmap_embed_state, kmap_embed *, int64_t *, int *, float *, float *, int *, void *)
STUB(kmap_embed_get_coords, kmap_embed *, float *, void *)
STUB(kmap_embed_get_best, kmap_embed *, float *, void *)
STUB(kmap_embed_get_losses, kmap_embed *, float *, int64_t, int64_t *, void *)
void *kmap_embed_coords_dev(kmap_embed *) { return nullptr; }
}
