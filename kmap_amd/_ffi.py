"""ctypes binding of libkmap_hip.so (C ABI declared in include/kmap_hip.h).

The HIP library is the ONLY compute backend of this package: if it cannot be loaded the
import of any operator raises (there is no CPU fallback).  `kmap_amd.build.build()` compiles it.
"""
import ctypes as C
import os
from pathlib import Path

import numpy as np

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "libkmap_hip.so"
_lib = None

vp, i64, i32, u32, u64, f32, f64 = C.c_void_p, C.c_int64, C.c_int, C.c_uint32, C.c_uint64, C.c_float, C.c_double
sz = C.c_size_t
P = C.POINTER

_SIGS = {
    "kmap_version": (i32, []),
    "kmap_last_error": (C.c_char_p, []),
    "kmap_device_count": (i32, [P(i32)]),
    "kmap_set_device": (i32, [i32]),
    "kmap_get_device": (i32, [P(i32)]),
    "kmap_device_arch": (i32, [C.c_char_p, i32]),
    "kmap_malloc": (i32, [P(vp), sz]),
    "kmap_free": (i32, [vp]),
    "kmap_memset": (i32, [vp, i32, sz, vp]),
    "kmap_memcpy_h2d": (i32, [vp, vp, sz, vp]),
    "kmap_memcpy_d2h": (i32, [vp, vp, sz, vp]),
    "kmap_memcpy_d2d": (i32, [vp, vp, sz, vp]),
    "kmap_memcpy2d_d2h": (i32, [vp, sz, vp, sz, sz, sz, vp]),
    "kmap_scratch_release": (i32, [C.c_size_t]),
    "kmap_stream_sync": (i32, [vp]),
    "kmap_stream_create": (i32, [P(vp)]),
    "kmap_stream_destroy": (i32, [vp]),
    "kmap_event_create": (i32, [P(vp)]),
    "kmap_event_destroy": (i32, [vp]),
    "kmap_event_record": (i32, [vp, vp]),
    "kmap_event_elapsed_ms": (i32, [vp, vp, P(f32)]),
    "kmap_hash_kmers_u32_dev": (i32, [vp, i64, i32, vp, vp]),
    "kmap_hash_kmers_u64_dev": (i32, [vp, i64, i32, vp, vp]),
    "kmap_hash_kmers_u32": (i32, [vp, i64, i32, vp]),
    "kmap_hash_kmers_u64": (i32, [vp, i64, i32, vp]),
    "kmap_dedupe_per_read_u32_dev": (i32, [vp, i64, vp, i64, vp]),
    "kmap_dedupe_per_read_u64_dev": (i32, [vp, i64, vp, i64, vp]),
    "kmap_dedupe_per_read_u32": (i32, [vp, i64, vp, i64]),
    "kmap_dedupe_per_read_u64": (i32, [vp, i64, vp, i64]),
    "kmap_revcom_u32_dev": (i32, [vp, i64, i32, vp, vp]),
    "kmap_revcom_u64_dev": (i32, [vp, i64, i32, vp, vp]),
    "kmap_revcom_u32": (i32, [vp, i64, i32, vp]),
    "kmap_revcom_u64": (i32, [vp, i64, i32, vp]),
    "kmap_hamdist_1vN_u32_dev": (i32, [vp, i64, u32, i32, i32, vp, vp]),
    "kmap_hamdist_1vN_u64_dev": (i32, [vp, i64, u64, i32, i32, vp, vp]),
    "kmap_hamdist_1vN_u32": (i32, [vp, i64, u32, i32, i32, vp]),
    "kmap_hamdist_1vN_u64": (i32, [vp, i64, u64, i32, i32, vp]),
    "kmap_mask_hamball_dev": (i32, [vp, i64, i32, vp, vp, i32, vp]),
    "kmap_mask_hamball": (i32, [vp, i64, i32, vp, vp, i32]),
    "kmap_counts_create": (i32, [P(vp)]),
    "kmap_counts_destroy": (i32, [vp]),
    "kmap_counts_run_seq_dev": (i32, [vp, vp, i64, vp, i64, i32, i32, i32, P(i64), vp]),
    "kmap_counts_run_hashes_dev": (i32, [vp, vp, i64, i32, i32, P(i64), vp]),
    "kmap_counts_load": (i32, [vp, vp, vp, i64, i32]),
    "kmap_counts_fetch": (i32, [vp, vp, vp]),
    "kmap_counts_fetch_stream": (i32, [vp, vp, vp, vp]),
    "kmap_counts_table_dev": (i32, [vp, P(vp), P(vp), P(i64)]),
    "kmap_counts_fetch_range": (i32, [vp, i32, i64, i64, vp, vp]),
    "kmap_counts_write_range": (i32, [vp, i32, i64, i64, i32, i64, vp]),
    "kmap_counts_total": (i32, [vp, P(i64)]),
    "kmap_hamball_extract": (i32, [vp, vp, i64, i32, u64, i32, i32, vp, vp, P(i64), vp]),
    "kmap_counts_hamball_extract": (i32, [vp, u64, i32, i32, i64, vp, vp, P(i64), vp]),
    "kmap_pos_density": (i32, [vp, vp, vp, vp, i64, i32, vp, i32, f64, vp]),
    "kmap_hamdist_pitch": (i64, [i64]),
    "kmap_knn_sums_kmers_u32_dev": (i32, [vp, vp, i64, i32, vp, i32, vp, i32, i64, i64, vp, i64, vp]),
    "kmap_knn_sums_kmers_u64_dev": (i32, [vp, vp, i64, i32, vp, i32, vp, i32, i64, i64, vp, i64, vp]),
    "kmap_label_kmers_dev": (i32, [vp, i64, i32, i32, vp, vp, vp, i32, i32, vp, vp]),
    "kmap_label_sums_dev": (i32, [vp, vp, i32, i64, i32, vp, vp]),
    "kmap_label_prefix_dev": (i32, [vp, vp, i32, i64, i32, vp, vp, vp]),
    "kmap_prefix_search_dev": (i32, [vp, i64, vp, i64, vp]),
    "kmap_label_members_dev": (i32, [vp, vp, i64, i32, i64, vp]),
    "kmap_gather_dev": (i32, [vp, i32, vp, i64, vp]),
    "kmap_counts_topk": (i32, [vp, i32, vp, vp, vp, P(i32)]),
    "kmap_counts_hamball_mass": (i32, [vp, vp, i32, i32, i32, vp]),
    "kmap_packed_groups": (i64, [i64]),
    "kmap_pack_reads_dev": (i32, [vp, i64, vp, vp, vp]),
    "kmap_unpack_reads_dev": (i32, [vp, vp, i64, vp, vp]),
    "kmap_hash_kmers_packed_dev": (i32, [vp, vp, i64, i32, vp, vp]),
    "kmap_counts_run_packed_dev": (i32, [vp, vp, vp, i64, vp, i64, i32, i32, i32, P(i64), vp]),
    "kmap_counts_run_packed_range_dev": (i32, [vp, vp, vp, i64, vp, i64, i32, i32, i32, u64, u64, P(i64), vp]),
    "kmap_counts_hist_packed_dev": (i32, [vp, vp, vp, i64, vp, i64, i32, i32, vp]),
    "kmap_counts_bins": (i32, [vp, P(vp), P(i64)]),
    "kmap_counts_finish": (i32, [vp, i32, i32, P(i64), vp]),
    "kmap_counts_presence_dev": (i32, [vp, i32, vp, vp]),
    "kmap_counts_merge_presence_dev": (i32, [vp, i32, vp, vp]),
    "kmap_counts_finish_range": (i32, [vp, i32, i32, u64, u64, P(i64), vp]),
    "kmap_counts_adopt_dev": (i32, [vp, vp, vp, i64, i32]),
    "kmap_mask_hamball_packed_dev": (i32, [vp, vp, i64, i32, vp, vp, i32, vp, vp]),
    "kmap_inval_set_prefix_dev": (i32, [vp, i64, vp]),
    "kmap_pack_planes_dev": (i32, [vp, i64, vp, vp]),
    "kmap_scan_run_packed_dev": (i32, [vp, vp, vp, i64, vp, i64, i32, u64, i32, i32, P(i64), vp, vp]),
    "kmap_scan_create": (i32, [P(vp)]),
    "kmap_scan_declare_uniform": (i32, [vp, vp, i64, i64, i64, P(i32), vp]),
    "kmap_scan_destroy": (i32, [vp]),
    "kmap_scan_run_dev": (i32, [vp, vp, i64, vp, i64, i32, u64, i32, i32, P(i64), vp]),
    "kmap_scan_fetch": (i32, [vp, vp, vp, vp]),
    "kmap_scan_fetch_stream": (i32, [vp, vp, vp, vp]),
    "kmap_scan_fetch_stream_u8": (i32, [vp, vp, vp, vp]),
    "kmap_scan_result_dev": (i32, [vp, P(vp), P(vp), P(i64), P(i64)]),
    "kmap_scan_summary": (i32, [vp, P(i64), P(i32), vp]),
    "kmap_write_occurrence_csv": (i32, [C.c_char_p, C.c_char_p, i64, i32, vp, vp, vp, P(i64)]),
    "kmap_write_occurrence_csv_u8": (i32, [C.c_char_p, C.c_char_p, i64, i32, vp, vp, vp, P(i64)]),
    "kmap_write_f2_tsv_line": (i32, [i32, vp, i64]),
    "kmap_cell_medians_i32": (i32, [vp, vp, i64, i64, vp]),
    "kmap_fasta_open": (i32, [C.c_char_p, P(vp), P(i64), P(i64)]),
    "kmap_fasta_read": (i32, [vp, vp, vp]),
    "kmap_fasta_close": (i32, [vp]),
    "kmap_synth_reads_dev": (i32, [vp, vp, i64, i32, C.c_uint64, vp, vp, vp, i32, C.c_double, vp]),
    "kmap_hamdist_matrix_u32_dev": (i32, [vp, vp, i64, i32, vp, i32, i64, i64, vp, i64, vp]),
    "kmap_hamdist_matrix_u64_dev": (i32, [vp, vp, i64, i32, vp, i32, i64, i64, vp, i64, vp]),
    "kmap_hamdist_matrix_u8": (i32, [vp, vp, i64, i32, vp, i32, vp]),
    "kmap_knn_sums_u8_dev": (i32, [vp, i64, vp, i64, i32, i64, i64, vp, i64, vp]),
    "kmap_knn_select_u8_dev": (i32, [vp, i64, i64, i32, i64, i64, vp, vp]),
    "kmap_rows_fresh_u8_dev": (i32, [vp, i64, i64, i64, i64, vp, vp]),
    "kmap_gather_rows_u8_dev": (i32, [vp, i64, i64, vp, i64, vp, i64, vp]),
    "kmap_knn_smooth_f32": (i32, [vp, vp, i64, i32, vp]),
    "kmap_ld_prob_mat_f32": (i32, [vp, i64, vp]),
    "kmap_cross_entropy_f32": (i32, [vp, vp, i64, vp]),
    "kmap_gradient_loss_f32": (i32, [vp, vp, vp, i64, vp]),
    "kmap_embed_create": (i32, [P(vp), i64, i64, i64, i32, f32, i32]),
    "kmap_embed_create_cyclic": (i32, [P(vp), i64, i32, i32, i32, f32]),
    "kmap_embed_cyclic_blocks": (i64, [i64, i32, i32]),
    "kmap_embed_destroy": (i32, [vp]),
    "kmap_embed_set_prob_f32": (i32, [vp, vp, i64]),
    "kmap_embed_set_prob_lut": (i32, [vp, vp, i64, vp, i32]),
    "kmap_embed_set_row_map": (i32, [vp, vp, i64]),
    "kmap_embed_set_coords": (i32, [vp, vp, vp]),
    "kmap_embed_set_jitter": (i32, [vp, vp, i32]),
    "kmap_embed_forces": (i32, [vp, vp, vp, vp]),
    "kmap_embed_apply": (i32, [vp, vp, vp, vp]),
    "kmap_embed_msg_floats": (i64, [i64]),
    "kmap_embed_forces_msg": (i32, [vp, vp, vp]),
    "kmap_embed_apply_msg": (i32, [vp, vp, vp]),
    "kmap_peer_create": (i32, [C.POINTER(vp), i32, i32, i64]),
    "kmap_peer_handle": (i32, [vp, vp]),
    "kmap_peer_connect": (i32, [vp, vp]),
    "kmap_peer_bus_id": (i32, [vp]),
    "kmap_peer_can_access": (i32, [vp, C.POINTER(i32)]),
    "kmap_peer_hello_push": (i32, [vp, u64]),
    "kmap_peer_hello_check": (i32, [vp, u64, C.POINTER(i32)]),
    "kmap_peer_set_timeout_ms": (i32, [vp, i64]),
    "kmap_peer_status": (i32, [vp, C.POINTER(i32), C.POINTER(i64)]),
    "kmap_peer_destroy": (i32, [vp]),
    "kmap_embed_step_peer": (i32, [vp, vp, i32, vp]),
    "kmap_embed_step": (i32, [vp, i32, vp]),
    "kmap_embed_state": (i32, [vp, P(i64), P(i32), P(f32), P(f32), P(i32), vp]),
    "kmap_embed_get_coords": (i32, [vp, vp, vp]),
    "kmap_embed_get_best": (i32, [vp, vp, vp]),
    "kmap_embed_get_losses": (i32, [vp, vp, i64, P(i64), vp]),
    "kmap_embed_coords_dev": (vp, [vp]),
    "kmap_selftest_seq_div": (i32, [i32, i32, i32, C.c_uint32, C.c_uint32, P(C.c_uint64), P(C.c_uint32)]),
}


class KmapError(RuntimeError):
    pass


def exported_symbols():
    """Names the header declares (used by the CPU test that checks the .so exports all of them)."""
    return sorted(_SIGS)


def lib():
    """Load libkmap_hip.so; raises if it has not been built (no fallback backend exists)."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise KmapError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(kmap_amd has no CPU fallback)")
        L = C.CDLL(str(LIB_PATH), mode=os.RTLD_GLOBAL if hasattr(os, "RTLD_GLOBAL") else C.DEFAULT_MODE)
        for name, (res, args) in _SIGS.items():
            f = getattr(L, name)   # AttributeError here = header and library out of sync
            f.restype, f.argtypes = res, args
        _lib = L
    return _lib


def last_error():
    """message of the calling thread's last failed library call"""
    return lib().kmap_last_error().decode(errors="replace")


def check(rc):
    if rc == 0:
        return
    msg = lib().kmap_last_error().decode(errors="replace")
    if rc == -1:
        raise ValueError(msg)
    if rc == -3:
        raise MemoryError(msg)
    if rc == -6:
        raise OSError(msg)
    raise KmapError(f"libkmap_hip error {rc}: {msg}")


def ptr(a):
    """void* of a C-contiguous numpy array (or None)."""
    if a is None:
        return None
    assert a.flags.c_contiguous
    return a.ctypes.data_as(vp)


def device_count():
    n = i32(0)
    check(lib().kmap_device_count(C.byref(n)))
    return n.value


def device_arch():
    buf = C.create_string_buffer(128)
    check(lib().kmap_device_arch(buf, 128))
    return buf.value.decode()


class DeviceBuffer:
    """A hipMalloc'ed buffer owned by Python (freed on GC)."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        p = vp()
        check(lib().kmap_malloc(C.byref(p), self.nbytes))
        self.ptr = p.value

    @classmethod
    def from_numpy(cls, a, stream=None):
        a = np.ascontiguousarray(a)
        b = cls(a.nbytes)
        check(lib().kmap_memcpy_h2d(b.ptr, ptr(a), a.nbytes, stream))
        return b

    def to_numpy(self, dtype, shape, stream=None, offset=0):
        out = np.empty(shape, dtype)
        check(lib().kmap_memcpy_d2h(ptr(out), self.ptr + offset, out.nbytes, stream))
        return out

    def zero(self, stream=None):
        check(lib().kmap_memset(self.ptr, 0, self.nbytes, stream))

    def free(self):
        if getattr(self, "ptr", None):
            lib().kmap_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceView(DeviceBuffer):
    """Non-owning alias of device memory someone else owns (a torch tensor, a slice of a DeviceBuffer): same `.ptr` /
    `.to_numpy` interface, `free()` only drops the alias.  `keep` holds a reference to the owner."""

    def __init__(self, dev_ptr, nbytes, keep=None):   # noqa: super().__init__ would allocate
        self.ptr, self.nbytes, self._keep = int(dev_ptr), int(nbytes), keep

    def free(self):
        self.ptr, self._keep = None, None


def sync(stream=None):
    check(lib().kmap_stream_sync(stream))


class Event:
    def __init__(self):
        p = vp()
        check(lib().kmap_event_create(C.byref(p)))
        self.ptr = p.value

    def record(self, stream=None):
        check(lib().kmap_event_record(self.ptr, stream))

    def elapsed_ms(self, stop):
        ms = f32(0)
        check(lib().kmap_event_elapsed_ms(self.ptr, stop.ptr, C.byref(ms)))
        return ms.value

    def __del__(self):
        try:
            if self.ptr:
                lib().kmap_event_destroy(self.ptr)
        except Exception:
            pass
