"""Error behaviour at the boundary: bad arguments come back as status codes + kmap_last_error(), which the Python layer
raises as ValueError (where the reference asserts / raises), never as a crash or a silent wrong answer."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_k_range_errors():
    import kmap_amd.kmer_count as K
    seq = np.zeros(100, np.uint8)
    for k in (0, -1, 32, 40):
        with pytest.raises(Exception):
            K.comp_kmer_hash(seq, k)                      # reference: get_hash_dtype raises (kmer_count.py:365)
    h = np.zeros(10, np.uint32)
    with pytest.raises(Exception):
        K.cal_hamming_dist(h, 0, 32)
    with pytest.raises(AssertionError):
        K.cal_hamming_dist_head(h, 0, 8, 9)               # consensus longer than k (kmer_count.py:531)


def test_status_codes_and_last_error():
    from kmap_amd import _ffi
    L = _ffi.lib()
    buf = _ffi.DeviceBuffer(1024)
    rc = L.kmap_hash_kmers_u32_dev(buf.ptr, 100, 16, buf.ptr, None)      # k = 16 needs the u64 entry point
    assert rc == -1 and b"u64" in L.kmap_last_error()
    rc = L.kmap_hamdist_matrix_u32_dev(buf.ptr, buf.ptr, 10, 8, None, 0, 0, 10, buf.ptr, 5, None)   # ld < n
    assert rc == -1 and b"ld" in L.kmap_last_error()
    rc = L.kmap_hamdist_matrix_u32_dev(buf.ptr, buf.ptr, 10, 8, None, 0, 5, 10, buf.ptr, 16, None)  # row range past n
    assert rc == -1
    clen = (C.c_int * 1)(9)
    rc = L.kmap_hamdist_matrix_u32_dev(buf.ptr, buf.ptr, 10, 8, clen, 1, 0, 10, buf.ptr, 16, None)  # clen > k
    assert rc == -1 and b"clen" in L.kmap_last_error()
    rc = L.kmap_knn_select_u8_dev(buf.ptr, 16, 10, 20, 0, 10, buf.ptr, None)                        # n_nb > n
    assert rc == -1
    h = _ffi.vp()
    assert L.kmap_embed_create(C.byref(h), 0, 0, 0, 10, 0.01, 0) == -1                              # n = 0
    assert L.kmap_embed_create(C.byref(h), 10, 0, 10, 100, 0.01, 0) == -1                           # n_best > 64
    assert L.kmap_embed_create(C.byref(h), 10, 0, 10, 10, 0.01, 7) == -1                            # unknown mode
    assert L.kmap_embed_create(C.byref(h), 10, 0, 10, 10, 0.01, 0) == 0
    assert L.kmap_embed_step(h.value, 1, None) == -1                                                 # nothing set yet
    assert b"not set" in L.kmap_last_error()
    L.kmap_embed_destroy(h.value)
    c = _ffi.vp()
    L.kmap_counts_create(C.byref(c))
    assert L.kmap_counts_fetch(c.value, None, None) == -1                                            # nothing counted
    L.kmap_counts_destroy(c.value)
    assert L.kmap_pack_planes_dev(buf.ptr + 4, 64, buf.ptr + 512, None) == -1                        # an offset into the code array
    assert b"aligned" in L.kmap_last_error()
    with pytest.raises(ValueError):
        _ffi.check(-1)


def test_empty_inputs_are_fine():
    import kmap_amd.kmer_count as K
    from kmap_amd.hamdist import hamdist_matrix_u8
    assert len(K.comp_kmer_hash(np.zeros(0, np.uint8), 8)) == 0
    assert len(K.cal_hamming_dist(np.zeros(0, np.uint32), 0, 8)) == 0
    assert hamdist_matrix_u8(np.zeros(0, np.uint64), np.zeros(0, np.int32), 8, [8]).shape == (0, 0)
    u, c = K.count_uniq_hash(np.full(50, 0xFFFFFFFF, np.uint32), 8)          # only invalid hashes
    assert len(u) == 0 and len(c) == 0
    out = K.mask_input(np.zeros(0, np.uint8), 4, np.array([1]), np.array([0]))
    assert len(out) == 0


def test_shared_histogram_table_has_one_owner():
    """All counts handles of a device share ONE 4^k-bin table.  The split API hist -> bins -> (all-reduce) -> finish must not
    compact a table another handle has overwritten in between, nor one the arena has freed: KMAP_E_STATE, not wrong counts."""
    from kmap_amd import _ffi, synth
    from kmap_amd.kmer_count import DeviceCounts
    from kmap_amd.motif_discovery import DeviceSeq
    L = _ffi.lib()
    seq, borders = synth.synth_reads(3000, 60, 5)
    ds = DeviceSeq(seq, borders)
    a, b = DeviceCounts(), DeviceCounts()
    hist = lambda h, k: L.kmap_counts_hist_packed_dev(h._h, ds.codes.ptr, ds.inval_orig.ptr, ds.n, ds.borders.ptr, ds.n_seq, k, 0, None)
    p, nb, nu = _ffi.vp(), _ffi.i64(0), _ffi.i64(0)
    assert hist(a, 7) == 0
    assert L.kmap_counts_bins(a._h, C.byref(p), C.byref(nb)) == 0 and nb.value == 4 ** 7
    assert L.kmap_counts_finish(a._h, 7, 1, C.byref(nu), None) == 0
    a.k, a.n_uniq = 7, nu.value
    ref_u, ref_c = a.fetch()
    # interleaved: A fills the table, B counts, A must not finish
    assert hist(a, 7) == 0
    ds.count(b, 6, dedupe=False, merge_revcom=True)
    assert L.kmap_counts_bins(a._h, C.byref(p), C.byref(nb)) == -5 and b"shared histogram" in L.kmap_last_error()
    assert L.kmap_counts_finish(a._h, 7, 1, C.byref(nu), None) == -5
    # the arena released between hist and finish: the same refusal instead of a read of freed memory
    assert hist(a, 7) == 0
    assert L.kmap_scratch_release(0) == 0
    assert L.kmap_counts_finish(a._h, 7, 1, C.byref(nu), None) == -5
    # and the handle is fine again after a fresh histogram
    assert hist(a, 7) == 0 and L.kmap_counts_finish(a._h, 7, 1, C.byref(nu), None) == 0
    a.k, a.n_uniq = 7, nu.value
    u, c = a.fetch()
    np.testing.assert_array_equal(u, ref_u)
    np.testing.assert_array_equal(c, ref_c)
    for h in (a, b):
        h.close()
    ds.close()
