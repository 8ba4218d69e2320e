"""GPU parity of the headline kernel: all-pairs Hamming matrix (bit-exact vs oracle and golden)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_matrix_golden(golden):
    from kmap_amd.hamdist import cal_samp_kmer_hamdist_mat, _convert_to_block_arr
    s = golden("scan_testfa.npz")
    conseqs = [str(c) for c in s["samp_conseqs"]]
    k = int(s["hamdist_kmer_len"])
    U = cal_samp_kmer_hamdist_mat(s["samp_kh"], s["samp_cnts"], s["samp_label"], conseqs, k, uniq_dist_flag=True)
    np.testing.assert_array_equal(U, s["hamdist_uniq_u8"])
    M = cal_samp_kmer_hamdist_mat(s["samp_kh"], s["samp_cnts"], s["samp_label"], conseqs, k)
    assert M.dtype == np.int64 and M.shape == (300, 300)
    np.testing.assert_array_equal(M, s["hamdist_mat_u8"])
    np.testing.assert_array_equal(_convert_to_block_arr(s["samp_label"], s["samp_cnts"]), s["hamdist_label"])
    assert np.all(U.diagonal() == 0) and np.array_equal(U, U.T)   # reference kmap_tests.py:555-556


@pytest.mark.parametrize("n,k,lens", [(1, 8, []), (15, 8, [8]), (16, 8, [6, 8]), (1000, 8, [8, 5]), (1025, 11, [11, 9, 7]),
                                      (3000, 15, [15, 3]), (2049, 16, [16, 12]), (1500, 31, [31, 20, 4]), (0, 8, []),
                                      # n >= 4096, k <= 16: the tiled one-hot kernel (1 or 2 code words, ragged last lanes / rows)
                                      (4096, 8, [8, 5]), (5003, 8, [8, 6]), (4500, 5, [5, 2]), (4100, 12, [12, 7, 12]),
                                      (6000, 16, [16, 3]), (8200, 9, [9]), (4200, 17, [17, 9]), (4097, 8, [8, 4]), (4111, 16, [16, 9])])
def test_matrix_vs_oracle(n, k, lens):
    from kmap_amd.hamdist import hamdist_matrix_u8
    from oracle import oracle as O
    rng = np.random.default_rng(n + k)
    kh = rng.integers(0, 4 ** k, size=n, dtype=np.uint64)
    lab = rng.integers(0, len(lens) + 1, size=n).astype(np.int32)   # label len(lens) = noise
    got = hamdist_matrix_u8(kh, lab, k, lens)
    np.testing.assert_array_equal(got, O.hamdist_matrix_u8(kh, lab, k, lens))


def test_matrix_row_blocks_device(golden):
    """Row-sharded launches (the multi-GPU decomposition) tile to the same matrix."""
    from kmap_amd import _ffi
    from kmap_amd.hamdist import hamdist_matrix_dev, pitch_for
    from oracle import oracle as O
    rng = np.random.default_rng(0)
    n, k, lens = 5000, 8, [8, 6]
    kh = rng.integers(0, 4 ** k, size=n, dtype=np.uint64).astype(np.uint32)
    lab = np.sort(rng.integers(0, 3, size=n)).astype(np.int32)
    ld = pitch_for(n)
    kh_d, lab_d = _ffi.DeviceBuffer.from_numpy(kh), _ffi.DeviceBuffer.from_numpy(lab)
    out_d = _ffi.DeviceBuffer(n * ld)
    out_d.zero()
    for r0, nr in ((0, 1700), (1700, 1), (1701, 3299)):
        hamdist_matrix_dev(kh_d.ptr, lab_d.ptr, n, k, lens, out_d.ptr + r0 * ld, ld, row0=r0, nrows=nr)
    got = out_d.to_numpy(np.uint8, (n, ld))[:, :n]
    np.testing.assert_array_equal(got, O.hamdist_matrix_u8(kh, lab, k, lens))


def test_matrix_full_size_properties():
    """BASELINE config size (N=50k, k=8): size-independent properties instead of an O(N^2) CPU check:
    symmetry + zero diagonal on sampled blocks, exact agreement with the oracle on sampled rows,
    and a checksum of per-row sums against the oracle's row sums for those rows."""
    from kmap_amd import _ffi
    from kmap_amd.hamdist import hamdist_matrix_dev, pitch_for
    from oracle import oracle as O
    rng = np.random.default_rng(50)
    n, k, lens = 50_000, 8, [8, 7]
    kh = rng.integers(0, 4 ** k, size=n, dtype=np.uint64).astype(np.uint32)
    lab = np.sort(rng.integers(0, 3, size=n)).astype(np.int32)
    ld = pitch_for(n)
    kh_d, lab_d = _ffi.DeviceBuffer.from_numpy(kh), _ffi.DeviceBuffer.from_numpy(lab)
    out_d = _ffi.DeviceBuffer(n * ld)
    hamdist_matrix_dev(kh_d.ptr, lab_d.ptr, n, k, lens, out_d.ptr, ld)
    _ffi.sync()
    rows = np.unique(np.concatenate([[0, 1, 63, 64, n - 1], rng.integers(0, n, size=40)]))
    for r in rows:
        got = out_d.to_numpy(np.uint8, (n,), offset=int(r) * ld)
        a = np.uint64(kh[r])
        x = kh.astype(np.uint64) ^ a
        same = (lab == lab[r]) & (lab[r] < len(lens))
        sh = np.where(same, 2 * (k - (lens[lab[r]] if lab[r] < len(lens) else k)), 0).astype(np.uint64)
        x = x >> sh
        want = np.zeros(n, np.uint8)
        for i in range(k):
            want += ((x >> np.uint64(2 * i)) & np.uint64(3)) != 0
        np.testing.assert_array_equal(got, want)
        assert got[r] == 0
    # symmetry on a sampled 512x512 block pair
    a0, b0 = 1024, 40_000
    blk1 = np.stack([out_d.to_numpy(np.uint8, (512,), offset=(a0 + i) * ld + b0) for i in range(512)])
    blk2 = np.stack([out_d.to_numpy(np.uint8, (512,), offset=(b0 + i) * ld + a0) for i in range(512)])
    np.testing.assert_array_equal(blk1, blk2.T)
