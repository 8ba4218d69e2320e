"""GPU parity tests of the k-mer operators: HIP (through the C ABI) vs the CPU oracle on seeded
inputs and vs the golden fixtures generated from the reference.  Integer work: bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def K():
    import kmap_amd.kmer_count as kc
    return kc


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


def synth_reads(rng, n_reads, lo, hi, p_n=0.01):
    """uint8 array with 255 separators + borders; a few N (255) bases inside reads."""
    lens = rng.integers(lo, hi + 1, size=n_reads)
    parts, st, borders = [], 0, []
    for L in lens:
        r = rng.integers(0, 4, size=L).astype(np.uint8)
        r[rng.random(L) < p_n] = 255
        parts += [r, np.array([255], np.uint8)]
        borders.append((st, st + L))
        st += L + 1
    return np.concatenate(parts), np.array(borders, dtype=np.int64)


@pytest.mark.parametrize("k", [1, 3, 8, 15, 16, 23, 31])
def test_hash_matches_oracle(K, O, k):
    rng = np.random.default_rng(k)
    seq, _ = synth_reads(rng, 300, 5, 200)
    np.testing.assert_array_equal(K.comp_kmer_hash(seq, k), O.comp_kmer_hash(seq, k))


def test_hash_edge_cases(K, O):
    for seq in (np.zeros(0, np.uint8), np.array([255], np.uint8), np.array([0, 1, 2], np.uint8),
                np.full(40, 255, np.uint8), np.zeros(5000, np.uint8)):
        for k in (1, 4, 16):
            np.testing.assert_array_equal(K.comp_kmer_hash(seq, k), O.comp_kmer_hash(seq, k))
    with pytest.raises(Exception):
        K.comp_kmer_hash(np.zeros(10, np.uint8), 32)   # kmer_count.py:365
    with pytest.raises(Exception):
        K.comp_kmer_hash(np.zeros(10, np.uint8), 0)


@pytest.mark.parametrize("k", [3, 5, 8, 15, 16, 20, 31])
def test_hash_count_golden(K, golden, k):
    g = golden("ops.npz")
    h = K.comp_kmer_hash(g["nseq_arr"], k)
    np.testing.assert_array_equal(h, g[f"nseq_hash_k{k}"])
    if True:   # k <= 16: histogram path; k >= 17: sort + run-length path
        u, c = K.count_uniq_hash(h, k)
        np.testing.assert_array_equal(u, g[f"nseq_uniq_k{k}"])
        np.testing.assert_array_equal(c, g[f"nseq_cnt_k{k}"])
        assert u.dtype == g[f"nseq_uniq_k{k}"].dtype and c.dtype == g[f"nseq_cnt_k{k}"].dtype


@pytest.mark.parametrize("k", [3, 4, 8, 15, 16, 21, 31])
def test_revcom_golden(K, golden, k):
    g = golden("ops.npz")
    np.testing.assert_array_equal(K.get_revcom_hash_arr(g[f"rc_in_k{k}"], k), g[f"rc_out_k{k}"])
    for h in g[f"rc_in_k{k}"][:5]:
        assert int(K.revcom_hash(h, k)) == int(K.get_revcom_hash_arr(np.array([h]), k)[0])


def test_merge_revcom_golden(K, golden):
    g = golden("ops.npz")
    for i in range(int(g["mrc_n"])):
        k = int(g[f"mrc{i}_k"])
        u, c = K.merge_revcom(g[f"mrc{i}_in_kh"].copy(), g[f"mrc{i}_in_cnt"].copy(), k)
        np.testing.assert_array_equal(u, g[f"mrc{i}_out_kh"])
        np.testing.assert_array_equal(c, g[f"mrc{i}_out_cnt"])


@pytest.mark.parametrize("k", [8, 12, 15, 16, 20, 31])
def test_hamming_golden(K, golden, k):
    g = golden("ops.npz")
    h = g[f"ham_in_k{k}"]
    np.testing.assert_array_equal(K.cal_hamming_dist(h, g[f"ham_cons_k{k}"], k), g[f"ham_out_k{k}"])
    cl = int(g[f"ham_clen_k{k}"])
    np.testing.assert_array_equal(K.cal_hamming_dist_head(h, g[f"ham_scons_k{k}"], k, cl), g[f"ham_head_k{k}"])
    np.testing.assert_array_equal(K.cal_hamming_dist_tail(h, g[f"ham_scons_k{k}"], k, cl), g[f"ham_tail_k{k}"])


def test_hamming_large_vs_oracle(K, O):
    rng = np.random.default_rng(3)
    for k in (8, 27):
        dt = K.get_hash_dtype(k)
        h = rng.integers(0, 4 ** k, size=1_000_003, dtype=np.uint64).astype(dt)
        h[::1001] = K.get_invalid_hash(dt)
        c = dt(rng.integers(0, 4 ** k, dtype=np.uint64))
        np.testing.assert_array_equal(K.cal_hamming_dist(h, c, k), O.cal_hamming_dist(h, c, k))


def test_mask_golden(K, golden):
    g = golden("ops.npz")
    for i in range(int(g["mask_n"])):
        out = K.mask_input(g[f"mask{i}_in"].copy(), int(g[f"mask{i}_k"]), g[f"mask{i}_cons"], g[f"mask{i}_r"])
        np.testing.assert_array_equal(out, g[f"mask{i}_out"])
    a = np.concatenate([K.dna2arr("ACGTACGTAC"), K.dna2arr("GGGGGGGGGG")])
    out = K.mask_input(a, 4, np.array([K.kmer2hash("TTTT")]), np.array([0]))
    assert K.arr2dna(out) == "ACGTACGNNNNNNNGGGGNNNN"


def test_mask_ham_ball_known_strings(K, golden):
    """reference tests/kmap_tests.py:268-284"""
    g = golden("ops.npz")
    mdd = K.init_motif_def_dict(K._pkg_file("default_motif_def_table.csv"))
    a = K.dna2arr(str(g["mhb_in1"]))[:-1].copy()
    assert K.arr2dna(K.mask_ham_ball(a, mdd, ["AAA", "CCCC"], [0, 0])) == "NNNNNNNNNNNNNNNNNNNNNNCTAGCTGCCAGTNNNNNNNNNNN"
    a = K.dna2arr(str(g["mhb_in2"]))[:-1].copy()
    assert K.arr2dna(K.mask_ham_ball(a, mdd, ["AAAAAAA", "CCCCCCCC", "GGGGGGGGG"])) == str(g["mhb_out2"])


def test_mask_vs_oracle_random(K, O):
    rng = np.random.default_rng(11)
    seq, _ = synth_reads(rng, 2000, 30, 150)
    for k, r in ((6, 1), (8, 2), (17, 7)):
        cons = rng.integers(0, 4 ** k, size=3, dtype=np.uint64)
        cons[0] = int(O.kmer2hash("T" * k))   # poly-T: matches invalid hashes (reference quirk)
        rad = np.array([r, r, 0])
        np.testing.assert_array_equal(K.mask_input(seq.copy(), k, cons, rad), O.mask_input(seq.copy(), k, cons, rad))


@pytest.mark.parametrize("k", [3, 4, 16])
def test_dedupe_golden(K, golden, k):
    g = golden("ops.npz")
    out = K.remove_duplicate_hash_per_seq(g[f"dd_hash_k{k}"].copy(), g["dd_borders"])
    np.testing.assert_array_equal(out, g[f"dd_out_k{k}"])


def test_dedupe_ragged_and_long_reads(K, O):
    rng = np.random.default_rng(5)
    # low-complexity reads (many duplicates), ragged lengths incl. empty and > LDS capacity (512)
    lens = [0, 1, 2, 63, 64, 65, 500, 511, 512, 513, 1024, 1025, 5000, 20000, 3, 0, 777]
    parts, borders, st = [], [], 0
    for L in lens:
        parts += [rng.integers(0, 2, size=L).astype(np.uint8), np.array([255], np.uint8)]
        borders.append((st, st + L))
        st += L + 1
    seq, borders = np.concatenate(parts), np.array(borders, dtype=np.int64)
    for k in (4, 9, 16):
        h = O.comp_kmer_hash(seq, k)
        np.testing.assert_array_equal(K.remove_duplicate_hash_per_seq(h.copy(), borders),
                                      O.remove_duplicate_hash_per_seq(h.copy(), borders))


@pytest.mark.parametrize("k", [6, 8, 9, 14, 16])
@pytest.mark.parametrize("rep", [0, 1])
def test_counts_testfa_golden(K, golden, k, rep):
    """Fused device path: hash + dedupe + count (+ merge) on tests/test.fa == the reference's arrays."""
    from kmap_amd import _ffi
    s, c = golden("scan_testfa.npz"), golden("counts_testfa.npz")
    seq_d = _ffi.DeviceBuffer.from_numpy(s["seq"])
    bor_d = _ffi.DeviceBuffer.from_numpy(s["borders"].astype(np.int64))
    dc = K.DeviceCounts()
    tag = f"k{k}_rep{rep}"
    dc.run_seq(seq_d.ptr, len(s["seq"]), bor_d.ptr, len(s["borders"]), k, dedupe=not rep, merge_revcom=False)
    u, n = dc.fetch()
    np.testing.assert_array_equal(u, c[f"{tag}_uniq"])
    np.testing.assert_array_equal(n, c[f"{tag}_cnt"])
    assert dc.total() == int(n.sum())
    dc.run_seq(seq_d.ptr, len(s["seq"]), bor_d.ptr, len(s["borders"]), k, dedupe=not rep, merge_revcom=True)
    mu, mn = dc.fetch()
    np.testing.assert_array_equal(mu, c[f"{tag}_muniq"])
    np.testing.assert_array_equal(mn, c[f"{tag}_mcnt"])
    assert mu.dtype == c[f"{tag}_muniq"].dtype and mn.dtype == c[f"{tag}_mcnt"].dtype
    # Hamming-ball mass of a few candidates vs the oracle's restatement of find_motif's inner loop
    from oracle import oracle as O
    cands = mu[np.argsort(mn)[-5:]]
    np.testing.assert_array_equal(dc.hamball_mass(cands, 2, True), O.hamball_mass(mu, mn, k, cands, 2, True))
    np.testing.assert_array_equal(dc.hamball_mass(cands, 1, False), O.hamball_mass(mu, mn, k, cands, 1, False))
    dc.close()


def test_counts_large_k_sort_path_vs_oracle(K, O):
    """17 <= k < 32 (sort + run-length encode + binary-search revcom merge), with and without per-read dedupe."""
    rng = np.random.default_rng(23)
    seq, borders = synth_reads(rng, 3000, 40, 160)
    seq[5000:5400] = np.tile(np.array([0, 1], np.uint8), 200)          # low complexity: duplicates + revcom partners
    from kmap_amd import _ffi
    seq_d, bor_d = _ffi.DeviceBuffer.from_numpy(seq), _ffi.DeviceBuffer.from_numpy(borders)
    dc = K.DeviceCounts()
    for k in (17, 20, 31):
        for dedupe, merge in ((True, True), (False, True), (False, False)):
            dc.run_seq(seq_d.ptr, len(seq), bor_d.ptr, len(borders), k, dedupe=dedupe, merge_revcom=merge)
            u, c = dc.fetch()
            ou, oc = O.count_kmers(seq, borders, k, rep_mode=not dedupe, revcom_mode=merge)
            np.testing.assert_array_equal(u, ou)
            np.testing.assert_array_equal(c, oc)
            assert u.dtype == np.uint64 and c.dtype == np.int64 and dc.total() == int(oc.sum())
    cands = u[:3]
    np.testing.assert_array_equal(dc.hamball_mass(cands, 8, True), O.hamball_mass(u, c, 31, cands, 8, True))
    dc.close()


def test_counts_random_vs_oracle(K, O):
    rng = np.random.default_rng(21)
    seq, borders = synth_reads(rng, 20000, 20, 160)
    from kmap_amd import _ffi
    seq_d, bor_d = _ffi.DeviceBuffer.from_numpy(seq), _ffi.DeviceBuffer.from_numpy(borders)
    dc = K.DeviceCounts()
    for k in (5, 8, 11, 13):
        for dedupe in (True, False):
            dc.run_seq(seq_d.ptr, len(seq), bor_d.ptr, len(borders), k, dedupe=dedupe, merge_revcom=True)
            u, c = dc.fetch()
            ou, oc = O.count_kmers(seq, borders, k, rep_mode=not dedupe, revcom_mode=True)
            np.testing.assert_array_equal(u, ou)
            np.testing.assert_array_equal(c, oc)
            # device top-k: largest count first, ties by lowest index
            for top_k in (1, 5, 16):
                want = np.lexsort((np.arange(len(oc)), -oc.astype(np.int64)))[:top_k]
                idx, kh, cn = dc.topk(top_k)
                np.testing.assert_array_equal(idx, want)
                np.testing.assert_array_equal(kh, ou[want])
                np.testing.assert_array_equal(cn, oc[want])
    with pytest.raises(ValueError):
        dc.topk(17)
    dc.close()
