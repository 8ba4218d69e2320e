"""Pins the CPU oracle (oracle/) against golden vectors produced by running the reference
itself (tests/golden/gen_golden.py).  CPU only.  Integer paths: bit-exact.  Float paths:
bit-exact where the arithmetic is IEEE-defined, else the tolerance is written in the test."""
import numpy as np
import pytest

from oracle import oracle as O

GOLD = "tests/golden"


# ---- reference known answers restated from tests/kmap_tests.py ------------------------------
def test_kmer2hash_hash2kmer_known(golden):
    g = golden("ops.npz")
    for s, h, b, rc in zip(g["kmer_strs"], g["kmer_hashes"], g["kmer_back"], g["kmer_rc_hashes"]):
        s = str(s)
        assert int(O.kmer2hash(s)) == int(h)
        assert O.hash2kmer(h, len(s)) == s == str(b)
        assert int(O.revcom_hash(h, len(s))) == int(rc)
        assert O.hash2kmer(rc, len(s)) == O.reverse_complement(s)
    # kmap_tests.py:241-266 (k=5 and k=28 round trips)
    assert O.hash2kmer(O.kmer2hash("ACTACTGGAGGACCTACGTAAGCCACGA"), 28) == "ACTACTGGAGGACCTACGTAAGCCACGA"


@pytest.mark.parametrize("k", [3, 5, 8, 15, 16, 20, 31])
def test_hash_and_count_string_with_N(golden, k):
    g = golden("ops.npz")
    h = O.comp_kmer_hash(g["nseq_arr"], k)
    assert h.dtype == g[f"nseq_hash_k{k}"].dtype
    np.testing.assert_array_equal(h, g[f"nseq_hash_k{k}"])
    u, c = O.count_uniq_hash(h, k)
    np.testing.assert_array_equal(u, g[f"nseq_uniq_k{k}"])
    np.testing.assert_array_equal(c, g[f"nseq_cnt_k{k}"])
    assert c.dtype == g[f"nseq_cnt_k{k}"].dtype


def test_count_matches_independent_counter(golden):
    """kmap_tests.py:173-189: counts equal an independent k=3 scan that skips windows with N."""
    g = golden("ops.npz")
    s = O.arr2dna(g["nseq_arr"][:-1])
    want = {}
    for i in range(len(s) - 2):
        w = s[i:i + 3]
        if "N" not in w:
            want[int(O.kmer2hash(w))] = want.get(int(O.kmer2hash(w)), 0) + 1
    u, c = O.count_uniq_hash(O.comp_kmer_hash(g["nseq_arr"], 3), 3)
    assert dict(zip(map(int, u), map(int, c))) == want


@pytest.mark.parametrize("k", [3, 4, 8, 15, 16, 21, 31])
def test_revcom_arr(golden, k):
    g = golden("ops.npz")
    np.testing.assert_array_equal(O.get_revcom_hash_arr(g[f"rc_in_k{k}"], k), g[f"rc_out_k{k}"])


def test_merge_revcom(golden):
    g = golden("ops.npz")
    for i in range(int(g["mrc_n"])):
        k = int(g[f"mrc{i}_k"])
        u, c = O.merge_revcom(g[f"mrc{i}_in_kh"], g[f"mrc{i}_in_cnt"], k)
        np.testing.assert_array_equal(u, g[f"mrc{i}_out_kh"])
        np.testing.assert_array_equal(c, g[f"mrc{i}_out_cnt"])
        assert u.dtype == g[f"mrc{i}_out_kh"].dtype and c.dtype == g[f"mrc{i}_out_cnt"].dtype
    # kmap_tests.py:229-230: palindrome-free k=3 pairs merge to 2; ACGT (k=4) doubles
    u, c = O.merge_revcom(np.array([27], np.uint32), np.array([100], np.int32), 4)
    assert list(u) == [27] and list(c) == [200]


@pytest.mark.parametrize("k", [8, 12, 15, 16, 20, 31])
def test_hamming_ops(golden, k):
    g = golden("ops.npz")
    h = g[f"ham_in_k{k}"]
    np.testing.assert_array_equal(O.cal_hamming_dist(h, g[f"ham_cons_k{k}"], k), g[f"ham_out_k{k}"])
    cl = int(g[f"ham_clen_k{k}"])
    np.testing.assert_array_equal(O.cal_hamming_dist_head(h, g[f"ham_scons_k{k}"], k, cl), g[f"ham_head_k{k}"])
    np.testing.assert_array_equal(O.cal_hamming_dist_tail(h, g[f"ham_scons_k{k}"], k, cl), g[f"ham_tail_k{k}"])


def test_mask_input(golden):
    g = golden("ops.npz")
    for i in range(int(g["mask_n"])):
        out = O.mask_input(g[f"mask{i}_in"].copy(), int(g[f"mask{i}_k"]), g[f"mask{i}_cons"], g[f"mask{i}_r"])
        np.testing.assert_array_equal(out, g[f"mask{i}_out"])
    # the poly-T / separator quirk spelled out
    a = np.concatenate([O.dna2arr("ACGTACGTAC"), O.dna2arr("GGGGGGGGGG")])
    out = O.mask_input(a, 4, np.array([O.kmer2hash("TTTT")]), np.array([0]))
    assert O.arr2dna(out) == "ACGTACGNNNNNNNGGGGNNNN"


def test_mask_ham_ball_known_strings(golden):
    """kmap_tests.py:268-284 exact masked strings (r from motif_def_table: k=7,8,9 -> 1,2,2)."""
    g = golden("ops.npz")
    a = O.dna2arr(str(g["mhb_in1"]))[:-1].copy()
    O.mask_input(a, 3, [O.kmer2hash("AAA")], [0])
    O.mask_input(a, 4, [O.kmer2hash("CCCC")], [0])
    assert O.arr2dna(a) == str(g["mhb_out1"]) == "NNNNNNNNNNNNNNNNNNNNNNCTAGCTGCCAGTNNNNNNNNNNN"
    a = O.dna2arr(str(g["mhb_in2"]))[:-1].copy()
    for s, r in (("AAAAAAA", 1), ("CCCCCCCC", 2), ("GGGGGGGGG", 2)):
        O.mask_input(a, len(s), [O.kmer2hash(s)], [r])
    assert O.arr2dna(a) == str(g["mhb_out2"])


@pytest.mark.parametrize("k", [3, 4, 16])
def test_dedupe(golden, k):
    g = golden("ops.npz")
    out = O.remove_duplicate_hash_per_seq(g[f"dd_hash_k{k}"].copy(), g["dd_borders"])
    np.testing.assert_array_equal(out, g[f"dd_out_k{k}"])


# ---- test.fa pipeline fixtures ------------------------------------------------------------------
@pytest.mark.parametrize("k", [6, 8, 9, 14, 16])
@pytest.mark.parametrize("rep", [0, 1])
def test_counts_testfa(golden, k, rep):
    s, c = golden("scan_testfa.npz"), golden("counts_testfa.npz")
    h = O.comp_kmer_hash(s["seq"], k)
    if k in (8, 16):
        np.testing.assert_array_equal(h, c[f"hash_k{k}"])
    if not rep:
        h = O.remove_duplicate_hash_per_seq(h, s["borders"])
    u, n = O.count_uniq_hash(h, k)
    tag = f"k{k}_rep{rep}"
    np.testing.assert_array_equal(u, c[f"{tag}_uniq"])
    np.testing.assert_array_equal(n, c[f"{tag}_cnt"])
    mu, mn = O.merge_revcom(u, n, k)
    np.testing.assert_array_equal(mu, c[f"{tag}_muniq"])
    np.testing.assert_array_equal(mn, c[f"{tag}_mcnt"])


def test_k8_default_mode_totals(golden):
    s = golden("scan_testfa.npz")
    u, c = O.count_kmers(s["seq"], s["borders"], 8, rep_mode=False)
    np.testing.assert_array_equal(u, s["k8_uniq"])
    np.testing.assert_array_equal(c, s["k8_cnt"])


@pytest.mark.parametrize("k,rep", [(8, 0), (10, 0), (8, 1)])
def test_find_motif(golden, motif_defs, k, rep):
    s, f = golden("scan_testfa.npz"), golden("find_motif_testfa.npz")
    res, _, _ = O.find_motif(s["seq"].copy(), s["borders"], k, motif_defs[k], rep_mode=bool(rep))
    tag = f"k{k}_rep{rep}"
    assert [int(x) for x in res.keys()] == [int(x) for x in f[f"{tag}_kh"]]
    got = np.array([list(v) for v in res.values()]).reshape(-1, 3)
    np.testing.assert_allclose(got, f[f"{tag}_vals"], rtol=1e-12, atol=0)


def _read_lines(name):
    with open(f"{GOLD}/scan_testfa/{name}") as fh:
        return fh.read().splitlines()


def test_motif_occurence_file(golden, motif_defs):
    s = golden("scan_testfa.npz")
    r_of = {k: d.max_ham_dist for k, d in motif_defs.items()}
    finals = _read_lines("final_conseq.txt")
    rng = np.random.RandomState(0)  # no read of test.fa has > 20 hits at minimum distance for these
    lines = O.motif_occurence_lines(s["seq"], s["borders"], finals, r_of, True, rng)
    assert lines == _read_lines("final.motif_occurence.csv")


def test_hamdist_matrix(golden):
    s = golden("scan_testfa.npz")
    conseqs = [str(c) for c in s["samp_conseqs"]]
    k = int(s["hamdist_kmer_len"])
    U = O.cal_samp_kmer_hamdist_mat(s["samp_kh"], s["samp_cnts"], s["samp_label"], conseqs, k, uniq_dist_flag=True)
    np.testing.assert_array_equal(U, s["hamdist_uniq_u8"])
    M = O.cal_samp_kmer_hamdist_mat(s["samp_kh"], s["samp_cnts"], s["samp_label"], conseqs, k)
    assert M.dtype == np.int64
    np.testing.assert_array_equal(M, s["hamdist_mat_u8"])
    np.testing.assert_array_equal(O.convert_to_block_arr(s["samp_label"], s["samp_cnts"]), s["hamdist_label"])
    # invariants the reference tests assert (kmap_tests.py:555-556)
    assert np.all(U.diagonal() == 0) and np.array_equal(U, U.T)


def test_sample_disp_kmer_labels(golden, motif_defs):
    """Labelling + revcom alignment are deterministic; the multinomial draw needs the reference's
    RNG state (np.random.seed(123) then every draw scan_motif made before), so check support."""
    s = golden("scan_testfa.npz")
    k = int(s["hamdist_kmer_len"])
    r_of = {kk: d.max_ham_dist for kk, d in motif_defs.items()}
    conseqs = _read_lines("final_conseq.txt")
    u, c, lab, cl = O.sample_disp_kmer(conseqs, k, r_of, s[f"k{k}_uniq"], s[f"k{k}_cnt"], 10 ** 9, 0)
    look = {int(h): int(l) for h, l in zip(u, lab)}
    assert cl == [str(x) for x in s["samp_conseqs"]]
    for h, l in zip(s["samp_kh"], s["samp_label"]):
        assert look[int(h)] == int(l)


# ---- embedding ---------------------------------------------------------------------------------------
def test_knn_smooth(golden):
    e, s = golden("embed_ops.npz"), golden("scan_testfa.npz")
    S = O.knn_smooth(s["hamdist_mat_u8"].astype(np.int64), int(e["n_nb"]), nb=e["nb"])
    np.testing.assert_array_equal(S, e["S"])  # integer sums: exact in f32
    S2 = O.knn_smooth(e["small_D"], 4, nb=e["small_nb"])
    np.testing.assert_array_equal(S2, e["small_S"])
    # reference tolerance precedent vs an f64 triple loop: 1e-4 (kmap_tests.py:612)
    D, nb = e["small_D"].astype(float), e["small_nb"]
    ref = np.array([[D[np.ix_(nb[i], nb[j])].sum() / 16 if i != j else 0 for j in range(10)] for i in range(10)])
    assert np.all(np.abs(S2 - ref) < 1e-4)


def test_sigmoid_hd_prob(golden):
    """numpy f32 exp may differ by an ulp between CPU ISAs: tolerance 2 ulp-ish (rtol 3e-7)."""
    e = golden("embed_ops.npz")
    k = int(e["kmer_len"])
    T = O.sigmoid(e["S"], 16.0, change_point=k / 2, scale_factor=0.2 * k - 0.2)
    assert T.dtype == np.float32
    np.testing.assert_allclose(T, e["sig"], rtol=3e-7, atol=0)
    np.testing.assert_allclose(O.hd_prob_from_smooth(e["S"], k), e["hd_prob"], rtol=2e-6, atol=1e-45)


def test_embed_ops(golden):
    e = golden("embed_ops.npz")
    q = O.cal_ld_prob_mat(e["op_ld"])
    np.testing.assert_array_equal(q, e["op_q"])  # IEEE f32 sub/mul/add/div only
    np.testing.assert_array_equal(O.gradient_loss(e["op_p"], e["op_q"], e["op_ld"]), e["op_grad"])
    np.testing.assert_allclose(O.cross_entropy(e["op_p"], e["op_q"]), e["op_loss"], rtol=1e-6)


@pytest.mark.parametrize("tag", ["n96", "n300"])
def test_umap_trace(golden, tag):
    u = golden(f"umap_{tag}.npz")
    k = int(u["kmer_len"])
    tr = {}
    final = O.kmap(u["D"].astype(np.int64), k, n_max_iter=int(u["n_iter"]), random_seed=int(u["seed"]),
                   nb=u["nb"], trace=tr)
    np.testing.assert_allclose(np.array(tr["losses"], np.float32), u["losses"], rtol=2e-6)
    coords = np.array(tr["coords"])
    assert coords.shape == u["coords"].shape
    # coordinates depend only on IEEE f32 ops + numpy's exp for p (<= ~1 ulp across ISAs): 1e-5 abs
    np.testing.assert_allclose(coords, u["coords"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(final, u["final"], rtol=0, atol=1e-5)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_umap_early_stop(golden, tag):
    """the reference's early-stop rule (visualization.py:310-311): runs that ended before n_max_iter"""
    u = golden("umap_earlystop.npz")
    tr = {}
    final = O.kmap(u[f"{tag}_D"].astype(np.int64), int(u["kmer_len"]), n_neighbour=int(u["n_nb"]), n_max_iter=int(u[f"{tag}_n_max_iter"]),
                   learning_rate=float(u[f"{tag}_lr"]), random_seed=int(u[f"{tag}_seed"]), nb=u[f"{tag}_nb"], trace=tr)
    assert len(tr["losses"]) == len(u[f"{tag}_losses"]) < int(u[f"{tag}_n_max_iter"])
    np.testing.assert_allclose(np.array(tr["losses"], np.float32), u[f"{tag}_losses"], rtol=2e-6)
    np.testing.assert_allclose(final, u[f"{tag}_final"], rtol=0, atol=1e-5 * max(1.0, float(np.abs(u[f"{tag}_final"]).max())))


# ---- report consumers: oracle and host logic vs the reference's outputs -----------------------
from pathlib import Path  # noqa: E402

RGOLD = Path(GOLD) / "report_testfa"

def _parse_occ(path):
    import csv
    rows = []
    with open(path, newline="") as fh:
        rd = csv.reader(fh, delimiter=";")
        next(rd)
        for row in rd:
            rows.append(([[int(v) for v in c.split(",")] if c.strip() else [] for c in row[1:-1]], float(row[-1])))
    return rows


def test_report_hamball_and_cnt_mat(golden):
    g = golden("report.npz")
    for tag in "abcd":
        conseq, r, rc = g[f"ball_{tag}_def"]
        k, r = len(conseq), int(r)
        if r == -1:
            from kmap_amd.kmer_count import init_motif_def_dict
            r = init_motif_def_dict(RGOLD / "motif_def_table.csv")[k].max_ham_dist
        u, c = O.ex_hamball(g[f"k{k}_uniq"], g[f"k{k}_cnt"], k, O.kmer2hash(conseq), r, bool(int(rc)))
        np.testing.assert_array_equal(u, g[f"ball_{tag}_kh"])
        np.testing.assert_array_equal(c, g[f"ball_{tag}_cnt"])
        np.testing.assert_array_equal(O.cal_cnt_mat(u, c, k), g[f"ball_{tag}_mat"])


def test_report_pos_density_oracle(golden):
    g = golden("report.npz")
    rows = _parse_occ(RGOLD / "synth4.motif_occurence.csv")
    for i, name in enumerate(g["s4_names"]):
        sel = [(cells[i], sl) for cells, sl in rows if cells[i]]
        d = O.motif_pos_density(sel, len(name), np.arange(0, 1.01, 0.01), 0.01)
        np.testing.assert_allclose(d, g[f"s4_dens_{i}"], rtol=1e-13, atol=1e-300)
        assert [len(sel), sum(len(c) for c, _ in sel)] == list(g[f"s4_dens_{i}_n"])


def test_report_co_occurrence_oracle_and_host(golden, tmp_path):
    """the oracle's per-row restatement and the product's vectorised host code both reproduce the reference's matrices,
    distance lists and the four text files (synthetic 4-motif file and tests/test.fa)"""
    from kmap_amd import reports as R
    g = golden("report.npz")
    src = RGOLD
    rows = _parse_occ(src / "synth4.motif_occurence.csv")
    names = [str(s) for s in g["s4_names"]]
    for res, dist, dd in (O.co_occurrence([c for c, _ in rows], 4), R.get_motif_co_occurence_mat(src / "synth4.motif_occurence.csv", 4)):
        np.testing.assert_array_equal(res, g["s4_co"])
        np.testing.assert_array_equal(dist, g["s4_dist"])
        for (i, j), v in dd.items():
            np.testing.assert_array_equal(np.array(v, np.float64), g[f"s4_dd_{i}_{j}"])
    co, dist, dd = R.get_motif_co_occurence_mat(src / "synth4.motif_occurence.csv", 4)
    co_sum = np.diag(co) + np.diag(co).reshape((-1, 1))
    R.write_co_occurence_mat(tmp_path / "a.tsv", co + 0.0, names)
    R.write_co_occurence_mat(tmp_path / "b.tsv", 2 * co / co_sum, names)
    R.write_co_occurence_mat(tmp_path / "c.tsv", dist, names)
    R.write_co_occurence_dist_arr(tmp_path / "d.txt", dd, names)
    for mine, ref in (("a.tsv", "s4_co_occurence_mat.tsv"), ("b.tsv", "s4_co_occurence_mat.norm.tsv"),
                      ("c.tsv", "s4_co_occurence_motif_dist_mat.tsv"), ("d.txt", "s4_co_occurence_motif_dist_data.txt")):
        assert (tmp_path / mine).read_text() == (src / ref).read_text(), ref
    for i in range(4):
        assert list(R.get_motif_seq_num(src / "synth4.motif_occurence.csv", i)) == list(g[f"s4_seqnum_{i}"])
    # tests/test.fa finals
    finals = [str(s) for s in g["final_conseq"]]
    co, dist, dd = R.get_motif_co_occurence_mat(src / "final.motif_occurence.csv", len(finals))
    R.write_co_occurence_mat(tmp_path / "e.tsv", co + 0.0, finals)
    R.write_co_occurence_dist_arr(tmp_path / "f.txt", dd, finals)
    assert (tmp_path / "e.tsv").read_text() == (src / "co_occurence_mat.tsv").read_text()
    assert (tmp_path / "f.txt").read_text() == (src / "co_occurence_motif_dist_data.txt").read_text()


def test_occurrence_subsample_golden():
    """reads with more than 20 hits at the minimum distance: the oracle's np.random.choice subsample reproduces the
    reference's rows (tests/golden/occ20, seed 77) -- this pins the RNG protocol the GPU tests check the product against"""
    from kmap_amd.kmer_count import encode_fasta_py, init_motif_def_dict, _pkg_file
    seq, borders = encode_fasta_py(f"{GOLD}/occ20/occ20.fa")
    mdd = init_motif_def_dict(_pkg_file("default_motif_def_table.csv"))
    r_of = {k: d.max_ham_dist for k, d in mdd.items() if isinstance(k, int)}
    rng = np.random.RandomState(77)
    lines = O.motif_occurence_lines(seq, borders, ["AAAAAAAA", "ACGTACGT", "AACCGGTTAA"], r_of, True, rng)
    want = open(f"{GOLD}/occ20/occ20.motif_occurence.csv").read().splitlines()
    assert lines == want
    assert sum(cell.count(",") == 19 for ln in want[1:] for cell in ln.split(";")[1:-1]) >= 10


# ---- the timed CPU baseline (oracle/kmap_cpu_baseline.c) computes what the oracle computes ----------------------------------
def test_cpu_baseline_equals_oracle():
    """bench.py's cpu_baseline leg times oracle/kmap_cpu_baseline.c (per-thread histograms, hashed per-read sets, fused embedding
    pass); a baseline that computed something else would be meaningless, so every function is pinned against the oracle."""
    from oracle import baseline as B, oracle as O
    from kmap_amd import synth
    from kmap_amd.kmer_count import gen_motif_def_dict, read_default_config_file
    seq, borders = synth.synth_reads(3000, 75, 11)
    seq[1000:1030] = 255                                       # a masked stretch inside a read
    seq[5020:5060] = 3                                         # poly-T inside read 66 ([5016, 5091)): duplicates within one read
    for k in (6, 8, 11):
        for dedupe in (True, False):
            for merge in (True, False):
                h = O.comp_kmer_hash(seq, k)
                if dedupe:
                    h = O.remove_duplicate_hash_per_seq(h, borders)
                u, c = O.count_uniq_hash(h, k)
                if merge:
                    u, c = O.merge_revcom(u, c, k)
                bu, bc = B.count(seq, borders, k, dedupe, merge, threads=3)
                assert dict(zip(u.tolist(), c.tolist())) == dict(zip(bu.tolist(), bc.tolist())), (k, dedupe, merge)
    # masking incl. the poly-T / separator quirk (golden G5) and a radius-1 ball
    m = O.dna2arr("ACGTACGTAC")
    arr = np.concatenate([m, O.dna2arr("GGGGGGGGGG")]).astype(np.uint8)
    want = O.mask_input(arr.copy(), 4, np.array([O.kmer2hash("TTTT")], np.uint64), np.array([0]))
    got = B.mask(arr.copy(), 4, [O.kmer2hash("TTTT")], [0], threads=2)
    np.testing.assert_array_equal(got, want)
    cons = [int(O.kmer2hash("AATCGATA")), int(O.revcom_hash(O.kmer2hash("AATCGATA"), 8))]
    np.testing.assert_array_equal(B.mask(seq.copy(), 8, cons, [1, 1], threads=3),
                                  O.mask_input(seq.copy(), 8, np.array(cons, np.uint64), np.array([1, 1])))
    # find_motif: same consensus sequences and proportions as the oracle's restatement of the reference loop
    mdd = gen_motif_def_dict(read_default_config_file())
    seq2, borders2 = synth.synth_reads(6000, 60, 5)
    for k in (7, 8, 9):
        res_o, _, _ = O.find_motif(seq2.copy(), borders2, k, mdd[k])
        res_b = B.find_motif(seq2.copy(), borders2, k, mdd[k], threads=3)
        assert [int(h) for h in res_o] == list(res_b), k
        for h, (prop, _, _) in res_o.items():
            assert abs(res_b[int(h)] - prop) <= 1e-12
        assert len(res_b) >= 1
    # one fused embedding pass: the gradient bit for bit (same f32 order), the loss to f32 round-off
    rng = np.random.default_rng(2)
    n = 257
    kh = rng.integers(0, 4 ** 8, size=n).astype(np.uint32)
    lut = np.exp(-np.arange(9, dtype=np.float32) / np.float32(3.0)).astype(np.float32)
    P = np.empty((n, n), np.float32)
    B.lib().kb_fill_prob(kh, n, lut, 1, P, 2)
    D = O.hamdist_matrix_u8(kh.astype(np.uint64), np.zeros(n, np.int32), 8, [8])
    np.testing.assert_array_equal(P, lut[D])
    np.fill_diagonal(P, 0.0)
    y = (rng.standard_normal((2, n)) * 2).astype(np.float32)
    q = O.cal_ld_prob_mat(y)
    g, loss = B.embed_forces(P, y, threads=3)
    np.testing.assert_array_equal((np.float32(4.0) * g).view(np.uint32), O.gradient_loss(P, q, y).view(np.uint32))
    ref = float(O.cross_entropy(P, q))
    assert abs(2.0 * loss - ref) <= 3e-6 * abs(ref)
    rows = np.array([0, 3, 100, 256, 17], np.int64)             # the row-slab form (checker of the SEQ kernel at sizes without a host matrix)
    gr = B.embed_forces_rows(P[rows], rows, y, threads=2)
    np.testing.assert_array_equal(gr.view(np.uint32), g[:, rows].view(np.uint32))
    P_off = P.copy()                                            # the on-the-fly form bench.py times: p from the k-mers, no matrix
    g3, l3 = B.embed_forces_kmers(kh, lut, 1, y, threads=3)
    B.lib().kb_fill_prob(kh, n, lut, 1, P_off, 2)              # (its diagonal is lut[0]; the pass skips j == i either way)
    g4, l4 = B.embed_forces(P_off, y, threads=3)
    np.testing.assert_array_equal(g3, g4)
    assert l3 == l4
    g2, l2 = B.embed_forces(P, y, 100, 180, threads=2)          # a row range (the 1-core sample of bench.py)
    np.testing.assert_array_equal(g2[:, 100:180], g[:, 100:180])
    assert not g2[:, :100].any() and not g2[:, 180:].any() and 0 < l2 < loss
    info = B.host_cpu_info()
    assert info["usable"] >= 1 and info["logical_cpus"] >= info["usable"] and isinstance(info["model"], str)
