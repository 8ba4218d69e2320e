"""GPU parity of the report consumers: Hamming-ball extraction / count matrix, motif position
density, and the scan_motif branches behind motif_pos_density_flag / motif_co_occurence_flag / gen_hamball_flag, against
the reference's outputs (tests/golden/report.npz, tests/golden/report_testfa/) and against the oracle on random inputs."""
import pickle
import shutil
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).resolve().parent / "golden"
RGOLD = GOLD / "report_testfa"


@pytest.fixture(scope="module")
def R():
    from kmap_amd import reports
    return reports


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


@pytest.fixture(scope="module")
def count_dir(tmp_path_factory, golden):
    """a res_dir holding the reference's k{k}.pkl arrays, its config.toml and motif_def_table.csv"""
    g = golden("report.npz")
    res = tmp_path_factory.mktemp("rep") / "res"
    (res / "kmer_count").mkdir(parents=True)
    for key in g:
        if key.endswith("_uniq") and key.startswith("k"):
            k = int(key[1:-5])
            with open(res / "kmer_count" / f"k{k}.pkl", "wb") as fh:
                pickle.dump([k, g[f"k{k}_uniq"], g[f"k{k}_cnt"]], fh)
    shutil.copyfile(RGOLD / "config.toml", res / "config.toml")
    shutil.copyfile(RGOLD / "motif_def_table.csv", res / "motif_def_table.csv")
    return res


def test_ex_hamball_kh_arr_golden(R, golden, count_dir):
    g = golden("report.npz")
    for tag in "abcd":
        conseq, r, rc = g[f"ball_{tag}_def"]
        u, c = R.ex_hamball_kh_arr(str(count_dir), str(conseq), int(r), str(count_dir / "motif_def_table.csv"), bool(int(rc)))
        np.testing.assert_array_equal(u, g[f"ball_{tag}_kh"])
        np.testing.assert_array_equal(c, g[f"ball_{tag}_cnt"])
        assert u.dtype == g[f"ball_{tag}_kh"].dtype and c.dtype == g[f"ball_{tag}_cnt"].dtype
        m = R.cal_cnt_mat(u, c, len(str(conseq)))
        np.testing.assert_array_equal(m, g[f"ball_{tag}_mat"])
        assert m.dtype == g[f"ball_{tag}_mat"].dtype


@pytest.mark.parametrize("rt", ["hash", "kmer", "matrix"])
def test_ex_hamball_files_golden(R, golden, count_dir, tmp_path, rt):
    conseq = str(golden("report.npz")["final_conseq"][0])
    R._ex_hamball(str(count_dir), conseq, rt, str(tmp_path / "out.txt"), max_ham_dist=1)
    assert (tmp_path / "out.txt").read_text() == (RGOLD / f"exhamball_{rt}.txt").read_text()


@pytest.mark.parametrize("k", [5, 8, 15, 16, 21, 31])
def test_hamball_extract_random_vs_oracle(R, O, k):
    rng = np.random.default_rng(100 + k)
    n = 200_000 if k >= 9 else 4 ** k
    hi = 4 ** k
    u = np.unique(rng.integers(0, hi, size=n, dtype=np.uint64)).astype(O.get_hash_dtype(k))
    c = rng.integers(1, 1000, size=len(u)).astype(O.get_cnt_dtype(k))
    cons = int(u[len(u) // 3])
    if cons > int(O.revcom_hash(cons, k)):
        cons = int(O.revcom_hash(cons, k))
    for r, rc in ((0, True), (max(1, k // 3), True), (k // 2, False), (k, True)):
        gu, gc, gm = R._hamball_extract(u, c, k, cons, r, rc)
        ou, oc = O.ex_hamball(u, c, k, cons, r, rc)
        np.testing.assert_array_equal(gu, ou)
        np.testing.assert_array_equal(gc, oc)
        np.testing.assert_array_equal(gm, O.cal_cnt_mat(ou, oc, k))
    assert len(R._hamball_extract(u[:0], c[:0], k, cons, 1, True)[0]) == 0


@pytest.mark.parametrize("k", [8, 15, 16, 21])
def test_hamball_extract_from_the_resident_table(R, O, k):
    """scan_motif's own Hamming-ball call reads the table a counts handle still holds in HBM instead of k{k}.pkl: same members,
    same order, same dtypes as the extraction from the host arrays (and as the oracle)"""
    import ctypes as C
    from kmap_amd import _ffi
    from kmap_amd._ffi import check, ptr
    from kmap_amd.kmer_count import DeviceCounts
    rng = np.random.default_rng(300 + k)
    u = np.unique(rng.integers(0, 4 ** k, size=150_000, dtype=np.uint64)).astype(O.get_hash_dtype(k))
    c = rng.integers(1, 1000, size=len(u)).astype(O.get_cnt_dtype(k))
    cons = int(u[len(u) // 2])
    if cons > int(O.revcom_hash(cons, k)):
        cons = int(O.revcom_hash(cons, k))
    dc = DeviceCounts()
    try:
        check(_ffi.lib().kmap_counts_load(dc._h, ptr(u), ptr(c), len(u), k))
        dc.k, dc.n_uniq = k, len(u)
        for r, rc in ((0, True), (max(1, k // 3), True), (k // 2, False), (k, True)):
            hu, hc, _ = R._hamball_extract(u, c, k, cons, r, rc, want_mat=False)
            du, dcnt = R._hamball_extract_resident(dc, k, cons, r, rc)
            assert du.dtype == hu.dtype and dcnt.dtype == hc.dtype
            np.testing.assert_array_equal(du, hu)
            np.testing.assert_array_equal(dcnt, hc)
            ou, oc = O.ex_hamball(u, c, k, cons, r, rc)
            np.testing.assert_array_equal(du, ou)
            np.testing.assert_array_equal(dcnt, oc)
    finally:
        dc.close()


def test_pos_density_golden(R, golden):
    g = golden("report.npz")
    x_arr = np.arange(0, 1.01, 0.01)
    np.testing.assert_array_equal(x_arr, g["x_arr"])
    occ = R.Occurrence.from_file(RGOLD / "synth4.motif_occurence.csv", 4)
    for i, name in enumerate(g["s4_names"]):
        for src in (occ, RGOLD / "synth4.motif_occurence.csv"):       # in-memory hit list and the reference's file signature
            n_seq, n_occ, d = R.get_motif_pos_density(src, i, len(str(name)), x_step=0.01, x_arr=x_arr)
            assert [n_seq, n_occ] == list(g[f"s4_dens_{i}_n"])
            np.testing.assert_allclose(d, g[f"s4_dens_{i}"], rtol=1e-12, atol=1e-300)   # f64; summation order over reads differs
    for i, name in enumerate(g["final_conseq"]):                       # default grid (np.arange(0, 1, x_step)) on tests/test.fa
        n_seq, n_occ, d = R.get_motif_pos_density(RGOLD / "final.motif_occurence.csv", i, len(str(name)))
        assert [n_seq, n_occ] == list(g[f"dens_default_{i}_n"])
        np.testing.assert_allclose(d, g[f"dens_default_{i}"], rtol=1e-12, atol=1e-300)


def test_pos_density_random_vs_oracle(R, O):
    rng = np.random.default_rng(9)
    n_seq = 5000
    seq_len = rng.integers(40, 500, size=n_seq).astype(np.int64)
    hits = np.where(rng.random(n_seq) < 0.6, rng.integers(1, 21, size=n_seq), 0).astype(np.int32)
    pos = np.concatenate([np.sort(rng.integers(0, seq_len[r] - 10, size=hits[r])) for r in range(n_seq)]).astype(np.int32)
    occ = R.Occurrence([hits], [pos], seq_len)
    x_arr = np.arange(0, 1.005, 0.005)                                  # 201 grid points: more than one pass per thread
    n, m, d = R.get_motif_pos_density(occ, 0, 11, x_step=0.005, x_arr=x_arr)
    offs = occ.offs(0)
    rows = [(pos[offs[r]:offs[r + 1]].tolist(), seq_len[r]) for r in range(n_seq) if hits[r]]
    np.testing.assert_allclose(d, O.motif_pos_density(rows, 11, x_arr, 0.005), rtol=1e-12, atol=1e-300)
    assert n == len(rows) and m == int(hits.sum())
    # no hits at all -> zeros; mismatching offsets are rejected before any launch
    z = R.Occurrence([np.zeros(7, np.int32)], [np.zeros(0, np.int32)], np.full(7, 100))
    assert not R.get_motif_pos_density(z, 0, 8)[2].any()
    bad = R.Occurrence([hits], [pos], seq_len)
    bad._offs[0] = offs + np.arange(len(offs))
    with pytest.raises(ValueError):
        R.get_motif_pos_density(bad, 0, 11)


def test_scan_motif_report_branches(R, golden, tmp_path):
    """preproc + scan_motif on tests/test.fa with the three report flags on == the reference's data files"""
    from kmap_amd._toml import dump_toml, load_toml
    from kmap_amd.kmer_count import _preproc
    from kmap_amd.motif_discovery import _scan_motif
    g = golden("report.npz")
    fa, res = tmp_path / "test.fa", tmp_path / "res"
    shutil.copyfile(GOLD / "test.fa", fa)
    res.mkdir()
    cfg = load_toml(RGOLD / "config.toml")
    assert cfg["motif_discovery"]["motif_pos_density_flag"] and cfg["motif_discovery"]["gen_hamball_flag"]
    cfg["general"]["input_fasta_file"], cfg["general"]["res_dir"] = str(fa), str(res)
    dump_toml(cfg, res / "config.toml")
    _preproc(str(fa), str(res))
    np.random.seed(123)
    _scan_motif(str(res))
    assert (res / "final.motif_occurence.csv").read_text() == (RGOLD / "final.motif_occurence.csv").read_text()
    with open(res / "motif_pos_density.np.pkl", "rb") as fh:
        x_arr, dens = pickle.load(fh)
    np.testing.assert_array_equal(x_arr, g["x_arr"])
    np.testing.assert_allclose(dens, g["density"], rtol=1e-12, atol=1e-300)
    for f in ("co_occurence_mat.tsv", "co_occurence_mat.norm.tsv", "co_occurence_motif_dist_mat.tsv",
              "co_occurence_motif_dist_data.txt"):
        assert (res / "co_occurence" / f).read_text() == (RGOLD / f).read_text(), f
    mats = sorted(RGOLD.glob("cntmat_*.csv"))
    assert len(mats) == 2
    for f in mats:
        assert (res / "hamming_balls" / f.name).read_text() == f.read_text(), f.name


def test_user_motif_occurence_file(golden, tmp_path):
    """get_user_motif_occurence_file (occurrence scan with user-given radii) on tests/test.fa == the reference's file (incl. the
    per-length radius override: the last consensus of a length decides)"""
    from kmap_amd import motif_discovery as MD
    np.random.seed(321)
    MD.get_user_motif_occurence_file(GOLD / "test.fa", ["AATCGATAGC", "CCTACGTA", "GGGGGGGG"], [3, 1, 2],
                                     tmp_path / "u.csv", True)
    assert (tmp_path / "u.csv").read_text() == (RGOLD / "user_motif_occurence.csv").read_text()
