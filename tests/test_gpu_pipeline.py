"""GPU end-to-end parity on the reference's own tests/test.fa (config C1): preproc -> scan_motif ->
visualize_kmers through kmap_amd must reproduce the files the reference wrote (tests/golden/scan_testfa*)."""
import pickle
import shutil
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).resolve().parent / "golden"


def _lines(p):
    with open(p) as fh:
        return fh.read().splitlines()


def _run_c1(tmp):
    """preproc + scan_motif of tests/test.fa with the reference run's config, seeded like gen_golden.py"""
    from kmap_amd._toml import dump_toml, load_toml
    from kmap_amd.kmer_count import _preproc
    from kmap_amd.motif_discovery import _scan_motif
    fa = tmp / "test.fa"
    shutil.copyfile(GOLD / "test.fa", fa)
    res = tmp / "res"
    res.mkdir()
    cfg = load_toml(GOLD / "scan_testfa" / "config.toml")          # the config the reference run used
    cfg["general"]["input_fasta_file"] = str(fa)
    cfg["general"]["res_dir"] = str(res)
    dump_toml(cfg, res / "config.toml")
    _preproc(str(fa), str(res))
    np.random.seed(123)                                              # gen_golden.py seeds the same way
    _scan_motif(str(res))
    return res


@pytest.fixture(scope="module")
def run_dir(tmp_path_factory):
    return _run_c1(tmp_path_factory.mktemp("c1"))


def test_preproc_arrays(run_dir, golden):
    s = golden("scan_testfa.npz")
    with open(run_dir / "input.bin.pkl", "rb") as fh:
        seq = pickle.load(fh)
    with open(run_dir / "input.seqboarder.bin.pkl", "rb") as fh:
        borders = pickle.load(fh)
    assert seq.dtype == np.uint8 and borders.dtype == s["borders"].dtype
    np.testing.assert_array_equal(seq, s["seq"])
    np.testing.assert_array_equal(borders, s["borders"])
    assert _lines(run_dir / "motif_def_table.csv") == _lines(GOLD / "scan_testfa" / "motif_def_table.csv")


@pytest.mark.parametrize("k", range(6, 13))
def test_kmer_count_pickles(run_dir, golden, k):
    s = golden("scan_testfa.npz")
    with open(run_dir / "kmer_count" / f"k{k}.pkl", "rb") as fh:
        kk, u, c = pickle.load(fh)
    assert kk == k and u.dtype == s[f"k{k}_uniq"].dtype and c.dtype == s[f"k{k}_cnt"].dtype
    np.testing.assert_array_equal(u, s[f"k{k}_uniq"])
    np.testing.assert_array_equal(c, s[f"k{k}_cnt"])


@pytest.mark.parametrize("name", ["candidate_conseq.csv", "final_conseq.txt", "final_conseq.info.csv",
                                  "final.motif_occurence.csv", "sample_kmers.tsv"])
def test_text_outputs_identical(run_dir, name):
    assert _lines(run_dir / name) == _lines(GOLD / "scan_testfa" / name)


@pytest.mark.parametrize("k", [8, 10])
def test_per_k_occurrence_files(run_dir, k):
    assert _lines(run_dir / "kmer_count" / f"k{k}.motif_occurence.csv") == _lines(GOLD / "scan_testfa" / f"k{k}.motif_occurence.csv")


def test_readme_two_final_motifs(run_dir):
    assert len(_lines(run_dir / "final_conseq.txt")) == 2          # reference README.md:105


def test_sample_and_matrix_pickles(run_dir, golden):
    s = golden("scan_testfa.npz")
    with open(run_dir / "sample_kmers.pkl", "rb") as fh:
        kh, cnts, lab, conseqs = pickle.load(fh)
    np.testing.assert_array_equal(kh, s["samp_kh"])
    np.testing.assert_array_equal(cnts, s["samp_cnts"])
    np.testing.assert_array_equal(lab, s["samp_label"])
    assert conseqs == [str(c) for c in s["samp_conseqs"]]
    with open(run_dir / "sample_kmer_hamdist_mat.pkl", "rb") as fh:
        klen, mat, labels = pickle.load(fh)
    assert klen == int(s["hamdist_kmer_len"]) and mat.dtype == np.int64
    np.testing.assert_array_equal(mat, s["hamdist_mat_u8"])
    np.testing.assert_array_equal(labels, s["hamdist_label"])


def test_find_motif_dropin_signature(run_dir, golden, motif_defs):
    """find_motif with the reference's numpy-in signature (mutates its argument like the reference)."""
    from kmap_amd.motif_discovery import find_motif
    s, f = golden("scan_testfa.npz"), golden("find_motif_testfa.npz")
    for k, rep in ((8, 0), (10, 0), (8, 1)):
        d = motif_defs[k]
        seq = s["seq"].copy()
        r = find_motif(seq, k, d.max_ham_dist, d.p_uniform, d.ratio_mu, d.ratio_std, d.ratio_cutoff, top_k=5, n_trial=10,
                       merge_revcom_mode=True, rep_mode=bool(rep), save_kmer_cnt_flag=False, kmer_cnt_pkl_file=None,
                       boarder_pkl_file=run_dir / "input.seqboarder.bin.pkl")
        tag = f"k{k}_rep{rep}"
        assert [int(x) for x in r] == [int(x) for x in f[f"{tag}_kh"]]
        np.testing.assert_allclose(np.array([list(v) for v in r.values()]).reshape(-1, 3), f[f"{tag}_vals"], rtol=1e-12)
        if len(r):
            assert not np.array_equal(seq, s["seq"])                # masked in place


def test_find_motif_device_topk_path(run_dir, golden, motif_defs, monkeypatch):
    """Above TOPK_DEVICE_MIN unique k-mers the candidates come from the device top-k; on tests/test.fa (distinct top
    counts) it must find the same motifs as the fetch + np.argpartition path."""
    from kmap_amd import motif_discovery as MD
    s = golden("scan_testfa.npz")
    bpk = run_dir / "input.seqboarder.bin.pkl"
    outs = []
    for thr in (MD.TOPK_DEVICE_MIN, 0):
        monkeypatch.setattr(MD, "TOPK_DEVICE_MIN", thr)
        d = motif_defs[11]
        outs.append(MD.find_motif(s["seq"].copy(), 11, d.max_ham_dist, d.p_uniform, d.ratio_mu, d.ratio_std, d.ratio_cutoff,
                                  save_kmer_cnt_flag=False, boarder_pkl_file=bpk))
    assert list(outs[0].keys()) == list(outs[1].keys()) and len(outs[0]) >= 1
    for kh in outs[0]:
        assert outs[0][kh] == outs[1][kh]
    # KMAP_EXACT=1 (or general.exact in config.toml): np.argpartition on the fetched table whatever the threshold says
    from kmap_amd import _policy
    from kmap_amd.kmer_count import DeviceCounts

    def no_topk(self, top_k):
        raise AssertionError("device top-k used in exact mode")
    for how in ("env", "config"):
        with monkeypatch.context() as mp:
            mp.setattr(DeviceCounts, "topk", no_topk)
            if how == "env":
                mp.setenv("KMAP_EXACT", "1")
            else:
                _policy.apply_config({"general": {"exact": True}})
            try:
                r = MD.find_motif(s["seq"].copy(), 11, d.max_ham_dist, d.p_uniform, d.ratio_mu, d.ratio_std, d.ratio_cutoff,
                                  save_kmer_cnt_flag=False, boarder_pkl_file=bpk)
            finally:
                _policy.reset()
        assert list(r.keys()) == list(outs[0].keys()) and all(r[kh] == outs[0][kh] for kh in r)
    # the large-table path writes k{k}.pkl from a background thread (protocol 5) while the trials run: same content
    pkl = run_dir.parent / "bg_k11.pkl"
    if pkl.exists():
        pkl.unlink()
    d = motif_defs[11]
    r = MD.find_motif(s["seq"].copy(), 11, d.max_ham_dist, d.p_uniform, d.ratio_mu, d.ratio_std, d.ratio_cutoff,
                      save_kmer_cnt_flag=True, kmer_cnt_pkl_file=pkl, boarder_pkl_file=bpk)
    assert list(r.keys()) == list(outs[0].keys())
    with open(pkl, "rb") as fh:
        kk, u, c = pickle.load(fh)
    assert kk == 11
    np.testing.assert_array_equal(u, s["k11_uniq"])
    np.testing.assert_array_equal(c, s["k11_cnt"])
    assert u.dtype == s["k11_uniq"].dtype and c.dtype == s["k11_cnt"].dtype


def test_occurrence_scan_vs_oracle_with_subsample(motif_defs):
    """Low-complexity reads with > 20 hits at the minimum distance exercise the reference's random subsample
    (np.random.choice order) -- compared with the oracle's restatement under the same seed."""
    from kmap_amd.motif_discovery import DeviceSeq, gen_motif_occurence_file
    from oracle import oracle as O
    import tempfile
    rng = np.random.default_rng(8)
    reads = ["A" * 60, "ACGT" * 12, "".join(rng.choice(list("ACGT"), 80)), "AAAAAAAC" * 6, "ACG", "", "TTTTTTTTTTTTTTTTTTTTTTTTTTTTTT",
             "".join(rng.choice(list("ACGTN"), 70))]
    arrs = [O.dna2arr(r) for r in reads]
    seq = np.concatenate(arrs)
    lens = np.array([len(a) for a in arrs])
    st = np.concatenate([[0], np.cumsum(lens)[:-1]])
    borders = np.stack([st, st + lens - 1], 1).astype(np.int64)
    conseqs = ["AAAAAA", "ACGTACGT", "TTTTTTTTTTTT"]
    r_of = {k: d.max_ham_dist for k, d in motif_defs.items()}
    with tempfile.TemporaryDirectory() as td:
        ds = DeviceSeq(seq, borders)
        np.random.seed(77)
        gen_motif_occurence_file(conseqs, motif_defs, None, Path(td) / "occ.csv", True, dev_seq=ds)
        ds.close()
        got = _lines(Path(td) / "occ.csv")
    np.random.seed(77)
    want = O.motif_occurence_lines(seq, borders, conseqs, r_of, True, np.random)
    assert got == want


def test_c5_shape_scan_and_count_vs_oracle(motif_defs):
    """Config C5 shape at test size: k = 14, max_ham_dist = 5 ball scan over 300 bp reads + counting at k = 14 / 16."""
    from kmap_amd import synth
    from kmap_amd.kmer_count import DeviceCounts, kmer2hash
    from kmap_amd.motif_discovery import DeviceSeq
    from oracle import oracle as O
    import ctypes as C
    seq, borders = synth.synth_reads(3000, 300, 3, motifs=("AGGACCTACGTACA", "AATCGATAGC"))
    seq[::997] = 255                                                     # a few N bases inside reads
    ds = DeviceSeq(seq, borders)
    hits, pos = ds.scan(14, kmer2hash("AGGACCTACGTACA"), motif_defs[14].max_ham_dist, True)
    buf, md, off = np.empty(400, np.int32), C.c_int(0), 0
    for i, (st, en) in enumerate(borders):
        m = O.lib().ko_scan_read(np.ascontiguousarray(seq[st:en]), en - st, 14, int(kmer2hash("AGGACCTACGTACA")), 5, 1, buf, C.byref(md))
        assert hits[i] == m
        np.testing.assert_array_equal(pos[off:off + m], buf[:m])
        off += m
    assert off == len(pos) and np.count_nonzero(hits) > 1000
    dc = DeviceCounts()
    for k in (14, 16):
        ds.count(dc, k, dedupe=True, merge_revcom=True)
        u, c = dc.fetch()
        ou, oc = O.count_kmers(seq, borders, k, rep_mode=False, revcom_mode=True)
        np.testing.assert_array_equal(u, ou)
        np.testing.assert_array_equal(c, oc)
    dc.close()
    ds.close()


def test_visualize_kmers_c1(run_dir, golden):
    """visualize_kmers on the C1 result directory, with the neighbour table of the reference run injected (np.argpartition's
    choice among ties differs between hosts): low_dim_data.tsv equals the golden trace's best snapshot printed with the
    reference's %3.3f format (the file contract is 1e-3 granular)."""
    from kmap_amd._toml import load_toml
    from kmap_amd.visualization import _visualize_kmers
    u = golden("umap_n300.npz")
    cfg = load_toml(run_dir / "config.toml")
    assert cfg["visualization"]["random_seed"] == int(u["seed"]) and cfg["visualization"]["n_max_iter"] == int(u["n_iter"])
    ld = _visualize_kmers(str(run_dir), neighbor_inds_mat=u["nb"])
    with open(run_dir / "sample_kmer_hamdist_mat.pkl", "rb") as fh:
        _, mat, labels = pickle.load(fh)
    rows = _lines(run_dir / "low_dim_data.tsv")
    assert rows[0] == "x\ty\tlabel" and len(rows) == 301
    assert [int(r.split("\t")[2]) for r in rows[1:]] == [int(x) for x in labels]
    np.testing.assert_allclose(ld, u["final"], rtol=0, atol=1e-5)
    want = [f"{x:3.3f}\t{y:3.3f}\t{int(l)}" for x, y, l in zip(u["final"][0], u["final"][1], labels)]
    mism = sum(a != b for a, b in zip(rows[1:], want))
    assert mism <= 3    # a coordinate within 1e-5 of a rounding boundary may print differently


def test_compact_handoff_equals_dense(run_dir, golden, tmp_path, monkeypatch):
    """above DENSE_PKL_MAX_N sampled k-mers scan_motif writes [kmer_len, None, labels] instead of the int64
    N x N matrix and visualize_kmers rebuilds the matrix on the device from sample_kmers.pkl.  With the threshold patched to
    0 the same C1 run must hand over the same labels and produce the same low_dim_data.tsv as the dense pickle (SEQ mode,
    the reference run's neighbour table injected), i.e. the golden embedding."""
    from kmap_amd import motif_discovery as MD
    from kmap_amd.visualization import EMBED_SEQ, _visualize_kmers
    u = golden("umap_n300.npz")
    monkeypatch.setattr(MD, "DENSE_PKL_MAX_N", 0)
    res = _run_c1(tmp_path)
    with open(res / "sample_kmer_hamdist_mat.pkl", "rb") as fh:
        klen, mat, labels = pickle.load(fh)
    with open(run_dir / "sample_kmer_hamdist_mat.pkl", "rb") as fh:
        klen_d, mat_d, labels_d = pickle.load(fh)
    assert mat is None and mat_d is not None and klen == klen_d
    np.testing.assert_array_equal(labels, labels_d)
    assert (res / "sample_kmers.pkl").read_bytes() == (run_dir / "sample_kmers.pkl").read_bytes()
    dense = _visualize_kmers(str(run_dir), mode=EMBED_SEQ, neighbor_inds_mat=u["nb"])
    compact = _visualize_kmers(str(res), mode=EMBED_SEQ, neighbor_inds_mat=u["nb"])
    np.testing.assert_array_equal(compact, dense)          # same device arithmetic from either hand-off
    assert (res / "low_dim_data.tsv").read_text() == (run_dir / "low_dim_data.tsv").read_text()
    np.testing.assert_allclose(compact, u["final"], rtol=0, atol=1e-5)


def test_scan_motif_background_table_savers(run_dir, tmp_path, monkeypatch):
    """Large count tables leave find_motif through a background TableSaver (own stream, pinned staging, parallel pickle writer)
    while the trials go on in a second handle, and sample_disp_kmer labels the table still resident in HBM instead of reading
    k{k}.pkl back.  With the size threshold patched to 0 every k of the C1 run takes that path: all files must equal the
    synchronous run's (the pickles array for array)."""
    from kmap_amd import motif_discovery as MD
    monkeypatch.setattr(MD, "SAVE_ASYNC_MIN", 0)
    res = _run_c1(tmp_path)
    names = sorted(p.relative_to(run_dir) for p in run_dir.rglob("*") if p.is_file() and p.name != "low_dim_data.tsv")
    assert sorted(p.relative_to(res) for p in res.rglob("*") if p.is_file()) == names and len(names) > 20
    for rel in names:
        a, b = run_dir / rel, res / rel
        if rel.name == "config.toml":
            continue
        if rel.suffix == ".pkl":
            with open(a, "rb") as fa, open(b, "rb") as fb:
                xa, xb = pickle.load(fa), pickle.load(fb)
            for u, v in zip(xa, xb):
                if isinstance(u, np.ndarray):
                    assert u.dtype == v.dtype
                np.testing.assert_array_equal(np.asarray(u, dtype=object) if isinstance(u, list) else u,
                                              np.asarray(v, dtype=object) if isinstance(v, list) else v, err_msg=str(rel))
        else:
            assert a.read_bytes() == b.read_bytes(), rel


def test_device_topk_tie_rule(motif_defs, monkeypatch):
    """Tables above TOPK_DEVICE_MIN unique k-mers take their top_k candidates from kmap_counts_topk: largest count first, ties
    by the LOWEST table index.  That is a documented deviation from the reference's np.argpartition(cnt, -top_k)[-top_k:]
    (motif_discovery.py:661), whose choice among equal counts at the top_k boundary is an artefact of numpy's introselect.
    Tie-heavy table: counts 9, 9, then a plateau of 7s crossing the boundary.  Expected: the two 9s and the three lowest-index
    7s.  np.argpartition returns the same multiset of counts but whichever three 7s introselect leaves in the tail."""
    from kmap_amd.kmer_count import DeviceCounts
    from kmap_amd._ffi import check, lib, ptr
    k = 9
    rng = np.random.default_rng(5)
    u = np.sort(rng.choice(4 ** k, size=5000, replace=False)).astype(np.uint32)
    c = rng.integers(1, 6, size=5000).astype(np.int32)
    sevens = np.array([40, 41, 700, 1200, 2500, 2501, 3900, 4999])
    c[sevens] = 7
    c[[1234, 77]] = 9
    dc = DeviceCounts()
    check(lib().kmap_counts_load(dc._h, ptr(u), ptr(c), len(u), k))
    dc.k, dc.n_uniq = k, len(u)
    idx, kh, cnt = dc.topk(5)
    assert list(cnt) == [9, 9, 7, 7, 7]
    assert list(idx) == [77, 1234, 40, 41, 700]
    np.testing.assert_array_equal(kh, u[[77, 1234, 40, 41, 700]])
    ref_pick = set(np.argpartition(c, -5)[-5:].tolist())
    assert {77, 1234} <= ref_pick and len(ref_pick & set(sevens.tolist())) == 3     # same multiset of counts ...
    dc.close()


def _inverse_cdf_reference(uniq, cnt, lab, quota, big_min, k):
    """test-local numpy statement of the sampling rule of sample_disp_kmer given labels: np.random.multinomial over the
    normalised counts for labels with <= big_min members (the reference's draw), inverse CDF by np.searchsorted on the
    cumulative counts for larger labels (the product's documented rule for tables the reference cannot process)"""
    inds, cnts = [], []
    for c, n_draw in enumerate(quota):
        ci = np.where(lab == c)[0]
        ws = cnt[ci]
        if len(ci) > big_min:
            cdf = np.cumsum(ws, dtype=np.float64)
            hits = np.searchsorted(cdf, np.floor(np.random.random_sample(int(n_draw)) * cdf[-1]), side="right")
            sel, t = np.unique(np.minimum(hits, len(ws) - 1), return_counts=True)
            inds.append(ci[sel])
            cnts.append(t)
            continue
        t = np.random.multinomial(n_draw, ws / sum(ws), size=1).squeeze()
        inds.append(ci[t > 0])
        cnts.append(t[t > 0])
    return np.concatenate(inds), np.concatenate(cnts)


def test_sample_disp_kmer_vs_oracle(run_dir, golden, motif_defs, monkeypatch):
    """sample_disp_kmer (table labelled on the device, host draws only):
    (a) tests/test.fa: == the oracle's numpy restatement of the reference (which the pipeline test pins to the reference's
        sample_kmers.pkl), and the whole-table return when more samples are asked for than k-mers exist;
    (b) random tables, low TOPK_DEVICE_MIN: labels / re-orientation == oracle, draws == the inverse-CDF rule above."""
    import pickle
    import warnings
    from kmap_amd import motif_discovery as MD
    from kmap_amd.kmer_count import init_motif_def_dict
    from oracle import oracle as O
    s = golden("scan_testfa.npz")
    conseqs = [str(c) for c in s["samp_conseqs"]]
    k = int(s["hamdist_kmer_len"])
    mdd = init_motif_def_dict(GOLD / "scan_testfa" / "motif_def_table.csv")
    r_of = {kk: d.max_ham_dist for kk, d in mdd.items() if isinstance(kk, int)}
    with open(run_dir / "kmer_count" / f"k{k}.pkl", "rb") as fh:
        _, u, c = pickle.load(fh)
    np.random.seed(123)
    got = MD.sample_disp_kmer(conseqs, k, mdd, run_dir / "kmer_count", n_total_sample=300, n_motif_kmer=150)
    np.random.seed(123)
    want = O.sample_disp_kmer(conseqs, k, r_of, u, c, n_total_sample=300, n_motif_kmer=150)
    for a, b in zip(got[:3], want[:3]):
        np.testing.assert_array_equal(a, b)
        assert a.dtype == b.dtype
    assert got[2].dtype == np.int64 and len(got[0]) > 100 and list(got[3]) == list(want[3])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = MD.sample_disp_kmer(conseqs, k, mdd, run_dir / "kmer_count", n_total_sample=10 ** 9, n_motif_kmer=150)
        want = O.sample_disp_kmer(conseqs, k, r_of, u, c, n_total_sample=10 ** 9, n_motif_kmer=150)
    for a, b in zip(got[:3], want[:3]):
        np.testing.assert_array_equal(a, b)
        assert a.dtype == b.dtype

    rng = np.random.default_rng(77)
    for kk, cons_list in ((10, ["AATCGATAGC", "ACCTACGT"]), (16, ["AACCGGTTAACCGGTA", "ACGTTGCA", "AAGGCCTTAA"])):
        hi = 4 ** kk
        ball = []
        for cs in cons_list:                                   # plant members around every consensus (both strands)
            base = int(MD.kmer2hash(cs)) << (2 * (kk - len(cs)))
            ball += [base ^ int(rng.integers(0, 4)) << (2 * int(rng.integers(0, kk))) for _ in range(400)]
        u = np.unique(np.concatenate([rng.integers(0, hi, size=60_000, dtype=np.uint64), np.array(ball, np.uint64)]))
        u = u.astype(MD.get_hash_dtype(kk))
        c = rng.integers(1, 50, size=len(u)).astype(MD.get_cnt_dtype(kk))
        d = run_dir.parent / f"samp_k{kk}"
        (d / "kc").mkdir(parents=True, exist_ok=True)
        with open(d / "kc" / f"k{kk}.pkl", "wb") as fh:
            pickle.dump([kk, u, c], fh)
        ordered = sorted(cons_list, key=len, reverse=True)
        monkeypatch.setattr(MD, "TOPK_DEVICE_MIN", 500)
        np.random.seed(5)
        got = MD.sample_disp_kmer(ordered, kk, mdd, d / "kc", n_total_sample=4000, n_motif_kmer=2000)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ou, oc, olab, _ = O.sample_disp_kmer(ordered, kk, r_of, u, c, n_total_sample=10 ** 12)   # labels + re-orientation only
        quota = MD._label_quota(np.bincount(olab, weights=oc, minlength=len(ordered) + 1), 4000, 2000)
        np.random.seed(5)
        inds, cnts = _inverse_cdf_reference(ou, oc, olab, quota, 500, kk)
        np.testing.assert_array_equal(got[0], ou[inds])
        np.testing.assert_array_equal(got[1], cnts)
        np.testing.assert_array_equal(got[2], olab[inds])
        assert got[0].dtype == ou.dtype and got[2].dtype == np.int64
        assert got[2].max() == len(cons_list) and int(got[1].sum()) == 4000
        # KMAP_EXACT=1: np.random.multinomial per label at every size (reference motif_discovery.py:912) == the oracle's restatement
        monkeypatch.setenv("KMAP_EXACT", "1")
        np.random.seed(5)
        got = MD.sample_disp_kmer(ordered, kk, mdd, d / "kc", n_total_sample=4000, n_motif_kmer=2000)
        np.random.seed(5)
        want = O.sample_disp_kmer(ordered, kk, r_of, u, c, n_total_sample=4000, n_motif_kmer=2000)
        for a, b in zip(got[:3], want[:3]):
            np.testing.assert_array_equal(a, b)
            assert a.dtype == b.dtype
        monkeypatch.delenv("KMAP_EXACT")


def test_genome_like_input_vs_oracle(motif_defs):
    """A few very long records (chromosome-like, with N runs and low-complexity stretches) among short reads: counting with
    and without per-read dedupe (LDS tables for k <= 10, partitioned histogram + tiled merge for k = 11..15, device atomics
    at k = 16, global hash set for the long reads' dedupe), masking, and the occurrence scan (whole-wave walk of long reads)
    all agree with the oracle."""
    from kmap_amd.kmer_count import DeviceCounts, kmer2hash
    from kmap_amd.motif_discovery import DeviceSeq
    from oracle import oracle as O
    import ctypes as C
    rng = np.random.default_rng(2024)
    lens = [150_000, 60_000, 20_001, 7] + list(rng.integers(30, 400, size=4000))
    parts, borders, st = [], [], 0
    for L in lens:
        rd = rng.integers(0, 4, size=L).astype(np.uint8)
        if L > 10_000:
            rd[L // 3:L // 3 + 5000] = 255                                 # an N run
            rd[L // 2:L // 2 + 3000] = np.tile(np.array([0, 1, 0, 3], np.uint8), 750)   # tandem repeat: many duplicates
            rd[100:2100] = 0                                               # poly-A
        parts += [rd, np.array([255], np.uint8)]
        borders.append((st, st + L))
        st += L + 1
    seq, borders = np.concatenate(parts), np.array(borders, np.int64)
    ds = DeviceSeq(seq, borders)
    dc = DeviceCounts()
    for k in (8, 12, 14, 16):
        for dedupe in (True, False):
            ds.count(dc, k, dedupe=dedupe, merge_revcom=True)
            u, c = dc.fetch()
            ou, oc = O.count_kmers(seq, borders, k, rep_mode=not dedupe, revcom_mode=True)
            np.testing.assert_array_equal(u, ou)
            np.testing.assert_array_equal(c, oc)
    cons = int(kmer2hash("ACATACATACAT"))
    ds.mask(12, np.array([cons]), np.array([2]))
    np.testing.assert_array_equal(ds.download(), O.mask_input(seq.copy(), 12, np.array([cons], np.uint64), np.array([2])))
    hits, pos = ds.scan(12, cons, 2, True)                                 # scans the ORIGINAL (unmasked) reads
    buf, md, off = np.empty(400_000, np.int32), C.c_int(0), 0
    for i, (a, b) in enumerate(borders):
        m = O.lib().ko_scan_read(np.ascontiguousarray(seq[a:b]), b - a, 12, cons, 2, 1, buf, C.byref(md))
        assert hits[i] == m, (i, b - a, hits[i], m)
        np.testing.assert_array_equal(pos[off:off + m], buf[:m])
        off += m
    assert off == len(pos) and hits[0] > 100
    dc.close()
    ds.close()


def test_occurrence_file_with_subsample_golden(tmp_path):
    """gen_motif_occurence_file on reads with > 20 hits: same file as the reference wrote with the same np.random seed"""
    from kmap_amd.kmer_count import init_motif_def_dict, _pkg_file
    from kmap_amd.motif_discovery import gen_motif_occurence_file
    mdd = init_motif_def_dict(_pkg_file("default_motif_def_table.csv"))
    np.random.seed(77)
    gen_motif_occurence_file(["AAAAAAAA", "ACGTACGT", "AACCGGTTAA"], mdd, GOLD / "occ20" / "occ20.fa", tmp_path / "o.csv", True)
    assert (tmp_path / "o.csv").read_text() == (GOLD / "occ20" / "occ20.motif_occurence.csv").read_text()


def test_occurrence_file_lazy_lists_and_background_writer(tmp_path):
    """the same golden through a resident DeviceSeq: hit lists stay in HBM (ScanHits), the > 20-hit draws fetch them, the CSV is
    written by a background thread; and a file without such reads goes out with byte-sized hit counts -- equal to the
    synchronous int32 writer's"""
    from kmap_amd.kmer_count import encode_fasta, init_motif_def_dict, _pkg_file
    from kmap_amd.motif_discovery import DeviceSeq, ScanHits, gen_motif_occurence_file, needs_draws, scan_hit_lists
    mdd = init_motif_def_dict(_pkg_file("default_motif_def_table.csv"))
    arr, borders = encode_fasta(str(GOLD / "occ20" / "occ20.fa"))
    ds = DeviceSeq(arr, borders)
    cons = ["AAAAAAAA", "ACGTACGT", "AACCGGTTAA"]
    writers = []
    np.random.seed(77)
    per = gen_motif_occurence_file(cons, mdd, None, tmp_path / "bg.csv", True, dev_seq=ds, writers=writers)
    assert len(writers) == 1
    writers[0].join()
    assert (tmp_path / "bg.csv").read_text() == (GOLD / "occ20" / "occ20.motif_occurence.csv").read_text()
    assert all(h.max(initial=0) <= 20 for h, _ in per)
    lists = scan_hit_lists(ds, cons[1:], mdd, True)                  # no read with > 20 hits of these two
    assert all(isinstance(r, ScanHits) for r in lists) and not needs_draws(lists)
    stats = [(r.n_reads_hit, r.total) for r in lists]
    gen_motif_occurence_file(cons[1:], mdd, None, tmp_path / "u8.csv", True, dev_seq=ds, writers=writers)     # byte counts, background
    writers[1].join()
    host = [list(r) for r in lists]                                  # int32 fetch of the other copies
    assert stats == [(int(np.count_nonzero(h)), len(p)) for h, p in host]
    from kmap_amd.motif_discovery import write_occurence_file
    write_occurence_file(host, cons[1:], tmp_path / "i32.csv", ds.out_n_seq, ds.out_read_len)
    assert (tmp_path / "u8.csv").read_bytes() == (tmp_path / "i32.csv").read_bytes()
    assert len((tmp_path / "u8.csv").read_text().splitlines()) > 3
    pending = scan_hit_lists(ds, cons[1:2], mdd, True)[0]
    ds.close()
    with pytest.raises(RuntimeError, match="closed before"):         # a list outliving its sequence fails loudly
        pending.host()


def test_background_writer_failure_surfaces_at_join(tmp_path):
    """a CSV writer that cannot open its file fails on its own thread; join() re-raises that on the caller's thread (scan_motif
    joins every writer before it reports success)"""
    from kmap_amd.kmer_count import encode_fasta, init_motif_def_dict, _pkg_file
    from kmap_amd.motif_discovery import DeviceSeq, gen_motif_occurence_file
    mdd = init_motif_def_dict(_pkg_file("default_motif_def_table.csv"))
    arr, borders = encode_fasta(str(GOLD / "occ20" / "occ20.fa"))
    ds = DeviceSeq(arr, borders)
    writers = []
    gen_motif_occurence_file(["ACGTACGT"], mdd, None, tmp_path / "no_such_dir" / "o.csv", True, dev_seq=ds, writers=writers)
    with pytest.raises(ValueError, match="cannot open"):
        writers[0].join()
    ds.close()


def test_scan_motif_deferred_occurrence_path(run_dir, tmp_path, monkeypatch):
    """scan_motif starts a k's occurrence CSV right after that k's find_motif unless a read needs the > 20-hit draw; then the
    file waits for its turn in ascending k (np.random order).  Forcing every k down the deferred path must give the same files."""
    from kmap_amd import motif_discovery as MD
    monkeypatch.setattr(MD, "needs_draws", lambda per: True)
    res = _run_c1(tmp_path)
    names = sorted(p.relative_to(run_dir) for p in run_dir.rglob("*") if p.is_file() and p.name != "low_dim_data.tsv")
    assert sorted(p.relative_to(res) for p in res.rglob("*") if p.is_file()) == names
    for rel in names:
        if rel.suffix in (".csv", ".txt"):
            assert (run_dir / rel).read_bytes() == (res / rel).read_bytes(), rel


def test_scan_motif_from_memory_mapped_input(run_dir, tmp_path, monkeypatch):
    """Large inputs are not unpickled: scan_motif maps input.bin.pkl and uploads from the mapping (read-only array).  With the
    size threshold at 0 the C1 run takes that path too and must write the same files."""
    from kmap_amd import kmer_count as KC
    monkeypatch.setattr(KC, "MAP_PICKLE_MIN_BYTES", 0)
    seen = []
    orig = KC.load_array_pickle

    def spy(path, min_bytes=None, populate=True):
        a = orig(path, min_bytes, populate)
        seen.append((Path(path).name, bool(a.flags.writeable)))
        return a
    from kmap_amd import motif_discovery as MD
    monkeypatch.setattr(MD, "load_array_pickle", spy)
    res = _run_c1(tmp_path)
    assert ("input.bin.pkl", False) in seen                          # the read array came back as a read-only view of the mapping
    names = sorted(p.relative_to(run_dir) for p in run_dir.rglob("*") if p.is_file() and p.name != "low_dim_data.tsv")
    assert sorted(p.relative_to(res) for p in res.rglob("*") if p.is_file()) == names
    for rel in names:
        if rel.suffix in (".csv", ".txt"):
            assert (run_dir / rel).read_bytes() == (res / rel).read_bytes(), rel


def test_second_dataset_repetitive_mode_and_noise_kmers(golden, tmp_path):
    """preproc + scan_motif on a second dataset (tests/golden/scan2: planted motifs, repetitive_mode = true, a noise k-mer
    file masked before counting, k = 6..9) == the files and arrays the reference produced with the same np.random seed"""
    from kmap_amd._toml import dump_toml, load_toml
    from kmap_amd.kmer_count import _preproc
    from kmap_amd.motif_discovery import _scan_motif
    g = golden("scan2.npz")
    src = GOLD / "scan2"
    fa, noise, res = tmp_path / "scan2.fa", tmp_path / "noise_kmers.txt", tmp_path / "res"
    shutil.copyfile(src / "scan2.fa", fa)
    shutil.copyfile(src / "noise_kmers.txt", noise)
    res.mkdir()
    cfg = load_toml(src / "config.toml")
    assert cfg["general"]["repetitive_mode"] is True
    cfg["general"]["input_fasta_file"], cfg["general"]["res_dir"] = str(fa), str(res)
    cfg["motif_discovery"]["noise_kmer_file"] = str(noise)
    dump_toml(cfg, res / "config.toml")
    _preproc(str(fa), str(res))
    np.random.seed(9)
    _scan_motif(str(res))
    for k in range(6, 10):
        with open(res / "kmer_count" / f"k{k}.pkl", "rb") as fh:
            kk, u, c = pickle.load(fh)
        np.testing.assert_array_equal(u, g[f"k{k}_uniq"])
        np.testing.assert_array_equal(c, g[f"k{k}_cnt"])
        assert u.dtype == g[f"k{k}_uniq"].dtype and c.dtype == g[f"k{k}_cnt"].dtype
    for f in ("candidate_conseq.csv", "final_conseq.txt", "final_conseq.info.csv", "final.motif_occurence.csv", "sample_kmers.tsv"):
        assert (res / f).read_text() == (src / f).read_text(), f
    with open(res / "sample_kmers.pkl", "rb") as fh:
        skh, scnt, slab, conseqs = pickle.load(fh)
    np.testing.assert_array_equal(skh, g["samp_kh"])
    np.testing.assert_array_equal(scnt, g["samp_cnts"])
    np.testing.assert_array_equal(slab, g["samp_label"])
    assert list(conseqs) == [str(c) for c in g["samp_conseqs"]]
    with open(res / "sample_kmer_hamdist_mat.pkl", "rb") as fh:
        klen, mat, lab = pickle.load(fh)
    assert klen == int(g["hamdist_kmer_len"]) and mat.dtype == np.int64
    np.testing.assert_array_equal(mat, g["hamdist_mat_u8"])
    np.testing.assert_array_equal(lab, g["hamdist_label"])


def test_exact_run_above_the_thresholds(monkeypatch):
    """config.toml `general.exact = true` (run_e2e mode "exact"): with every scale threshold pulled below the run's sizes -- compact
    hand-off, device top-k, inverse-CDF draws, device neighbour rule would all apply -- both verbs still take the reference's numpy
    calls (np.argpartition :661 / visualization.py:100, np.random.multinomial :912) and its arithmetic (SEQ), so every output file
    equals the run below the thresholds (the regime that IS the reference's path); `visualization.embed_mode = "fast"` (the
    opt-in key) runs to a file of the same shape."""
    import pickle
    import shutil
    from pathlib import Path
    from kmap_amd import motif_discovery as MD
    from kmap_amd.e2e import run_e2e
    monkeypatch.delenv("KMAP_EXACT", raising=False)
    monkeypatch.delenv("KMAP_KNN", raising=False)
    dirs = []
    try:
        base = run_e2e("C1s", "default", keep=True)
        dirs.append(base["res_dir"])
        monkeypatch.setattr(MD, "TOPK_DEVICE_MIN", 50)
        monkeypatch.setattr(MD, "DENSE_PKL_MAX_N", 100)
        exact = run_e2e("C1s", "exact", keep=True)
        dirs.append(exact["res_dir"])
        fast = run_e2e("C1s", "fast", keep=True)
        dirs.append(fast["res_dir"])
        b, e, f = (Path(d) for d in dirs)
        assert base["final_conseq"] == exact["final_conseq"] and len(fast["final_conseq"]) >= 1   # fast: device rules apply (ties may differ)
        for name in ("candidate_conseq.csv", "final_conseq.txt", "low_dim_data.tsv"):
            assert (b / name).read_bytes() == (e / name).read_bytes(), name
        sb, se = (pickle.load(open(d / "sample_kmers.pkl", "rb")) for d in (b, e))
        for x, y in zip(sb[:3], se[:3]):
            np.testing.assert_array_equal(x, y)
        hb, he = (pickle.load(open(d / "sample_kmer_hamdist_mat.pkl", "rb")) for d in (b, e))
        assert hb[1] is not None and he[1] is None                      # dense int64 matrix vs compact hand-off
        np.testing.assert_array_equal(hb[2], he[2])
        tb, tf = (np.loadtxt(d / "low_dim_data.tsv", skiprows=1) for d in (b, f))
        assert tb.shape == tf.shape and np.isfinite(tf).all()
    finally:
        for d in dirs:
            shutil.rmtree(d, ignore_errors=True)
