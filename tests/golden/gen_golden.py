#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE ITSELF.

Authoring-container only (needs /root/reference; the GPU box never runs this).
The reference (chengl7-lab/kmap, pure Python + Taichi kernels) is imported from
/root/reference/src with the stand-in modules of tests/golden/refshim/ ahead of it on
sys.path, so that every `@ti.kernel` body of src/kmap/taichi_core.py executes as plain
Python over numpy scalars (IEEE f32/u32/u64 semantics, sequential order, no FMA).
Nothing of the reference's source is copied: only inputs and the outputs it computed
are stored (as .npz / text data files).

    python tests/golden/gen_golden.py            # everything (several minutes)
    python tests/golden/gen_golden.py ops scan   # selected groups

Groups: ops (G2-G5 operator vectors), scan (G1,G6-G8 pipeline on tests/test.fa),
embed (G9,G10 smoothing + umap traces), report (occurrence-file consumers and Hamming-ball
extraction: position density, co-occurrence matrices, count matrices), occ20 (occurrence rows with the
> 20-hit random subsample), earlystop (umap runs that end by the early-stop rule).
"""
import os
import pickle
import shutil
import sys
import tempfile
import warnings
from pathlib import Path

HERE = Path(__file__).resolve().parent
REF = Path("/root/reference")
os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
sys.path.insert(0, str(REF / "src"))
sys.path.insert(0, str(HERE / "refshim"))

import numpy as np  # noqa: E402

warnings.filterwarnings("ignore", category=RuntimeWarning)

import kmap.kmer_count as kc  # noqa: E402
import kmap.motif_discovery as md  # noqa: E402
import kmap.visualization as vz  # noqa: E402

# ---------------------------------------------------------------------------------------
# Taichi does not bounds-check: the reference's hash kernels read up to k-1 bytes past the
# end of the array (taichi_core.py:14-18) and then overwrite the result with the invalid
# hash.  Under Python that read raises, so feed the kernel a padded copy (any pad value
# gives the same output because `st_pos + k > arr_size` already forces invalid).
for _name in ("kmer2hash_kernel_uint32", "kmer2hash_kernel_uint64"):
    _orig = getattr(kc, _name)

    def _padded(arr, arr_size, k, hash_arr, inv, miss, _orig=_orig):
        pad = np.concatenate([np.asarray(arr), np.full(k, 255, dtype=np.uint8)])
        return _orig(pad, arr_size, k, hash_arr, inv, miss)

    setattr(kc, _name, _padded)
md.comp_kmer_hash_taichi = kc.comp_kmer_hash_taichi  # same function object, patched globals

# off-path reporting that needs real Biopython / plotting
md._align_conseq = lambda *a, **k: None
md.plot_cooccurrence_network = lambda *a, **k: None
md.plot_co_occur_motif_locations = lambda *a, **k: None


def save(name, **arrs):
    path = HERE / name
    np.savez_compressed(path, **arrs)
    print(f"  wrote {path.name}: {sum(np.asarray(v).nbytes for v in arrs.values())} raw bytes")


# ---------------------------------------------------------------------------------------
def gen_ops():
    rng = np.random.default_rng(20240101)
    out = {}

    # --- scalar hash <-> kmer (kmap_tests.py:241-266 known answers) ---------------------
    kmers = ["ACTGA", "ACTACTGGAGGACCTACGTAAGCCACGA", "AATCGATAGC", "AGGACCTACGTAC", "TTTTTTTT", "A"]
    out["kmer_strs"] = np.array(kmers)
    out["kmer_hashes"] = np.array([int(kc.kmer2hash(s)) for s in kmers], dtype=np.uint64)
    out["kmer_back"] = np.array([kc.hash2kmer(kc.kmer2hash(s), len(s)) for s in kmers])
    out["kmer_rc_hashes"] = np.array([int(kc.revcom_hash(kc.kmer2hash(s), len(s))) for s in kmers], dtype=np.uint64)

    # --- hashing of a string with N's (kmap_tests.py:173-189) --------------------------
    seq = ("TTTTCGTNCACGACGCTACCTTAAAGCATCCTTCTNTGATACCATAGANNNNNGCAGCTCCTTATCGTTTTAGCTTTCGT"
           "ATTCGTCTAATCGTCTTTTACTCGACGAAAA")
    arr = kc.dna2arr(seq)
    out["nseq_arr"] = arr
    for k in (3, 5, 8, 15, 16, 20, 31):
        h = kc.comp_kmer_hash_taichi(arr, k)
        out[f"nseq_hash_k{k}"] = h
        u, c = kc.count_uniq_hash(h.copy(), k)
        out[f"nseq_uniq_k{k}"] = u
        out[f"nseq_cnt_k{k}"] = c

    # --- revcom arrays ------------------------------------------------------------------
    for k in (3, 4, 8, 15):
        h = rng.integers(0, 4 ** k, size=200, dtype=np.uint64).astype(np.uint32)
        out[f"rc_in_k{k}"] = h
        out[f"rc_out_k{k}"] = kc.get_revcom_hash_arr(h, k)
    for k in (16, 21, 31):
        h = rng.integers(0, 4 ** k, size=200, dtype=np.uint64)
        out[f"rc_in_k{k}"] = h
        out[f"rc_out_k{k}"] = kc.get_revcom_hash_arr(h, k)

    # --- merge_revcom (G3) ---------------------------------------------------------------
    cases = []
    # kmap_tests.py:212-238 explicit k=3 list
    cases.append((3, np.array([0, 2, 10, 11, 17, 18, 19, 23, 27, 33, 36, 38, 41, 43, 46, 51, 53, 57, 59]), None))
    # even k with palindromes present (ACGT=27, AATT=15, TTAA=240 ...), all 4-mers present
    cases.append((4, np.arange(256), rng.integers(1, 50, size=256)))
    # sparse k=4: exercises "replace without re-sorting"
    sel = np.sort(rng.choice(256, size=90, replace=False))
    cases.append((4, sel, rng.integers(1, 1000, size=90)))
    sel = np.sort(rng.choice(4 ** 8, size=5000, replace=False))
    cases.append((8, sel, rng.integers(1, 100, size=5000)))
    sel = np.unique(rng.integers(0, 4 ** 16, size=3000, dtype=np.uint64))
    # add explicit revcom partners for a third of them
    part = np.array([int(kc.revcom_hash(x, 16)) for x in sel[::3]], dtype=np.uint64)
    sel = np.unique(np.concatenate([sel, part]))
    cases.append((16, sel, rng.integers(1, 100, size=len(sel))))
    out["mrc_n"] = np.array(len(cases))
    for i, (k, kh, cnt) in enumerate(cases):
        kh = kh.astype(kc.get_hash_dtype(k))
        cnt = (np.ones(len(kh)) if cnt is None else cnt).astype(kc.get_cnt_dtype(k))
        out[f"mrc{i}_k"] = np.array(k)
        out[f"mrc{i}_in_kh"] = kh.copy()
        out[f"mrc{i}_in_cnt"] = cnt.copy()
        okh, ocnt = kc.merge_revcom(kh.copy(), cnt.copy(), k, keep_lower_hash_flag=True)
        out[f"mrc{i}_out_kh"] = okh
        out[f"mrc{i}_out_cnt"] = ocnt

    # --- Hamming 1-vs-N, head, tail (G4) ------------------------------------------------
    for k, clen in ((8, 6), (12, 9), (15, 15), (16, 11), (20, 16), (31, 17)):
        dt = kc.get_hash_dtype(k)
        h = rng.integers(0, 4 ** k, size=300, dtype=np.uint64).astype(dt)
        h[::37] = kc.get_invalid_hash(dt)  # invalid hashes take part like any value
        cons = dt(rng.integers(0, 4 ** k, dtype=np.uint64))
        out[f"ham_in_k{k}"] = h
        out[f"ham_cons_k{k}"] = np.array(cons)
        out[f"ham_out_k{k}"] = kc.cal_hamming_dist(h, cons, k)
        scons = kc.get_hash_dtype(k)(rng.integers(0, 4 ** clen, dtype=np.uint64))
        out[f"ham_clen_k{k}"] = np.array(clen)
        out[f"ham_scons_k{k}"] = np.array(scons)
        out[f"ham_head_k{k}"] = kc.cal_hamming_dist_head(h, scons, k, clen)
        out[f"ham_tail_k{k}"] = kc.cal_hamming_dist_tail(h, scons, k, clen)

    # --- masking (G5) --------------------------------------------------------------------
    mcases = []
    # poly-T / separator quirk
    a = np.concatenate([kc.dna2arr("ACGTACGTAC"), kc.dna2arr("GGGGGGGGGG")])
    mcases.append((a, 4, np.array([kc.kmer2hash("TTTT")]), np.array([0])))
    # kmap_tests.py:192-209 style
    a = kc.dna2arr(seq)
    mcases.append((a, 5, np.array([kc.kmer2hash(seq[0:5])]), np.array([2])))
    # two consensuses incl. revcom, as find_motif does
    rs = "".join(rng.choice(list("ACGT"), size=400))
    a = np.concatenate([kc.dna2arr(rs[i:i + 50]) for i in range(0, 400, 50)])
    c = kc.kmer2hash(rs[10:18])
    mcases.append((a, 8, np.array([c, kc.revcom_hash(c, 8)]), np.array([2, 2])))
    c = kc.kmer2hash(rs[100:117])
    mcases.append((a, 17, np.array([c, kc.revcom_hash(c, 17)]), np.array([7, 7])))
    # no hit at all (early return)
    mcases.append((kc.dna2arr("ACACACACACACAC"), 6, np.array([kc.kmer2hash("GGGGGG")]), np.array([0])))
    out["mask_n"] = np.array(len(mcases))
    for i, (a, k, ckh, r) in enumerate(mcases):
        out[f"mask{i}_in"] = a.copy()
        out[f"mask{i}_k"] = np.array(k)
        out[f"mask{i}_cons"] = ckh.astype(np.uint64)
        out[f"mask{i}_r"] = r.astype(np.int64)
        out[f"mask{i}_out"] = kc.mask_input(a.copy(), k, ckh, r)
    # mask_ham_ball known answers (kmap_tests.py:268-284)
    mdd = kc.init_motif_def_dict(REF / "src/kmap/default_motif_def_table.csv")
    s1 = "AAAAAAAAAAAAAAAAAAAAAACTAGCTGCCAGTCCCCCCCCCCC"
    r1 = kc.mask_ham_ball(kc.dna2arr(s1)[:-1], mdd, ["AAA", "CCCC"], [0, 0])
    s2 = "AAAAAAAAAAAAAAAAAAAAAACTAGCTGGGGGGGGGGGGGGGGGGGGGGGGGGCCAGTCCCCCCCCCCC"
    r2 = kc.mask_ham_ball(kc.dna2arr(s2)[:-1], mdd, ["AAAAAAA", "CCCCCCCC", "GGGGGGGGG"])
    out["mhb_in1"], out["mhb_out1"] = np.array(s1), np.array(kc.arr2dna(r1))
    out["mhb_in2"], out["mhb_out2"] = np.array(s2), np.array(kc.arr2dna(r2))

    # --- per-read dedupe -----------------------------------------------------------------
    reads = ["ACACACACACAC", "AAAAAAAAAA", "ACGTNACGTACGT", "ACG", "TTTTTTTTTTTTTTTTTTTT", "ACGTTGCAACGTTGCA"]
    a = np.concatenate([kc.dna2arr(r) for r in reads])
    lens = np.array([len(r) + 1 for r in reads])
    st = np.concatenate([[0], np.cumsum(lens)[:-1]])
    borders = np.stack([st, st + lens - 1], axis=1).astype(np.int64)
    out["dd_arr"], out["dd_borders"] = a, borders
    for k in (3, 4, 16):
        h = kc.comp_kmer_hash_taichi(a, k)
        inv = kc.get_invalid_hash(kc.get_hash_dtype(k))
        out[f"dd_hash_k{k}"] = h.copy()
        out[f"dd_out_k{k}"] = kc.remove_duplicate_hash_per_seq(h.copy(), borders, inv)

    # --- consensus merging (kmap_tests.py:614-618) ----------------------------------------
    ex = ["ACGTACGT", "CGTACGT", "TACGTT", "ACGT", "TAC", "CGTA", "ACG", "CCTAGGGG", "CTAGGGG", "TAGGGG", "AGG", "GG"]
    out["mcs_in"] = np.array(ex)
    out["mcs_out"] = np.array(md.merge_consensus_seqs(ex))

    # --- motif definition table ----------------------------------------------------------
    ks = sorted(k for k in mdd if isinstance(k, int))
    out["mdef_k"] = np.array(ks)
    out["mdef_cutoff"] = np.array([mdd[k].ratio_cutoff for k in ks])
    save("ops.npz", **out)


# ---------------------------------------------------------------------------------------
def gen_scan():
    """G1, G2, G6, G7, G8: preproc + scan_motif on the reference's tests/test.fa."""
    fa_src = REF / "tests" / "test.fa"
    shutil.copyfile(fa_src, HERE / "test.fa")  # data fixture of the reference's own tests
    tmp = Path(tempfile.mkdtemp(prefix="kmap_golden_"))
    res = tmp / "res"
    cwd = os.getcwd()
    os.chdir(tmp)
    shutil.copyfile(fa_src, tmp / "test.fa")
    try:
        import tomli
        cfg = kc.read_default_config_file()
        cfg["general"]["input_fasta_file"] = "test.fa"
        cfg["general"]["res_dir"] = "res"
        cfg["kmer_count"]["min_k"] = 6
        cfg["kmer_count"]["max_k"] = 12
        for f in ("motif_pos_density_flag", "motif_co_occurence_flag", "gen_hamball_flag"):
            cfg["motif_discovery"][f] = False
        cfg["motif_discovery"]["n_total_sample"] = 300
        cfg["motif_discovery"]["n_motif_sample"] = 150
        cfg["visualization"]["gen_fig_flag"] = False
        cfg["visualization"]["random_seed"] = 7
        cfg["visualization"]["n_max_iter"] = 60
        res.mkdir()
        import tomli_w
        with open(res / "config.toml", "wb") as fh:
            tomli_w.dump(cfg, fh)
        kc._preproc("test.fa", "res")
        np.random.seed(123)
        md._scan_motif("res")

        out = {}
        with open(res / "input.bin.pkl", "rb") as fh:
            out["seq"] = pickle.load(fh)
        with open(res / "input.seqboarder.bin.pkl", "rb") as fh:
            out["borders"] = pickle.load(fh)
        for k in range(6, 13):
            with open(res / "kmer_count" / f"k{k}.pkl", "rb") as fh:
                kk, u, c = pickle.load(fh)
            assert kk == k
            out[f"k{k}_uniq"], out[f"k{k}_cnt"] = u, c
        with open(res / "sample_kmers.pkl", "rb") as fh:
            skh, scnt, slab, conseqs = pickle.load(fh)
        out["samp_kh"], out["samp_cnts"], out["samp_label"] = skh, scnt, slab
        out["samp_conseqs"] = np.array(conseqs)
        with open(res / "sample_kmer_hamdist_mat.pkl", "rb") as fh:
            klen, mat, lab = pickle.load(fh)
        out["hamdist_kmer_len"] = np.array(klen)
        assert mat.max() < 256 and mat.dtype == np.int64
        out["hamdist_mat_u8"] = mat.astype(np.uint8)  # int64 in the reference; stored narrow
        out["hamdist_label"] = lab
        out["hamdist_uniq_u8"] = md.cal_samp_kmer_hamdist_mat(
            skh, scnt, slab, conseqs, int(klen), uniq_dist_flag=True).astype(np.uint8)
        save("scan_testfa.npz", **out)
        dst = HERE / "scan_testfa"
        dst.mkdir(exist_ok=True)
        for f in ("candidate_conseq.csv", "final_conseq.txt", "final_conseq.info.csv",
                  "final.motif_occurence.csv", "sample_kmers.tsv", "config.toml", "motif_def_table.csv"):
            shutil.copyfile(res / f, dst / f)
        for k in (8, 10):
            shutil.copyfile(res / "kmer_count" / f"k{k}.motif_occurence.csv", dst / f"k{k}.motif_occurence.csv")
        with open(res / "config.toml", "rb") as fh:
            print("  config:", tomli.load(fh)["kmer_count"])

        # G2 extras: counting in both repetitive_mode settings, and with larger k / u64 hashes
        seq, borders = out["seq"], out["borders"]
        g2 = {}
        for k in (6, 8, 9, 14, 16):
            h = kc.comp_kmer_hash_taichi(seq, k)
            inv = kc.get_invalid_hash(kc.get_hash_dtype(k))
            if k in (8, 16):
                g2[f"hash_k{k}"] = h.copy()
            for rep in (True, False):
                hh = h.copy()
                if not rep:
                    hh = kc.remove_duplicate_hash_per_seq(hh, borders, inv)
                u, c = kc.count_uniq_hash(hh, k)
                tag = f"k{k}_rep{int(rep)}"
                g2[f"{tag}_uniq"], g2[f"{tag}_cnt"] = u.copy(), c.copy()
                mu, mc = kc.merge_revcom(u.copy(), c.copy(), k, keep_lower_hash_flag=True)
                g2[f"{tag}_muniq"], g2[f"{tag}_mcnt"] = mu, mc
        save("counts_testfa.npz", **g2)

        # G6: find_motif result dicts for selected k (fresh copies, no pkl caching)
        g6 = {}
        mdd = kc.gen_motif_def_dict(cfg)
        for k, rep in ((8, False), (10, False), (8, True)):
            d = mdd[k]
            r = md.find_motif(seq.copy(), k, d.max_ham_dist, d.p_uniform, d.ratio_mu, d.ratio_std, d.ratio_cutoff,
                              top_k=5, n_trial=10, merge_revcom_mode=True, rep_mode=rep, save_kmer_cnt_flag=False,
                              kmer_cnt_pkl_file=None, boarder_pkl_file=res / "input.seqboarder.bin.pkl")
            tag = f"k{k}_rep{int(rep)}"
            g6[f"{tag}_kh"] = np.array(list(r.keys()), dtype=np.uint64)
            g6[f"{tag}_vals"] = np.array([list(v) for v in r.values()], dtype=np.float64).reshape(-1, 3)
        save("find_motif_testfa.npz", **g6)
    finally:
        os.chdir(cwd)
        shutil.rmtree(tmp, ignore_errors=True)


# ---------------------------------------------------------------------------------------
def gen_embed():
    """G9, G10: smoothing and umap traces on the N=300 matrix of the scan group."""
    z = np.load(HERE / "scan_testfa.npz", allow_pickle=False)
    D = z["hamdist_mat_u8"].astype(np.int64)
    k = int(z["hamdist_kmer_len"])
    n_nb = 20
    out = {"kmer_len": np.array(k), "n_nb": np.array(n_nb)}
    nb = np.argpartition(D, n_nb, axis=1)[:, :n_nb]  # same call as visualization.py:100, same process
    out["nb"] = nb.astype(np.int32)
    S = vz.knn_smooth(D, n_nb)
    assert np.array_equal(np.argpartition(D, n_nb, axis=1)[:, :n_nb], nb)
    out["S"] = S
    T = vz.sigmoid(S, 16.0, change_point=k / 2, scale_factor=0.2 * k - 0.2)
    out["sig"] = T
    out["hd_prob"] = np.exp(-T / 0.5).astype("float32")

    # a small case with a non-Hamming integer matrix (kmap_tests.py:579-612 shape)
    rng = np.random.default_rng(5)
    d2 = rng.integers(0, 100, size=(10, 10))
    d2 = np.triu(d2, 1)
    d2 = d2 + d2.T
    out["small_D"] = d2
    out["small_nb"] = np.argpartition(d2, 4, axis=1)[:, :4].astype(np.int32)
    out["small_S"] = vz.knn_smooth(d2, 4)

    # L3 float operators on random inputs (N=48)
    n = 48
    ld = rng.standard_normal((2, n)).astype("float32")
    im = np.ascontiguousarray(np.array([(i, j) for i in range(n) for j in range(n) if i < j]).T.astype("int32"))
    q = vz.cal_ld_prob_mat_taichi(ld, im)
    p = np.exp(-rng.uniform(0, 16, size=(n, n)) / 0.5).astype("float32")
    p = np.minimum(p, p.T)
    p[0, 1] = p[1, 0] = 0.0  # exercises the eps branch of the cross-entropy kernel
    p[2, 3] = p[3, 2] = 1.0
    out["op_ld"], out["op_q"], out["op_p"] = ld, q, p
    out["op_loss"] = np.array(vz.cross_entropy_taichi(p, q, im), dtype=np.float32)
    out["op_grad"] = vz.gradient_loss_taichi(p, q, ld)
    save("embed_ops.npz", **out)

    # G10: full kmap() traces
    for tag, n_iter, seed, sub in (("n300", 60, 7, None), ("n96", 200, 11, 96)):
        Dm = D if sub is None else D[np.ix_(np.arange(0, 300, 300 // sub)[:sub], np.arange(0, 300, 300 // sub)[:sub])]
        losses, snaps, jit = [], [], []
        o_ce, o_jit = vz.cross_entropy_taichi, vz.add_jitter

        def ce(hd, ldp, im):
            v = o_ce(hd, ldp, im)
            losses.append(v)
            return v

        def aj(ld_data, eps):
            before = ld_data.copy()
            r = o_jit(ld_data, eps)
            jit.append(int(np.count_nonzero(before != r)))
            snaps.append(r.copy())
            return r

        vz.cross_entropy_taichi, vz.add_jitter = ce, aj
        try:
            final = vz.kmap(Dm, k, n_neighbour=n_nb, n_max_iter=n_iter, learning_rate=0.01,
                            n_best_result=10, random_seed=seed, debug=False)
        finally:
            vz.cross_entropy_taichi, vz.add_jitter = o_ce, o_jit
        nbm = np.argpartition(Dm, n_nb, axis=1)[:, :n_nb].astype(np.int32)
        np.random.seed(seed)
        init = np.random.randn(2, len(Dm)).astype("float32")
        print(f"  {tag}: {len(losses)} losses, {sum(jit)} jitter hits, final loss {min(losses)}")
        save(f"umap_{tag}.npz", D=Dm.astype(np.uint8), nb=nbm, kmer_len=np.array(k), seed=np.array(seed),
             n_iter=np.array(n_iter), losses=np.array(losses, dtype=np.float32), init=init,
             coords=np.array(snaps, dtype=np.float32), jitter_hits=np.array(jit), final=final)


def gen_earlystop():
    """umap runs of the reference that END BY THE EARLY-STOP RULE (visualization.py:310-311) before n_max_iter: sub-matrices of
    the N=300 matrix with a large learning rate, so that every pair reaches the q clip and the loss becomes exactly constant.
    The loss steps before the floor are ~1e-5 relative (>> the 1e-7 stop threshold), which makes the stop iteration robust to
    last-bit differences of the loss."""
    z = np.load(HERE / "scan_testfa.npz", allow_pickle=False)
    D = z["hamdist_mat_u8"].astype(np.int64)
    k = int(z["hamdist_kmer_len"])
    n_nb = 20
    out = {"kmer_len": np.array(k), "n_nb": np.array(n_nb)}
    for tag, sub, lr, seed, n_iter in (("a", 32, 16.0, 1, 500), ("b", 48, 16.0, 1, 50)):
        idx = np.arange(0, 300, 300 // sub)[:sub]
        Dm = D[np.ix_(idx, idx)]
        losses = []
        o_ce = vz.cross_entropy_taichi

        def ce(hd, ldp, im):
            v = o_ce(hd, ldp, im)
            losses.append(v)
            return v

        vz.cross_entropy_taichi = ce
        try:
            final = vz.kmap(Dm, k, n_neighbour=n_nb, n_max_iter=n_iter, learning_rate=lr, n_best_result=10, random_seed=seed,
                            debug=False)
        finally:
            vz.cross_entropy_taichi = o_ce
        assert len(losses) < n_iter, "this case is meant to stop early"
        nbm = np.argpartition(Dm, n_nb, axis=1)[:, :n_nb].astype(np.int32)
        print(f"  earlystop {tag}: N={sub} lr={lr} stopped after {len(losses)} of {n_iter} iterations, last losses {losses[-3:]}")
        out.update({f"{tag}_D": Dm.astype(np.uint8), f"{tag}_nb": nbm, f"{tag}_seed": np.array(seed), f"{tag}_lr": np.array(lr),
                    f"{tag}_n_max_iter": np.array(n_iter), f"{tag}_losses": np.array(losses, dtype=np.float32), f"{tag}_final": final})
    save("umap_earlystop.npz", **out)


def _synthetic_occurrence_file(path, rng, n_reads=400, n_motif=4):
    """An occurrence CSV in the reference's format with 4 motifs, multi-hit cells (up to 20 sorted locations), empty
    cells and varying read lengths -- input data for the consumers, written by this script (not by the reference)."""
    names = ["AATCGATAGC", "ACCTACGTA", "GGATCCAAT", "CCGTTAAC"][:n_motif]
    lines = ["seq_ind;" + ";".join(f"motif_{i}_{c}" for i, c in enumerate(names)) + ";seq_len"]
    for r in range(n_reads):
        seq_len = int(rng.integers(60, 400))
        cells, any_hit = [], False
        for m in range(n_motif):
            if rng.random() < (0.55, 0.4, 0.25, 0.05)[m]:
                n_hit = int(rng.choice([1, 1, 1, 2, 3, 4, 7, 20]))
                locs = np.sort(rng.choice(seq_len - len(names[m]) + 1, size=min(n_hit, seq_len - 20), replace=False))
                cells.append(",".join(str(int(v)) for v in locs))
                any_hit = True
            else:
                cells.append("")
        if any_hit:
            lines.append(f"{r};" + ";".join(cells) + f";{seq_len}")
    Path(path).write_text("\n".join(lines) + "\n")
    return names


def gen_report():
    """get_motif_pos_density, get_motif_co_occurence_mat (+ their writers), ex_hamball_kh_arr,
    cal_cnt_mat, _ex_hamball -- outputs of the reference on tests/test.fa (flags on) and on a synthetic occurrence file."""
    for name in ("_draw_motif_pos_density", "_draw_motif_pos_density_all", "draw_motif_distance_distribution", "_draw_logo"):
        setattr(md, name, lambda *a, **k: None)   # plotting only
    fa_src = REF / "tests" / "test.fa"
    tmp = Path(tempfile.mkdtemp(prefix="kmap_golden_"))
    res = tmp / "res"
    cwd = os.getcwd()
    os.chdir(tmp)
    shutil.copyfile(fa_src, tmp / "test.fa")
    dst = HERE / "report_testfa"
    dst.mkdir(exist_ok=True)
    try:
        cfg = kc.read_default_config_file()
        cfg["general"]["input_fasta_file"] = "test.fa"
        cfg["general"]["res_dir"] = "res"
        cfg["kmer_count"]["min_k"] = 6
        cfg["kmer_count"]["max_k"] = 12
        for f in ("motif_pos_density_flag", "motif_co_occurence_flag", "gen_hamball_flag"):
            cfg["motif_discovery"][f] = True
        cfg["motif_discovery"]["n_total_sample"] = 300
        cfg["motif_discovery"]["n_motif_sample"] = 150
        cfg["visualization"]["gen_fig_flag"] = False
        res.mkdir()
        import tomli_w
        with open(res / "config.toml", "wb") as fh:
            tomli_w.dump(cfg, fh)
        kc._preproc("test.fa", "res")
        np.random.seed(123)
        md._scan_motif("res")
        shutil.copyfile(res / "config.toml", dst / "config.toml")
        with open(res / "motif_pos_density.np.pkl", "rb") as fh:
            x_arr, dens = pickle.load(fh)
        out = {"x_arr": x_arr, "density": dens}
        finals = (res / "final_conseq.txt").read_text().split()
        out["final_conseq"] = np.array(finals)
        for i, c in enumerate(finals):
            n_seq, n_occ, d = md.get_motif_pos_density(res / "final.motif_occurence.csv", i, len(c))   # default grid
            out[f"dens_default_{i}"], out[f"dens_default_{i}_n"] = d, np.array([n_seq, n_occ])
        for f in ("co_occurence_mat.tsv", "co_occurence_mat.norm.tsv", "co_occurence_motif_dist_mat.tsv",
                  "co_occurence_motif_dist_data.txt"):
            shutil.copyfile(res / "co_occurence" / f, dst / f)
        for f in sorted((res / "hamming_balls").glob("cntmat_*.csv")):
            shutil.copyfile(f, dst / f.name)
        # ex_hamball return types + arrays, default radius (-1) and an explicit one, revcom on/off
        c0 = finals[0]
        for rt in ("hash", "kmer", "matrix"):
            md._ex_hamball("res", c0, rt, str(dst / f"exhamball_{rt}.txt"), max_ham_dist=1)
        for tag, (cs, r, rc) in {"a": (finals[0], -1, True), "b": (finals[1], 3, True), "c": (finals[1], 2, False),
                                 "d": ("ACGTAC", 1, True)}.items():
            u, c = md.ex_hamball_kh_arr("res", cs, r, str(res / "motif_def_table.csv"), rc)
            out[f"ball_{tag}_kh"], out[f"ball_{tag}_cnt"] = u, c
            out[f"ball_{tag}_mat"] = md.cal_cnt_mat(u, c, len(cs))
            out[f"ball_{tag}_def"] = np.array([cs, str(r), str(int(rc))])
        for k in {len(finals[0]), len(finals[1]), 6}:
            with open(res / "kmer_count" / f"k{k}.pkl", "rb") as fh:
                kk, u, c = pickle.load(fh)
            out[f"k{k}_uniq"], out[f"k{k}_cnt"] = u, c
        shutil.copyfile(res / "final.motif_occurence.csv", dst / "final.motif_occurence.csv")
        shutil.copyfile(res / "motif_def_table.csv", dst / "motif_def_table.csv")

        # synthetic 4-motif occurrence file through the reference's consumers
        names = _synthetic_occurrence_file(dst / "synth4.motif_occurence.csv", np.random.default_rng(5))
        occ = dst / "synth4.motif_occurence.csv"
        co, dist, dd = md.get_motif_co_occurence_mat(occ, len(names))
        out["s4_names"], out["s4_co"], out["s4_dist"] = np.array(names), co, dist
        for (i, j), v in dd.items():
            out[f"s4_dd_{i}_{j}"] = np.array(v, dtype=np.float64)
        co_sum = np.diag(co) + np.diag(co).reshape((-1, 1))
        md.write_co_occurence_mat(dst / "s4_co_occurence_mat.tsv", co + 0.0, names)
        md.write_co_occurence_mat(dst / "s4_co_occurence_mat.norm.tsv", 2 * co / co_sum, names)
        md.write_co_occurence_mat(dst / "s4_co_occurence_motif_dist_mat.tsv", dist, names)
        md.write_co_occurence_dist_arr(dst / "s4_co_occurence_motif_dist_data.txt", dd, names)
        x_step = 0.01
        xa = np.arange(0, 1.0 + x_step, x_step)
        for i, c in enumerate(names):
            n_seq, n_occ, d = md.get_motif_pos_density(occ, i, len(c), x_step=x_step, x_arr=xa)
            out[f"s4_dens_{i}"], out[f"s4_dens_{i}_n"] = d, np.array([n_seq, n_occ])
            out[f"s4_seqnum_{i}"] = np.array(md.get_motif_seq_num(occ, i))
        # user-given motifs / radii on tests/test.fa (get_user_motif_occurence_file; same-length radius override quirk)
        np.random.seed(321)
        md.get_user_motif_occurence_file(Path("test.fa"), ["AATCGATAGC", "CCTACGTA", "GGGGGGGG"], [3, 1, 2],
                                         dst / "user_motif_occurence.csv", True)
        save("report.npz", **out)
    finally:
        os.chdir(cwd)
        shutil.rmtree(tmp, ignore_errors=True)


def gen_scan2():
    """A second end-to-end run of the reference: synthetic reads with two planted motifs, repetitive_mode = true (no per-read
    dedupe), a noise k-mer file (masked before counting), k = 6..9.  FASTA / noise file written by this script."""
    rng = np.random.default_rng(42)
    dst = HERE / "scan2"
    dst.mkdir(exist_ok=True)
    motifs = ["AATCGATAGC", "CCTACGTA"]
    recs = []
    for i in range(700):
        L = int(rng.integers(40, 90))
        seq = rng.choice(list("ACGT"), size=L)
        if i % 5 < 2:
            m = list(motifs[i % 2])
            if rng.random() < 0.3:
                m[int(rng.integers(0, len(m)))] = rng.choice(list("ACGT"))
            a = int(rng.integers(0, L - len(m)))
            seq[a:a + len(m)] = m
        if i % 11 == 0:
            a = int(rng.integers(0, L - 16))
            seq[a:a + 16] = list("ACACACACACACACAC")          # low-complexity noise, listed in the noise k-mer file
        if i % 37 == 0:
            seq[int(rng.integers(0, L))] = "N"
        recs.append(f">s{i}\n" + "".join(seq) + "\n")
    (dst / "scan2.fa").write_text("".join(recs))
    (dst / "noise_kmers.txt").write_text("ACACACAC\nCACACACA\n")
    tmp = Path(tempfile.mkdtemp(prefix="kmap_golden_"))
    res = tmp / "res"
    cwd = os.getcwd()
    os.chdir(tmp)
    shutil.copyfile(dst / "scan2.fa", tmp / "scan2.fa")
    shutil.copyfile(dst / "noise_kmers.txt", tmp / "noise_kmers.txt")
    try:
        cfg = kc.read_default_config_file()
        cfg["general"]["input_fasta_file"] = "scan2.fa"
        cfg["general"]["res_dir"] = "res"
        cfg["general"]["repetitive_mode"] = True
        cfg["kmer_count"]["min_k"] = 6
        cfg["kmer_count"]["max_k"] = 9
        cfg["motif_discovery"]["noise_kmer_file"] = "noise_kmers.txt"
        for f in ("motif_pos_density_flag", "motif_co_occurence_flag", "gen_hamball_flag"):
            cfg["motif_discovery"][f] = False
        cfg["motif_discovery"]["n_total_sample"] = 200
        cfg["motif_discovery"]["n_motif_sample"] = 100
        cfg["visualization"]["gen_fig_flag"] = False
        res.mkdir()
        import tomli_w
        with open(res / "config.toml", "wb") as fh:
            tomli_w.dump(cfg, fh)
        kc._preproc("scan2.fa", "res")
        np.random.seed(9)
        md._scan_motif("res")
        out = {}
        for k in range(6, 10):
            with open(res / "kmer_count" / f"k{k}.pkl", "rb") as fh:
                kk, u, c = pickle.load(fh)
            out[f"k{k}_uniq"], out[f"k{k}_cnt"] = u, c
        with open(res / "sample_kmers.pkl", "rb") as fh:
            skh, scnt, slab, conseqs = pickle.load(fh)
        out["samp_kh"], out["samp_cnts"], out["samp_label"], out["samp_conseqs"] = skh, scnt, slab, np.array(conseqs)
        with open(res / "sample_kmer_hamdist_mat.pkl", "rb") as fh:
            klen, mat, lab = pickle.load(fh)
        out["hamdist_kmer_len"], out["hamdist_mat_u8"], out["hamdist_label"] = np.array(klen), mat.astype(np.uint8), lab
        save("scan2.npz", **out)
        for f in ("candidate_conseq.csv", "final_conseq.txt", "final_conseq.info.csv", "final.motif_occurence.csv",
                  "sample_kmers.tsv", "config.toml", "motif_def_table.csv"):
            shutil.copyfile(res / f, dst / f)
        print("  finals:", (res / "final_conseq.txt").read_text().split())
    finally:
        os.chdir(cwd)
        shutil.rmtree(tmp, ignore_errors=True)


def gen_occ20():
    """Occurrence rows for reads with MORE THAN 20 hits at the minimum distance: the reference then keeps a random 20 of them
    (np.random.choice, motif_discovery.py:1466-1470).  Input FASTA written by this script; output by the reference."""
    rng = np.random.default_rng(20)
    dst = HERE / "occ20"
    dst.mkdir(exist_ok=True)
    recs = []
    for i in range(80):
        L = int(rng.integers(60, 260))
        seq = rng.choice(list("ACGT"), size=L)
        kind = i % 4
        if kind == 0:                                   # long poly-A run: dozens of exact AAAAAAAA hits
            a = int(rng.integers(0, L - 50))
            seq[a:a + int(rng.integers(35, 50))] = "A"
        elif kind == 1:                                 # poly-T run: hits through the reverse complement
            a = int(rng.integers(0, L - 50))
            seq[a:a + int(rng.integers(30, 45))] = "T"
        elif kind == 2:                                 # tandem ACGT repeats + an N
            a = int(rng.integers(0, L - 60))
            seq[a:a + 56] = list("ACGT" * 14)
            seq[int(rng.integers(0, L))] = "N"
        recs.append(f">r{i}\n" + "".join(seq) + "\n")
    (dst / "occ20.fa").write_text("".join(recs))
    cfg = kc.read_default_config_file()
    mdd = kc.gen_motif_def_dict(cfg)
    conseqs = ["AAAAAAAA", "ACGTACGT", "AACCGGTTAA"]
    np.random.seed(77)
    md.gen_motif_occurence_file(conseqs, mdd, dst / "occ20.fa", dst / "occ20.motif_occurence.csv", True)
    n_big = sum(1 for ln in (dst / "occ20.motif_occurence.csv").read_text().splitlines()[1:]
                for cell in ln.split(";")[1:-1] if cell.count(",") == 19)
    print(f"  wrote occ20/: {n_big} cells with exactly 20 (subsampled) locations")
    assert n_big >= 10


if __name__ == "__main__":
    groups = sys.argv[1:] or ["ops", "scan", "embed", "report", "occ20", "scan2", "earlystop"]
    for g in groups:
        print(f"[{g}]")
        {"ops": gen_ops, "scan": gen_scan, "embed": gen_embed, "report": gen_report, "occ20": gen_occ20, "scan2": gen_scan2,
         "earlystop": gen_earlystop}[g]()
