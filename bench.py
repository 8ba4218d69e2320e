#!/usr/bin/env python3
"""bench.py -- headline benchmark of the kmap hot path on MI355X.

Metric (BASELINE.json): Hamming pairs/s of the sampled-k-mer all-pairs matrix at N = 50 k, k = 8 (config C3), plus the
end-to-end scan_motif + visualize_kmers wall time.  One "step" = one pass of the Hamming-matrix kernel over the whole batch of
sampled k-mers: N x N ordered pairs written as uint8 into HBM; hashes/labels are resident in HBM before the timed region.
value = pairs processed by all ranks / max-over-ranks wall time.

The sample is what the C3 pipeline hands over: bench.py runs `scan_motif` on the C3 synthetic reads first (outside the timed
region) and takes sample_kmers.pkl -- expanded by counts, labels 0 / 1 / noise, finals CCTACGTA (8) and ATCGATA (7), so the
rows of the 7-mer's label take the kernel's prefix-compare branch -- not an idealised all-full-length sample.

Multi-GPU (--gpus G, launched by torch.distributed.run, one rank per GPU): the matrix is sharded by row blocks, every rank
holds all N hashes, no data-path collective.  `value` is the weak-scaling Hamming stage: N_total = 50 000 * sqrt(G)
(rounded to 16) so that every GPU keeps 2.5e9 pairs per step.  The same line carries `c4`: BASELINE config C4, N = 200 000 FIXED
(strong scaling): Hamming rows + the row-sharded embedding iteration with its all-reduces.

Extra objects on the JSON line:
  roofline      HIP-event kernel time of the headline kernel (mean / min / median per launch) vs the 8 TB/s HBM peak, algorithmic
                bytes = rows*N + 5N; `stages`: the same for every other stage of the C3 path with SURVEY 8(d)'s algorithmic bytes
  cpu_baseline  the CPU oracle (OpenMP) on a bounded sample of the same workload: Hamming rows, and `e2e` = find_motif +
                smoothing + embedding iterations extrapolated to the C3 job (a reported baseline, not the target)
  e2e           C3 scan_motif + visualize_kmers wall time: k = 6..9 in both embedding modes, and the reference's default k = 6..16
"""
import argparse
import json
import math
import os
import pickle
import statistics
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ceiling ~6290 GB/s
K = 8
N_BASE = 50_000
N_C4 = 200_000


# ---- the sample the C3 pipeline hands over ------------------------------------------------------------------------------
def pipeline_sample(res_dir, n_total):
    """Expanded (hashes uint32, labels int32, consensus lengths) of `n_total` sampled k-mers drawn by sample_disp_kmer from the
    kept C3 result directory (n_total = the run's own n_total_sample re-uses its sample_kmers.pkl)."""
    from kmap_amd._toml import load_toml
    from kmap_amd.kmer_count import gen_motif_def_dict
    from kmap_amd.motif_discovery import sample_disp_kmer
    res = Path(res_dir)
    cfg = load_toml(res / "config.toml")
    if n_total == cfg["motif_discovery"]["n_total_sample"]:
        with open(res / "sample_kmers.pkl", "rb") as fh:
            kh, cnts, lab, conseqs = pickle.load(fh)
    else:
        finals = (res / "final_conseq.txt").read_text().split()
        klen = max(len(c) for c in finals)
        np.random.seed(123)
        kh, cnts, lab, conseqs = sample_disp_kmer(finals, klen, gen_motif_def_dict(cfg), res / "kmer_count", n_total_sample=n_total,
                                                  n_motif_kmer=n_total // 2, revcom_mode=cfg["kmer_count"]["revcom_mode"])
    klen = max(len(c) for c in conseqs)
    assert klen == K, f"the C3 pipeline's longest final consensus is expected to be an {K}-mer, got {conseqs}"
    return (np.repeat(np.asarray(kh), cnts).astype(np.uint32), np.repeat(np.asarray(lab), cnts).astype(np.int32),
            [len(c) for c in conseqs], list(conseqs))


def timed_launches(fn, reps, warmup=2):
    """per-launch HIP-event times (ms) of `fn` on the library's stream"""
    from kmap_amd import _ffi
    for _ in range(warmup):
        fn()
    _ffi.sync()
    evs = [_ffi.Event() for _ in range(reps + 1)]
    evs[0].record()
    for i in range(reps):
        fn()
        evs[i + 1].record()
    _ffi.sync()
    return [evs[i].elapsed_ms(evs[i + 1]) for i in range(reps)]


def roof(bytes_per_launch, ms_list, what, note=None):
    ms = statistics.median(ms_list)
    ach = bytes_per_launch / (ms * 1e-3) / 1e9
    d = {"what": what, "algorithmic_bytes": bytes_per_launch, "ms_median": ms, "ms_min": min(ms_list), "ms_mean": sum(ms_list) / len(ms_list),
         "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS}
    if note:
        d["note"] = note
    return d


def stage_rooflines(reads, kh, lab, conseq_lens):
    """Every stage of the C3 path besides the headline kernel, timed in isolation on resident inputs with HIP events; bytes are
    SURVEY 8(d)'s algorithmic figures (1 B per read position and pass for the read stages -- the uint8 contract; the kernels
    actually read the 0.375 B/position packed stream --, N^2 + 2 N^2 for the smoothing, 2 N^2 per embedding iteration)."""
    import ctypes as C
    from kmap_amd import _ffi, visualization as V
    from kmap_amd.hamdist import hamdist_matrix_dev, pitch_for
    from kmap_amd.kmer_count import DeviceCounts, kmer2hash
    from kmap_amd.motif_discovery import DeviceSeq
    seq, borders = reads
    out = {}
    ds = DeviceSeq(seq, borders)
    dc = DeviceCounts()
    npos = float(len(seq))
    out["count_pass_k8"] = roof(npos + 4 ** 8 * 4, timed_launches(lambda: ds.count(dc, 8, dedupe=False, merge_revcom=True), 6),
                                "histogram count of all 8-mers + revcom merge + compaction (later find_motif rounds)")
    out["count_pass_k8_dedupe"] = roof(npos + 4 ** 8 * 4, timed_launches(lambda: ds.count(dc, 8, dedupe=True, merge_revcom=True), 4),
                                       "first find_motif round: per-read de-duplication fused in front of the histogram")
    out["count_pass_k14"] = roof(npos + 4 ** 14 * 4, timed_launches(lambda: ds.count(dc, 14, dedupe=False, merge_revcom=True), 4),
                                 "partitioned histogram (11 <= k <= 15) + tiled revcom merge")
    cons = int(kmer2hash("CCTACGTA"))
    out["mask_k8"] = roof(npos, timed_launches(lambda: (ds.reset(), ds.mask(8, np.array([cons, cons ^ 0x1B]), np.array([2, 2]))), 6),
                          "mask_input: Hamming-ball flag + cover of two consensuses (incl. the n/8-byte restore)")
    lib = _ffi.lib()
    h = _ffi.vp()
    _ffi.check(lib.kmap_scan_create(C.byref(h)))
    tot = _ffi.i64(0)

    def scan(k, kh_, r):
        _ffi.check(lib.kmap_scan_run_packed_dev(h.value, ds.codes.ptr, ds.inval_orig.ptr, ds.n, ds.borders.ptr, ds.n_seq, k, kh_, r, 1,
                                                C.byref(tot), None))
    out["scan_k8_r2"] = roof(npos, timed_launches(lambda: scan(8, cons, 2), 6),
                             "occurrence scan of one consensus over all reads, device part (incl. its one host sync for the hit total)")
    out["scan_k14_r5"] = roof(npos, timed_launches(lambda: scan(14, int(kmer2hash("AGGACCTACGTACA")), 5), 4),
                              "C5-style ball scan (k = 14, radius 5) on the C3 reads")
    lib.kmap_scan_destroy(h.value)
    dc.close()
    ds.close()
    # ---- sampled k-mers: smoothing and embedding iterations at N = 50 k
    n = len(kh)
    ldd = pitch_for(n)
    kh_d, lab_d = _ffi.DeviceBuffer.from_numpy(kh), _ffi.DeviceBuffer.from_numpy(lab)
    D_d = _ffi.DeviceBuffer(n * ldd)
    hamdist_matrix_dev(kh_d.ptr, lab_d.ptr, n, K, conseq_lens, D_d.ptr, ldd)
    nb_holder = []

    def select():
        for b in nb_holder:
            b.free()
        nb_holder[:] = [V.knn_select_dev(D_d.ptr, ldd, n, 20)]
    out["knn_select"] = roof(float(n) * n, timed_launches(select, 4), "20 nearest rows per row of D (device tie rule), one read of D")
    nb_d = nb_holder[0]
    lds = (n + 127) & ~127
    sums_d = _ffi.DeviceBuffer(n * lds * 2)
    out["knn_sums"] = roof(3.0 * n * n, timed_launches(lambda: V.knn_sums_kmers_dev(kh_d.ptr, lab_d.ptr, n, K, conseq_lens, nb_d, 20, out=sums_d.ptr), 4),
                           "neighbour sums from base-count profiles (8d: N^2 read at ideal reuse + 2 N^2 written; this kernel reads no matrix)")
    D_d.free()
    lut = V.hd_prob_lut(K, 20, 400 * K)
    ld0, ph = V._init_draws(n, 10, 7)
    for tag, mode, iters in (("embed_iter_fast", V.EMBED_FAST, 40), ("embed_iter_seq", V.EMBED_SEQ, 10)):
        sess = V.EmbedSession(n, 10, 0.01, mode)
        _ffi.check(lib.kmap_embed_set_prob_lut(sess._h, sums_d.ptr, lds, _ffi.ptr(lut), len(lut)))    # sums stay ours (not in _keep)
        sess.set_coords(ld0, ph)
        sess.set_jitter(np.random.normal(0, 0.01, 4096))
        ms = [t / 5 for t in timed_launches(lambda: sess.step(5), iters // 5)]
        out[tag] = roof(2.0 * n * n, ms, "one embedding iteration = forces + loss reduction + apply (" +
                        ("FAST: symmetric kernel, each unordered pair once -> N^2 bytes of sums actually read" if mode == V.EMBED_FAST
                         else "SEQ: the reference's summation order, bit-pinned path") + ")")
        sess.close()
    for b in (sums_d, nb_d, kh_d, lab_d):
        b.free()
    return out


# ---- CPU baselines (oracle = the CPU restatement; the checker, timed here as the reported baseline) ------------------------
def cpu_baseline(kh, lab, conseq_lens, target_s=10.0):
    """Oracle (CPU restatement, OpenMP) timed on a bounded row sample of the same N."""
    from oracle import oracle as O
    L = O.lib()
    n = len(kh)
    kh64 = np.ascontiguousarray(kh, np.uint64)
    cl = np.ascontiguousarray(conseq_lens, np.int32)
    cores = os.cpu_count() or 1
    rows = min(n, 2048)
    out = np.empty((rows, n), np.uint8)
    L.ko_hamdist_rows(kh64, lab, n, K, cl, len(cl), 0, rows, out)          # warm-up (thread pool, page faults)
    done, t0 = 0, time.perf_counter()
    while True:                                                            # row blocks round-robin over the matrix
        r0 = done % max(n - rows + 1, 1)
        L.ko_hamdist_rows(kh64, lab, n, K, cl, len(cl), r0, rows, out)
        done += rows
        dt = time.perf_counter() - t0
        if dt >= target_s:
            break
    return {"value": done * n / dt, "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": f"{done} of {n} rows x {n} cols (oracle ko_hamdist_rows, OpenMP, {dt:.1f} s)"}


def cpu_e2e_baseline(kh, lab, conseq_lens, n_reads_sample=30_000, n_embed=3000, embed_iters=3):
    """The CPU oracle on a bounded sample of the C3 job, extrapolated (and labelled so): find_motif for k = 6..9 on
    n_reads_sample synthetic reads (x 10 M / n_reads_sample: every stage is linear in the reads), smoothing + embedding
    iterations at n_embed sampled k-mers (x (50 000 / n_embed)^2 per iteration, x 2500 iterations)."""
    from kmap_amd import synth
    from kmap_amd.kmer_count import gen_motif_def_dict, read_default_config_file
    from oracle import oracle as O
    O.lib()
    cores = os.cpu_count() or 1
    seq, borders = synth.synth_reads(n_reads_sample, 150, 2)
    mdd = gen_motif_def_dict(read_default_config_file())
    t0 = time.perf_counter()
    for k in range(6, 10):
        O.find_motif(seq.copy(), borders, k, mdd[k])
    fm = time.perf_counter() - t0
    idx = np.linspace(0, len(kh) - 1, n_embed).astype(np.int64)
    t0 = time.perf_counter()
    D = O.hamdist_matrix_u8(kh[idx].astype(np.uint64), lab[idx], K, conseq_lens)
    nb = O.knn_select_stable(D, 20)
    S = O.knn_smooth(D.astype(np.int64), 20, nb=nb)
    smooth = time.perf_counter() - t0
    T = O.sigmoid(S, 16.0, change_point=K / 2, scale_factor=0.2 * K - 0.2)
    t0 = time.perf_counter()
    O.umap(T, n_max_iter=embed_iters, random_seed=7)
    it = (time.perf_counter() - t0) / embed_iters
    scale_reads, scale_n2 = 10_000_000 / n_reads_sample, (N_BASE / n_embed) ** 2
    ext = {"scan_motif_s": fm * scale_reads, "smoothing_s": smooth * scale_n2, "embedding_s": it * scale_n2 * 2500}
    return {"kind": "port", "cores": cores, "extrapolated": True,
            "measured": {"find_motif_k6_9_s": fm, "n_reads": n_reads_sample, "hamming_select_smooth_s": smooth, "embed_s_per_iter": it,
                         "n_kmers": n_embed, "embed_iters": embed_iters},
            "extrapolated_to_c3": {**ext, "e2e_s": sum(ext.values())},
            "sample": (f"oracle find_motif k=6..9 on {n_reads_sample} x 150 bp reads (x{scale_reads:.0f}, linear in reads; the occurrence "
                       f"scans and file writing of scan_motif are NOT included), Hamming + neighbour selection + smoothing and "
                       f"{embed_iters} embedding iterations at N={n_embed} (x{scale_n2:.0f} for N=50000, x2500 iterations)")}


# ---- multi-GPU legs ----------------------------------------------------------------------------------------------------
def embed_dist_leg(dist, torch, kh, lab, conseqs, n, iters=100):
    """Sharded embedding of the same N sampled k-mers on all ranks (kmap_amd.distributed): wall time of the iteration loop
    alone (device-synchronised on both sides, max over ranks), after a short run that absorbs one-time costs.  Errors are
    reported, not raised: a rank that failed sets a flag that every rank sees before anyone enters the next phase."""
    from kmap_amd.distributed import kmap_from_kmers_distributed
    err, loop_s, hbm = "", 0.0, {}
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    for it in (3, iters):
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()):
            break
        dist.barrier()
        tr = {}
        try:
            kmap_from_kmers_distributed(kh, np.ones(n, np.int64), lab, conseqs, K, n_max_iter=it, random_seed=7, trace=tr)
            loop_s, hbm = tr["loop_s"], tr["hbm"]
        except Exception as e:   # noqa: BLE001
            err = f"{type(e).__name__}: {e}"[:300]
            flag.fill_(1)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    if int(flag.item()):
        return {"error": err or "another rank failed"}
    t = torch.tensor([loop_s], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    cyc = os.environ.get("KMAP_DIST_CYCLIC", "1") != "0" and n >= 16384
    return {"n_kmers": n, "mode": "FAST, each unordered pair once, cyclic 256-row blocks per rank" if cyc else "FAST, contiguous row blocks",
            "iterations": iters, "loop_s": float(t.item()), "ms_per_iteration": float(t.item()) / iters * 1e3,
            "collectives_per_iteration": 2, "d_rows_per_rank": hbm.get("d_rows"), "d_bytes_per_rank": hbm.get("d_bytes")}


def c4_leg(dist, torch, res_dir, rank, world, barrier):
    """BASELINE config C4: N = 200 000 sampled k-mers FIXED as the GPU count grows (strong scaling).  Hamming rows of this rank
    (HIP events, no collective) and the embedding iteration (row-sharded FAST loop with its two all-reduces for world > 1)."""
    from kmap_amd import _ffi, visualization as V
    from kmap_amd.distributed import row_partition
    from kmap_amd.hamdist import hamdist_matrix_dev, pitch_for
    kh, lab, lens, conseqs = pipeline_sample(res_dir, N_C4) if rank == 0 else (None, None, None, None)
    if dist is not None:
        box = [kh, lab, lens, conseqs]
        dist.broadcast_object_list(box, 0)      # set-up only (1.6 MB once), outside every timed region
        kh, lab, lens, conseqs = box
    n = len(kh)
    row0, nrows = row_partition(n, world, rank)
    ld = pitch_for(n)
    kh_d, lab_d = _ffi.DeviceBuffer.from_numpy(kh), _ffi.DeviceBuffer.from_numpy(lab)
    out_d = _ffi.DeviceBuffer(max(nrows, 1) * ld)
    barrier()
    t0 = time.perf_counter()
    ms = timed_launches(lambda: hamdist_matrix_dev(kh_d.ptr, lab_d.ptr, n, K, lens, out_d.ptr, ld, row0=row0, nrows=nrows), 10)
    barrier()
    wall = time.perf_counter() - t0
    for b in (out_d, kh_d, lab_d):
        b.free()
    res = {"n_kmers": n, "scaling": "strong", "rows_per_gpu": nrows, "hamming_ms_median_rank0": statistics.median(ms), "hamming_ms_min_rank0": min(ms),
           "hamming_pairs_per_s": float(n) * n * 12 / wall, "hamming_note": "12 launches (2 warm-up + 10) of every rank's rows / wall time incl. barriers"}
    its = (3, 23)
    loops = []
    err = ""
    flag = torch.zeros(1, dtype=torch.int32, device="cuda") if dist is not None else None
    for it in its:
        if dist is not None:             # a rank that failed in the previous run is seen by all before anyone starts the next
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if int(flag.item()):
                break
        tr = {}
        try:
            if dist is not None:
                from kmap_amd.distributed import kmap_from_kmers_distributed
                kmap_from_kmers_distributed(kh, np.ones(n, np.int64), lab, conseqs, K, n_max_iter=it, random_seed=7, trace=tr, mode=V.EMBED_FAST)
            else:
                V.kmap_from_kmers(kh, np.ones(n, np.int64), lab, conseqs, K, n_max_iter=it, random_seed=7, trace=tr, mode=V.EMBED_FAST)
            loops.append(tr["loop_s"])
        except Exception as e:   # noqa: BLE001 -- reported in the line; the headline above is already measured
            err = f"{type(e).__name__}: {e}"[:300]
            if flag is not None:
                flag.fill_(1)
            else:
                break
    if dist is not None:
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()):
            err = err or "another rank failed"
    if err or len(loops) < 2:
        res["embed_error"] = err or "embedding leg did not run"
        return res
    per_it = (loops[1] - loops[0]) / (its[1] - its[0])
    if dist is not None:
        t = torch.tensor([per_it], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        per_it = float(t.item())
    res["embed_ms_per_iteration"] = per_it * 1e3
    res["embed_note"] = f"FAST, from the difference of a {its[1]}- and a {its[0]}-iteration run (max over ranks)"
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-embed-dist", action="store_true", help="skip the multi-GPU embedding leg (world > 1)")
    ap.add_argument("--no-c4", action="store_true", help="skip the N = 200 000 strong-scaling leg")
    ap.add_argument("--no-stages", action="store_true", help="skip the per-stage roofline timings")
    ap.add_argument("--e2e", default="full", choices=["none", "k9", "full"],
                    help="end-to-end timings on C3 (rank 0, N=1 only): k9 = k 6..9 in both embedding modes; full = also the default k 6..16")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    import torch
    from kmap_amd import _ffi
    from kmap_amd.e2e import run_e2e, synth_config_reads
    from kmap_amd.hamdist import hamdist_matrix_dev, pitch_for
    # KMAP_BENCH_BACKEND=gloo + KMAP_BENCH_SAME_GPU=1 rehearse the multi-rank path on a one-GPU box (timing collectives
    # over gloo, every rank on GPU 0); the real runs use one GPU per rank and RCCL ("nccl").
    backend = os.environ.get("KMAP_BENCH_BACKEND", "nccl")
    dev = 0 if os.environ.get("KMAP_BENCH_SAME_GPU") else local_rank
    torch.cuda.set_device(dev)
    _ffi.check(_ffi.lib().kmap_set_device(dev))
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        _ffi.sync()

    # ---- the C3 pipeline (rank 0): its hand-over is the workload of the timed kernel; its wall time is e2e["k6_9"]["default"] ----
    n = int(round(N_BASE * math.sqrt(world) / 16)) * 16
    reads, first, res_dir = None, None, None
    if rank == 0:
        os.environ["KMAP_DIST_DISABLE"] = "1"       # rank 0 runs the pipeline on its own GPU; the verbs must not join the process group
        reads = synth_config_reads("C3")
        first = run_e2e("C3", "default", reads=reads, keep=True)
        os.environ.pop("KMAP_DIST_DISABLE")
        res_dir = first["res_dir"]
        kh, lab, lens, conseqs = pipeline_sample(res_dir, n)
    else:
        kh = lab = lens = conseqs = None
    if dist is not None:
        box = [kh, lab, lens, conseqs]
        dist.broadcast_object_list(box, 0)          # set-up only, outside the timed region
        kh, lab, lens, conseqs = box
    n = len(kh)
    rows_per = (n + world - 1) // world
    row0 = rank * rows_per
    nrows = max(0, min(rows_per, n - row0))
    ld = pitch_for(n)
    kh_d, lab_d = _ffi.DeviceBuffer.from_numpy(kh), _ffi.DeviceBuffer.from_numpy(lab)
    out_d = _ffi.DeviceBuffer(max(nrows, 1) * ld)

    def step():
        hamdist_matrix_dev(kh_d.ptr, lab_d.ptr, n, K, lens, out_d.ptr, ld, row0=row0, nrows=nrows)

    for _ in range(args.warmup):
        step()
    barrier()
    evs = [_ffi.Event() for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(args.steps):
        step()
        evs[i + 1].record()
    barrier()
    wall = time.perf_counter() - t0
    per_launch = [evs[i].elapsed_ms(evs[i + 1]) for i in range(args.steps)]
    kern_ms = evs[0].elapsed_ms(evs[args.steps]) / args.steps
    if dist is not None:
        t = torch.tensor([wall], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())

    # spot-check the timed output against the oracle (a wrong fast kernel is not a result): rows of every label
    from oracle import oracle as O
    if nrows:
        picks = sorted({row0, row0 + nrows // 2, row0 + nrows - 1} | {int(np.searchsorted(lab, l)) for l in np.unique(lab) if row0 <= int(np.searchsorted(lab, l)) < row0 + nrows})
        for r in picks:
            got = out_d.to_numpy(np.uint8, (n,), offset=(r - row0) * ld)
            want = np.empty((1, n), np.uint8)
            O.lib().ko_hamdist_rows(np.ascontiguousarray(kh, np.uint64), lab, n, K, np.ascontiguousarray(lens, np.int32), len(lens), r, 1, want)
            assert np.array_equal(got, want[0]), f"timed kernel output differs from the oracle in row {r}"
    out_d.free()

    embed_dist = None
    if dist is not None and not args.no_embed_dist:
        embed_dist = embed_dist_leg(dist, torch, kh, lab, conseqs, n)
    c4 = None
    if not args.no_c4:
        if dist is not None:
            box = [res_dir]
            dist.broadcast_object_list(box, 0)
        c4 = c4_leg(dist, torch, res_dir, rank, world, barrier)

    if rank == 0:
        pairs_total = float(n) * float(n)
        algo_bytes = nrows * n + 5 * n          # u8 out + u32 hashes + u8 group ids, per launch on this rank
        achieved = algo_bytes / (kern_ms * 1e-3) / 1e9
        short_rows = int(sum(int(np.count_nonzero(lab == l)) for l, c in enumerate(lens) if c < K))
        line = {
            "metric": "hamming_pairs_per_s", "value": pairs_total * args.steps / wall, "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"C3 Hamming stage: all-pairs Hamming matrix of the N={n} 8-mers the C3 pipeline samples "
                                   f"(finals {'/'.join(conseqs)}), uint8 out, row-sharded over {world} GPU(s)",
                       "n_kmers": n, "k": K, "final_conseq": conseqs, "conseq_lens": lens, "rows_of_short_consensus_labels": short_rows,
                       "rows_per_gpu": rows_per, "pairs_per_gpu_per_step": float(nrows) * n,
                       "sample": "sample_kmers.pkl of scan_motif on 10M x 150 bp synthetic reads (k=6..9), expanded by counts"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "hamdist_tile_kernel<1 code word> (+ build_gid / build_codes pre-passes, inside the timed region)",
                         "kernel_ms": kern_ms, "kernel_ms_min": min(per_launch), "kernel_ms_median": statistics.median(per_launch),
                         "kernel_ms_max": max(per_launch), "algorithmic_bytes": algo_bytes},
        }
        # HBM bytes per launch from the PMC passes of the same command (rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE, separate
        # runs, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes); collected offline, stored under profiles/
        pmc = sorted((ROOT / "profiles").glob("r*_bench_pmc.json"))
        if pmc and world == 1:
            try:
                kern = json.loads(pmc[-1].read_text())["kernels"]
                hk = [v for k_, v in kern.items() if "hamdist_tile_kernel" in k_ or "hamdist_matrix_kernel" in k_][0]
                line["roofline"]["traffic"] = hk["hbm_bytes_per_launch"]
                line["roofline"]["traffic_source"] = f"profiles/{pmc[-1].name}"
            except Exception:
                pass
        if embed_dist is not None:
            line["embed_dist"] = embed_dist
        if c4 is not None:
            line["c4"] = c4
        if world == 1 and not args.no_stages:
            line["roofline"]["stages"] = stage_rooflines(reads, kh, lab, lens)
        kh_d.free()
        lab_d.free()
        if world == 1 and args.e2e != "none":
            # the other half of BASELINE.json's metric: end-to-end wall time on a clean res_dir (synthetic reads are generated
            # and written outside the timed stages; scan_motif loads them from the pickles like the reference)
            def pack(r):
                return {"scan_motif_s": r["times"]["scan_motif_s"], "visualize_kmers_s": r["times"]["visualize_kmers_s"],
                        "e2e_s": r["times"]["e2e_s"], "final_conseq": r["final_conseq"], "stages": r["stages"]}
            e2e = {"k6_9": {"default": pack(first), "seq": pack(run_e2e("C3", "seq", reads=reads))}}
            if args.e2e == "full":
                e2e["k6_16"] = {"default": pack(run_e2e("C3", "default", min_k=6, max_k=16, reads=reads))}
            e2e["workload"] = (f"C3: {first['n_reads']} x {first['read_len']} bp synthetic reads, N={first['n_total']} sampled k-mers, "
                               f"{first['iters']} iterations, 1 GPU, clean res_dir; k6_9: k = 6..9 (longest final = the configs' k = 8), k6_16: the "
                               f"reference's default k range (default_config.toml:7-8); default = package default embedding mode (FAST "
                               f"wavefront sums above N = 16384: per-step pinned, no digit-level reference exists at this N), seq = the "
                               f"reference's summation order (the parity-grade number)")
            line["e2e"] = e2e
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(kh, lab, lens)
            line["cpu_baseline"]["e2e"] = cpu_e2e_baseline(kh, lab, lens)
            if "e2e" in line:
                cpu_s = line["cpu_baseline"]["e2e"]["extrapolated_to_c3"]["e2e_s"]
                line["cpu_baseline"]["e2e"]["gpu_e2e_s"] = {"default": line["e2e"]["k6_9"]["default"]["e2e_s"], "seq": line["e2e"]["k6_9"]["seq"]["e2e_s"]}
                line["cpu_baseline"]["e2e"]["ratio_cpu_over_gpu"] = {m: cpu_s / v for m, v in line["cpu_baseline"]["e2e"]["gpu_e2e_s"].items()}
        print(json.dumps(line), flush=True)
    if res_dir:
        import shutil
        shutil.rmtree(res_dir, ignore_errors=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
