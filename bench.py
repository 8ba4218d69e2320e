#!/usr/bin/env python3
"""bench.py -- headline benchmark of the kmap hot path on MI355X.

Metric (BASELINE.json): Hamming pairs/s of the sampled-k-mer all-pairs matrix at N = 50 k, k = 8
(config C3).  One "step" = one pass of the Hamming-matrix kernel over the whole batch of sampled
k-mers: N x N ordered pairs written as uint8 into HBM; hashes/labels are resident in HBM before
the timed region.  value = pairs processed by all ranks / max-over-ranks wall time.

Multi-GPU (--gpus G, launched by torch.distributed.run, one rank per GPU): the matrix is sharded
by row blocks, every rank holds all N hashes, no data-path collective.  Weak scaling:
N_total = 50 000 * sqrt(G) (rounded to 16) so that every GPU keeps 2.5e9 pairs per step.

Extra objects on the JSON line: `roofline` (HIP-event kernel time vs the 8 TB/s HBM peak, algorithmic
bytes = rows*N + 5N) and `cpu_baseline` (the CPU oracle's OpenMP matrix kernel on a bounded row
sample of the same workload; a reported baseline, not the target).
"""
import argparse
import json
import math
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ceiling ~6290 GB/s
K = 8
N_BASE = 50_000
CONSEQ_LENS = [8, 8]    # two full-length motif labels + noise label 2 (synthetic C3 sample)


def synth_sample(n, seed):
    """Sampled k-mers as scan_motif hands them over: hashes expanded by counts (duplicates allowed --
    k=8 has only 32 896 revcom-merged k-mers), labels grouped 0,1,noise."""
    rng = np.random.default_rng(seed)
    n_motif = n // 2
    kh = np.concatenate([rng.integers(0, 4 ** K, size=n_motif // 2, dtype=np.uint64),
                         rng.integers(0, 4 ** K, size=n_motif - n_motif // 2, dtype=np.uint64),
                         rng.integers(0, 4 ** K, size=n - n_motif, dtype=np.uint64)]).astype(np.uint32)
    lab = np.concatenate([np.zeros(n_motif // 2), np.ones(n_motif - n_motif // 2), np.full(n - n_motif, 2)]).astype(np.int32)
    return kh, lab


def cpu_baseline(kh, lab, target_s=12.0):
    """Oracle (CPU restatement, OpenMP) timed on a bounded row sample of the same N."""
    from oracle import oracle as O
    L = O.lib()
    n = len(kh)
    kh64 = np.ascontiguousarray(kh, np.uint64)
    cl = np.ascontiguousarray(CONSEQ_LENS, np.int32)
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
    rows2 = done
    return {"value": rows2 * n / dt, "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": f"{rows2} of {n} rows x {n} cols (oracle ko_hamdist_rows, OpenMP, {dt:.1f} s)"}


def embed_dist_leg(dist, torch, kh, lab, n, out_d, iters=100):
    """Sharded embedding of the same N sampled k-mers on all ranks (kmap_amd.distributed): wall time of the iteration loop
    alone (device-synchronised on both sides, max over ranks), after a short run that absorbs one-time costs.  Errors are
    reported, not raised: a rank that failed sets a flag that every rank sees before anyone enters the next phase."""
    from kmap_amd.distributed import kmap_from_kmers_distributed
    out_d.free()
    err, loop_s = "", 0.0
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    for it in (3, iters):
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()):
            break
        dist.barrier()
        tr = {}
        try:
            kmap_from_kmers_distributed(kh, np.ones(n, np.int64), lab, ["A" * K, "C" * K], K, n_max_iter=it, random_seed=7, trace=tr)
            loop_s = tr["loop_s"]
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
            "collectives_per_iteration": 2}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-embed-dist", action="store_true", help="skip the multi-GPU embedding leg (world > 1)")
    ap.add_argument("--e2e", default="C3", choices=["none", "C2", "C3"],
                    help="also time scan_motif + visualize_kmers end to end on this synthetic config (rank 0, N=1 only)")
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

    # ---- workload (weak scaling: constant pairs per GPU) ----
    n = int(round(N_BASE * math.sqrt(world) / 16)) * 16
    rows_per = (n + world - 1) // world
    row0 = rank * rows_per
    nrows = max(0, min(rows_per, n - row0))
    kh, lab = synth_sample(n, seed=2)
    ld = pitch_for(n)
    kh_d, lab_d = _ffi.DeviceBuffer.from_numpy(kh), _ffi.DeviceBuffer.from_numpy(lab)
    out_d = _ffi.DeviceBuffer(max(nrows, 1) * ld)

    def step():
        hamdist_matrix_dev(kh_d.ptr, lab_d.ptr, n, K, CONSEQ_LENS, out_d.ptr, ld, row0=row0, nrows=nrows)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        _ffi.sync()

    for _ in range(args.warmup):
        step()
    barrier()
    ev0, ev1 = _ffi.Event(), _ffi.Event()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    barrier()
    wall = time.perf_counter() - t0
    kern_ms = ev0.elapsed_ms(ev1) / args.steps
    if dist is not None:
        t = torch.tensor([wall], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())

    # spot-check the timed output against the oracle (a wrong fast kernel is not a result)
    from oracle import oracle as O
    chk_rows = min(nrows, 8)
    if chk_rows:
        got = out_d.to_numpy(np.uint8, (chk_rows, ld))[:, :n]
        want = np.empty((chk_rows, n), np.uint8)
        O.lib().ko_hamdist_rows(np.ascontiguousarray(kh, np.uint64), lab, n, K, np.ascontiguousarray(CONSEQ_LENS, np.int32),
                                len(CONSEQ_LENS), row0, chk_rows, want)
        assert np.array_equal(got, want), "timed kernel output differs from the oracle"

    # ---- multi-GPU embedding leg (world > 1): the row-sharded iteration with its two all-reduces over RCCL ----
    embed_dist = None
    if dist is not None and not args.no_embed_dist:
        embed_dist = embed_dist_leg(dist, torch, kh, lab, n, out_d)

    if rank == 0:
        pairs_total = float(n) * float(n)
        algo_bytes = nrows * n + 5 * n          # u8 out + u32 hashes + u8 group ids, per launch on this rank
        achieved = algo_bytes / (kern_ms * 1e-3) / 1e9
        line = {
            "metric": "hamming_pairs_per_s", "value": pairs_total * args.steps / wall, "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"C3 Hamming stage: all-pairs Hamming matrix of N={n} sampled 8-mers "
                                   f"(u32 hashes, labels 0/1/noise), uint8 out, row-sharded over {world} GPU(s)",
                       "n_kmers": n, "k": K, "rows_per_gpu": rows_per, "pairs_per_gpu_per_step": float(nrows) * n},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel": "hamdist_tile_kernel<1 code word> (+ build_gid / build_codes pre-passes, inside the timed region)",
                         "kernel_ms": kern_ms, "algorithmic_bytes": algo_bytes},
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
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(kh, lab)
        if world == 1 and args.e2e != "none":
            # the other half of BASELINE.json's metric: end-to-end wall time on a clean res_dir (synthetic reads are
            # generated and written outside the timed stages; scan_motif loads them from the pickles like the reference)
            from kmap_amd.e2e import run_e2e
            del out_d
            line["e2e"] = {}
            for mode in ("default", "seq"):   # default = FAST above N = 16384 (SEQ below); seq = reference summation order
                r = run_e2e(args.e2e, mode)
                line["e2e"][mode] = {"scan_motif_s": r["times"]["scan_motif_s"], "visualize_kmers_s": r["times"]["visualize_kmers_s"],
                                     "e2e_s": r["times"]["e2e_s"], "final_conseq": r["final_conseq"], "stages": r["stages"]}
            line["e2e"]["workload"] = (f"{args.e2e}: {r['n_reads']} x {r['read_len']} bp synthetic reads, k={r['k_range'][0]}..{r['k_range'][1]}, "
                                       f"N={r['n_total']} sampled k-mers, {r['iters']} iterations, 1 GPU; default = package default embedding mode "
                                       f"(FAST wavefront sums above N = 16384), seq = the reference's summation order")
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
