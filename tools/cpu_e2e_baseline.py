#!/usr/bin/env python3
"""CPU baseline of the end-to-end path with the oracle (C/OpenMP + numpy), on a bounded sample:
scan_motif-equivalent work (count + find_motif for k = 6..9, occurrence scan of the final consensuses) on `--reads` synthetic
reads, and `--iters` embedding iterations at `--n` sampled k-mers (time per iteration is extrapolated to 2500).
Reported with the host's core count; a baseline, not a target."""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=1_000_000)
    ap.add_argument("--read_len", type=int, default=150)
    ap.add_argument("--n", type=int, default=5000)
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    from kmap_amd import synth
    from kmap_amd.kmer_count import gen_motif_def_dict, read_default_config_file
    from oracle import oracle as O
    O.lib()
    seq, borders = synth.synth_reads(args.reads, args.read_len, 2)
    mdd = gen_motif_def_dict(read_default_config_file())
    out = {"cores": os.cpu_count(), "reads": args.reads, "read_len": args.read_len}
    t0 = time.perf_counter()
    cands = []
    for k in range(6, 10):
        res, _, _ = O.find_motif(seq.copy(), borders, k, mdd[k])
        cands += [O.hash2kmer(h, k) for h in res]
    out["find_motif_k6_9_s"] = time.perf_counter() - t0
    out["candidates"] = cands
    # occurrence scan: the C kernel per read for each candidate (the reference scans per k + finals)
    import ctypes as C
    t0 = time.perf_counter()
    buf, md = np.empty(args.read_len + 1, np.int32), C.c_int(0)
    n_scan = min(args.reads, 200_000)
    for c in cands[:3]:
        kh, k = int(O.kmer2hash(c)), len(c)
        for st, en in borders[:n_scan]:
            O.lib().ko_scan_read(seq[st:en], en - st, k, kh, mdd[k].max_ham_dist, 1, buf, C.byref(md))
    out["occurrence_scan_s_per_read_consensus"] = (time.perf_counter() - t0) / max(1, n_scan * len(cands[:3]))
    # embedding: oracle umap iterations at N sampled k-mers
    rng = np.random.default_rng(0)
    n = args.n
    kh = rng.integers(0, 4 ** 8, size=n, dtype=np.uint64)
    t0 = time.perf_counter()
    D = O.hamdist_matrix_u8(kh, np.zeros(n, np.int32), 8, [8])
    out["hamdist_matrix_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    nb = O.knn_select_stable(D, 20)
    S = O.knn_smooth(D.astype(np.int64), 20, nb=nb)
    out["knn_smooth_s"] = time.perf_counter() - t0
    T = O.sigmoid(S, 16.0, change_point=4.0, scale_factor=1.4)
    t0 = time.perf_counter()
    O.umap(T, n_max_iter=args.iters, random_seed=7)
    dt = (time.perf_counter() - t0) / args.iters
    out["embed_s_per_iter"] = dt
    out["embed_2500_iters_extrapolated_s"] = dt * 2500
    out["n"] = n
    print(json.dumps(out))


if __name__ == "__main__":
    main()
