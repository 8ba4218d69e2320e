#!/usr/bin/env python3
"""Throughput of the Hamming-ball occurrence scan and of k-mer counting over resident reads (BASELINE config C5 shape:
k = 14, max_ham_dist = 5, 300 bp reads).  Reports positions/s and GB/s against the 1 B/position algorithmic traffic."""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=5_000_000)
    ap.add_argument("--read_len", type=int, default=300)
    ap.add_argument("--k", type=int, default=14)
    ap.add_argument("--radius", type=int, default=5)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    from kmap_amd import _ffi, synth
    from kmap_amd.kmer_count import DeviceCounts, kmer2hash
    from kmap_amd.motif_discovery import DeviceSeq
    motif = "AGGACCTACGTACA"[:args.k] if args.k <= 14 else "AGGACCTACGTACA" + "C" * (args.k - 14)
    t0 = time.perf_counter()
    seq, borders = synth.synth_reads(args.reads, args.read_len, 3, motifs=(motif, "AATCGATAGC"))
    t_syn = time.perf_counter() - t0
    ds = DeviceSeq(seq, borders)
    n = len(seq)
    out = {"reads": args.reads, "read_len": args.read_len, "positions": n, "k": args.k, "radius": args.radius, "synth_s": t_syn}
    hits, pos = ds.scan(args.k, kmer2hash(motif), args.radius, True)   # warm-up + result
    _ffi.sync()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        ds.scan(args.k, kmer2hash(motif), args.radius, True)
    _ffi.sync()
    dt = (time.perf_counter() - t0) / args.reps
    out["scan"] = {"s_per_pass_incl_fetch": dt, "positions_per_s": n / dt, "GBps_algorithmic": n / dt / 1e9,
                   "reads_with_hit": int(np.count_nonzero(hits)), "total_hits": int(hits.sum())}
    import ctypes as C
    tot = _ffi.i64(0)
    t0 = time.perf_counter()
    for _ in range(args.reps):   # device part only: nibble pass + per-read passes + scan of the counts (results stay in HBM)
        _ffi.check(_ffi.lib().kmap_scan_run_packed_dev(ds._scan, ds.codes.ptr, ds.inval_orig.ptr, ds.n, ds.borders.ptr, ds.n_seq,
                                                       args.k, int(kmer2hash(motif)), args.radius, 1, C.byref(tot), ds.planes.ptr, None))
    _ffi.sync()
    dt = (time.perf_counter() - t0) / args.reps
    out["scan"].update({"s_per_pass_device": dt, "positions_per_s_device": n / dt})
    dc = DeviceCounts()
    for dedupe in (True, False):
        ds.count(dc, args.k, dedupe=dedupe, merge_revcom=True)
        _ffi.sync()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            nu = ds.count(dc, args.k, dedupe=dedupe, merge_revcom=True)
        _ffi.sync()
        dt = (time.perf_counter() - t0) / args.reps
        out[f"count_dedupe{int(dedupe)}"] = {"s_per_pass": dt, "positions_per_s": n / dt, "n_uniq": int(nu)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
