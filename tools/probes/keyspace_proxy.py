#!/usr/bin/env python3
"""One GPU: every rank's share of a key-space-sharded count pass (kmap_counts_run_packed_range_dev) on the C3 reads, shard after shard,
next to the one-GPU pass and to a read shard's pass (whose table collective is NOT included).  HIP events, median of `reps`.
  python3 tools/probes/keyspace_proxy.py [--ks 12,14,16] [--gs 2,4,8] [--dedupe]"""
import argparse
import json
import statistics
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ks", default="12,14,16")
    ap.add_argument("--gs", default="2,4,8")
    ap.add_argument("--dedupe", action="store_true")
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--config", default="C3")
    args = ap.parse_args()
    from bench import timed_launches
    from kmap_amd.distributed import row_partition
    from kmap_amd.e2e import synth_config_reads
    from kmap_amd.kmer_count import DeviceCounts
    from kmap_amd.motif_discovery import DeviceSeq
    seq, borders = synth_config_reads(args.config)
    ds, dc = DeviceSeq(seq, borders), DeviceCounts()
    out = {}

    def med(fn):
        return statistics.median(timed_launches(fn, args.reps, warmup=1))
    for k in [int(v) for v in args.ks.split(",")]:
        one = med(lambda: ds.count(dc, k, dedupe=args.dedupe, merge_revcom=True))
        row = {"one_gpu_ms": one}
        n_bins = 4 ** k
        for G in [int(v) for v in args.gs.split(",")]:
            bounds = [(n_bins * r // G) & ~7 for r in range(G)] + [n_bins]
            ms = [med(lambda: ds.count_range(dc, k, args.dedupe, True, bounds[r], bounds[r + 1] - bounds[r])) for r in range(G)]
            row[f"G{G}"] = {"shard_ms": [round(v, 3) for v in ms], "max_ms": max(ms), "speedup": one / max(ms)}
        out[k] = row
        print(k, json.dumps(row), flush=True)
    # the read-sharded pass of one rank at G = 8 (its table passes do not shrink; + the all-reduce of 4^k x 4 B)
    r0, nr = row_partition(len(borders), 8, 3)
    lo, hi = int(borders[r0, 0]), int(borders[r0 + nr - 1, 1]) + 1
    ds.close()
    sh = DeviceSeq(np.ascontiguousarray(seq[lo:hi]), borders[r0:r0 + nr] - lo)
    for k in [int(v) for v in args.ks.split(",")]:
        out[k]["read_shard_G8_ms"] = med(lambda: sh.count(dc, k, dedupe=args.dedupe, merge_revcom=True))
        print(k, "read shard of 8:", out[k]["read_shard_G8_ms"], flush=True)
    sh.close()
    dc.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
