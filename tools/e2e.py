#!/usr/bin/env python3
"""End-to-end `scan_motif` + `visualize_kmers` on synthetic reads (see kmap_amd/e2e.py).

    python tools/e2e.py --config C2            # 100 k x 150 bp, N = 5 k, 2500 iterations
    python tools/e2e.py --config C3 --mode fast
"""
import argparse
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    from kmap_amd.e2e import CONFIGS, run_e2e
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C2", choices=sorted(CONFIGS))
    ap.add_argument("--mode", default="default", choices=["default", "seq", "fast", "exact"])
    ap.add_argument("--min_k", type=int, default=6)
    ap.add_argument("--max_k", type=int, default=9)
    ap.add_argument("--iters", type=int, default=None)
    ap.add_argument("--keep", action="store_true")
    ap.add_argument("--reports", action="store_true", help="also produce the position-density / co-occurrence / Hamming-ball data files")
    args = ap.parse_args()
    print(json.dumps(run_e2e(args.config, args.mode, args.min_k, args.max_k, args.iters, args.keep, reports=args.reports)))


if __name__ == "__main__":
    main()
