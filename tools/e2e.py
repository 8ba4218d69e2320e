#!/usr/bin/env python3
"""End-to-end `scan_motif` + `visualize_kmers` on synthetic reads (BASELINE configs C2 / C3), with wall-clock per stage.

    python tools/e2e.py --config C2            # 100 k x 150 bp, N = 5 k, 2500 iterations
    python tools/e2e.py --config C3 --mode fast
Plot-only flags are off (as BASELINE.md prescribes for both sides); k range 6..9 so that the longest final consensus
is an 8-mer ("k = 8" of the configs); np.random.seed(123) before scan_motif; visualization.random_seed = 7.
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

CONFIGS = {
    "C1s": dict(n_reads=2000, read_len=60, seed=9, n_total=300, n_motif=150, iters=100),
    "C2": dict(n_reads=100_000, read_len=150, seed=1, n_total=5000, n_motif=2500, iters=2500),
    "C3": dict(n_reads=10_000_000, read_len=150, seed=2, n_total=50_000, n_motif=25_000, iters=2500),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C2", choices=sorted(CONFIGS))
    ap.add_argument("--mode", default="seq", choices=["seq", "fast"])
    ap.add_argument("--min_k", type=int, default=6)
    ap.add_argument("--max_k", type=int, default=9)
    ap.add_argument("--iters", type=int, default=None)
    ap.add_argument("--keep", action="store_true")
    args = ap.parse_args()
    c = CONFIGS[args.config]
    os.environ["KMAP_EMBED_MODE"] = args.mode
    from kmap_amd import synth
    from kmap_amd.motif_discovery import _scan_motif
    from kmap_amd.visualization import _visualize_kmers
    t = {}
    t0 = time.perf_counter()
    seq, borders = synth.synth_reads(c["n_reads"], c["read_len"], c["seed"])
    t["synth_s"] = time.perf_counter() - t0
    res = Path(tempfile.mkdtemp(prefix=f"kmap_{args.config}_"))
    over = {"kmer_count": {"min_k": args.min_k, "max_k": args.max_k},
            "motif_discovery": {"motif_pos_density_flag": False, "motif_co_occurence_flag": False, "gen_hamball_flag": False,
                                "n_total_sample": c["n_total"], "n_motif_sample": c["n_motif"]},
            "visualization": {"gen_fig_flag": False, "random_seed": 7, "n_max_iter": args.iters or c["iters"]}}
    t0 = time.perf_counter()
    synth.write_res_dir(res, seq, borders, over)
    t["write_inputs_s"] = time.perf_counter() - t0
    del seq, borders
    try:
        np.random.seed(123)
        t0 = time.perf_counter()
        _scan_motif(str(res))
        t["scan_motif_s"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        _visualize_kmers(str(res))
        t["visualize_kmers_s"] = time.perf_counter() - t0
        t["e2e_s"] = t["scan_motif_s"] + t["visualize_kmers_s"]
        finals = (res / "final_conseq.txt").read_text().split()
        rows = (res / "low_dim_data.tsv").read_text().splitlines()
        from kmap_amd import motif_discovery as md, visualization as vzm
        md.STAGE_TIMES.update({"viz_" + k: v for k, v in vzm.STAGE_TIMES.items()})
        out = {"config": args.config, "mode": args.mode, **c, "k_range": [args.min_k, args.max_k], "final_conseq": finals,
               "n_embedded": len(rows) - 1, "times": t, "stages": getattr(md, "STAGE_TIMES", {})}
        print(json.dumps(out))
    finally:
        if not args.keep:
            shutil.rmtree(res, ignore_errors=True)


if __name__ == "__main__":
    main()
