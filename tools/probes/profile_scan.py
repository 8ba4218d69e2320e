#!/usr/bin/env python3
"""cProfile of a C3 `scan_motif` in-process: where the host time of the verb goes (second run: warm library).
    python tools/probes/profile_scan.py [C3] [max_k = 9] [reports]"""
import cProfile
import io
import pstats
import sys
import tempfile
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np


def main():
    from kmap_amd import e2e, synth, motif_discovery as md
    cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
    max_k = int(sys.argv[2]) if len(sys.argv) > 2 else 9
    reports = len(sys.argv) > 3 and sys.argv[3] == "reports"      # the reference's default report flags on
    c = e2e.CONFIGS[cfg]
    reads = e2e.synth_config_reads(cfg)
    over = {"kmer_count": {"min_k": 6, "max_k": max_k},
            "motif_discovery": {"motif_pos_density_flag": reports, "motif_co_occurence_flag": reports, "gen_hamball_flag": reports,
                                "n_total_sample": c["n_total"], "n_motif_sample": c["n_motif"]},
            "visualization": {"gen_fig_flag": False, "random_seed": 7, "n_max_iter": 10}}
    for rep in range(2):
        res = Path(tempfile.mkdtemp(prefix="kmap_prof_"))
        synth.write_res_dir(res, reads[0], reads[1], over)
        np.random.seed(123)
        pr = cProfile.Profile()
        md.STAGE_TIMES.clear()
        t0 = time.perf_counter()
        pr.enable()
        md._scan_motif(str(res))
        pr.disable()
        print(f"run {rep}: {time.perf_counter() - t0:.3f} s", {k: round(v, 3) for k, v in md.STAGE_TIMES.items()})
        if rep == int(__import__('os').environ.get('PROFILE_REP', '1')):
            s = io.StringIO()
            pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
            print(s.getvalue())
            s = io.StringIO()
            pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(25)
            print(s.getvalue())


if __name__ == "__main__":
    main()
