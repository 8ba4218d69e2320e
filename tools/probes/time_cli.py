#!/usr/bin/env python3
"""The three verbs as a user runs them -- three processes, `python -m kmap_amd preproc | scan_motif | visualize_kmers` -- on a
synthetic FASTA of config C3's size (10 M x 150 bp reads with the two planted motifs, k = 6..9, N = 50 000, 2500 iterations):
wall time of each process (interpreter start, imports, library load and HIP initialisation included), next to the in-process
stage times bench.py's `e2e` reports.

    python tools/probes/time_cli.py [--config C3|C2] [--mode default|fast]
"""
import argparse
import os
import shutil
import subprocess
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))


def write_fasta(path, seq, borders):
    """the reads of the uint8 array contract as FASTA text (fixed-length reads: one vectorised pass per million reads)"""
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    n = len(borders)
    L = int(borders[0, 1] - borders[0, 0])
    assert np.all(borders[:, 1] - borders[:, 0] == L)
    with open(path, "wb") as fh:
        for r0 in range(0, n, 1_000_000):
            m = min(1_000_000, n - r0)
            rec = np.empty((m, 10 + L + 1), dtype=np.uint8)
            hdr = np.char.add(">r", np.char.zfill(np.arange(r0, r0 + m).astype(str), 7)).astype("S9")
            rec[:, :9] = np.frombuffer(hdr.tobytes(), dtype=np.uint8).reshape(m, 9)
            rec[:, 9] = 10
            codes = seq[int(borders[r0, 0]):int(borders[r0 + m - 1, 1]) + 1].reshape(m, L + 1)[:, :L]
            rec[:, 10:10 + L] = lut[np.minimum(codes, 4)]
            rec[:, -1] = 10
            rec.tofile(fh)
    return os.path.getsize(path)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C3")
    ap.add_argument("--mode", default="default", choices=["default", "fast"])
    args = ap.parse_args()
    from kmap_amd import synth
    from kmap_amd._toml import dump_toml
    from kmap_amd.e2e import CONFIGS
    from kmap_amd.kmer_count import FileNameDict, read_default_config_file
    c = CONFIGS[args.config]
    tmp = Path(tempfile.mkdtemp(prefix="kmap_cli_"))
    try:
        t = time.perf_counter()
        seq, borders = synth.synth_reads(c["n_reads"], c["read_len"], c["seed"])
        fa = tmp / "reads.fa"
        size = write_fasta(fa, seq, np.asarray(borders))
        del seq, borders
        print(f"{args.config}: {c['n_reads']} x {c['read_len']} bp reads -> {size / 1e9:.2f} GB of FASTA ({time.perf_counter() - t:.1f} s to make)")
        res = tmp / "res"
        res.mkdir()
        cfg = read_default_config_file()
        cfg["general"]["input_fasta_file"] = str(fa)
        cfg["general"]["res_dir"] = str(res)
        cfg["kmer_count"].update({"min_k": 6, "max_k": 9})
        cfg["motif_discovery"].update({"motif_pos_density_flag": False, "motif_co_occurence_flag": False, "gen_hamball_flag": False,
                                       "n_total_sample": c["n_total"], "n_motif_sample": c["n_motif"]})
        cfg["visualization"].update({"gen_fig_flag": False, "random_seed": 7, "n_max_iter": c["iters"]})
        if args.mode == "fast":
            cfg["visualization"]["embed_mode"] = "fast"
        dump_toml(cfg, res / FileNameDict["config_file"])
        env = dict(os.environ, PYTHONPATH=str(ROOT) + os.pathsep + os.environ.get("PYTHONPATH", ""))
        total = 0.0
        for verb, extra in (("preproc", ["--fasta_file", str(fa)]), ("scan_motif", []), ("visualize_kmers", [])):
            t0 = time.perf_counter()
            r = subprocess.run([sys.executable, "-m", "kmap_amd", verb, "--res_dir", str(res), *extra], env=env, capture_output=True, text=True)
            dt = time.perf_counter() - t0
            total += dt
            print(f"  {verb:<16} {dt:6.2f} s wall  (rc {r.returncode})")
            if r.returncode != 0:
                print(r.stdout[-1500:], r.stderr[-3000:])
                return 1
        print(f"  {'all three':<16} {total:6.2f} s wall")
        print("  final consensus:", (res / "final_conseq.txt").read_text().split(), " rows embedded:",
              len((res / "low_dim_data.tsv").read_text().splitlines()) - 1)
        t0 = time.perf_counter()
        r = subprocess.run([sys.executable, "-c", "import kmap_amd"], env=env)
        print(f"  (python -c 'import kmap_amd': {time.perf_counter() - t0:.2f} s)")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
