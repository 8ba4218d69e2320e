#!/usr/bin/env python3
"""upload stage of scan_motif as bench.py sees it: reads generated in this process, run_e2e three times (C3, k = 6..9, 20 iterations)"""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))


def main():
    import torch
    torch.cuda.set_device(0)
    from kmap_amd import _ffi
    from kmap_amd.e2e import run_e2e, synth_config_reads
    _ffi.check(_ffi.lib().kmap_set_device(0))
    reads = synth_config_reads("C3")
    for i in range(3):
        r = run_e2e("C3", "fast", reads=reads, iters=20)
        st = r["stages"]
        print(i, "scan_motif_s %.3f" % r["times"]["scan_motif_s"], {k: round(v, 3) for k, v in st.items() if k in ("load_inputs", "upload", "find_motif", "join_table_writers", "sample_kmers", "occurrence_per_k")}, flush=True)


if __name__ == "__main__":
    main()
