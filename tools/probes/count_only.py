#!/usr/bin/env python3
"""Only count passes at C3 (k, repetitions, dedupe 0/1): workload for kernel-time / PMC passes on the counting kernels."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))


def main():
    from kmap_amd import _ffi, synth
    from kmap_amd.kmer_count import DeviceCounts
    from kmap_amd.motif_discovery import DeviceSeq
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    dedupe = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
    seq, borders = synth.synth_reads(10_000_000, 150, 2)
    ds = DeviceSeq(seq, borders)
    dc = DeviceCounts()
    for rep in range(reps):
        _ffi.sync()
        t0 = time.perf_counter()
        ds.count(dc, k, dedupe=dedupe, merge_revcom=True)
        _ffi.sync()
        print(f"count k={k} dedupe={dedupe}: {1e3 * (time.perf_counter() - t0):.2f} ms", flush=True)


if __name__ == "__main__":
    main()
