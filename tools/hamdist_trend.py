#!/usr/bin/env python3
"""Per-launch times of the headline Hamming kernel over a long back-to-back run (is the spread a clock ramp after idle, or the
kernel?) next to a device fill of the same bytes.  Env switches of the kernel (KMAP_HAMDIST_*) apply; one process per variant."""
import statistics
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from kmap_amd import _ffi  # noqa: E402
from kmap_amd.hamdist import hamdist_matrix_dev, pitch_for  # noqa: E402
from kmap_amd.kmer_count import kmer2hash  # noqa: E402

K = 8


def sample(n, seed):
    rng = np.random.default_rng(seed)
    parts, labs = [], []
    for lab, (core, m) in enumerate((("CCTACGTA", n // 3), ("ATCGATAC", n // 6))):
        kh = np.full(m, int(kmer2hash(core)), np.uint64)
        for _ in range(2):
            pos, val = rng.integers(0, K, size=m), rng.integers(0, 4, size=m).astype(np.uint64)
            sh = (2 * pos).astype(np.uint64)
            kh = (kh & ~(np.uint64(3) << sh)) | (val << sh)
        parts.append(kh)
        labs.append(np.full(m, lab))
    rest = n - sum(len(p) for p in parts)
    parts.append(rng.integers(0, 4 ** K, size=rest, dtype=np.uint64))
    labs.append(np.full(rest, 2))
    return np.concatenate(parts).astype(np.uint32), np.concatenate(labs).astype(np.int32), [8, 7]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    idle = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    kh, lab, lens = sample(n, 6)
    ld = pitch_for(n)
    kh_d, lab_d, out_d = _ffi.DeviceBuffer.from_numpy(kh), _ffi.DeviceBuffer.from_numpy(lab), _ffi.DeviceBuffer(n * ld)
    for _ in range(5):
        hamdist_matrix_dev(kh_d.ptr, lab_d.ptr, n, K, lens, out_d.ptr, ld)
    _ffi.sync()
    if idle:
        time.sleep(idle)
    evs = [_ffi.Event() for _ in range(reps + 1)]
    evs[0].record()
    for i in range(reps):
        hamdist_matrix_dev(kh_d.ptr, lab_d.ptr, n, K, lens, out_d.ptr, ld)
        evs[i + 1].record()
    _ffi.sync()
    ms = [evs[i].elapsed_ms(evs[i + 1]) for i in range(reps)]
    groups = [ms[i:i + 20] for i in range(0, reps, 20)]
    print("per-20 medians:", " ".join(f"{statistics.median(g):.3f}" for g in groups))
    print("first 10:", " ".join(f"{x:.3f}" for x in ms[:10]))
    print(f"all: median {statistics.median(ms):.4f} min {min(ms):.4f} mean {sum(ms) / len(ms):.4f} max {max(ms):.4f}")
    fills = []
    for i in range(12):
        e0, e1 = _ffi.Event(), _ffi.Event()
        e0.record()
        out_d.zero()
        e1.record()
        _ffi.sync()
        fills.append(e0.elapsed_ms(e1))
    print(f"fill {n * ld} B: median {statistics.median(fills):.4f} min {min(fills):.4f} ms -> {n * ld / statistics.median(fills) / 1e6:.0f} GB/s; "
          f"kernel at the median: {(n * n + 5 * n) / statistics.median(ms) / 1e6:.0f} GB/s")


if __name__ == "__main__":
    main()
