#!/usr/bin/env python3
"""Only the neighbour-sums stage at N = 50 000, k = 8 (labels: 41 % / 17 % / noise, finals of 8 and 7 bases like the C3 hand-over;
random neighbour table): workload for kernel-time passes on knn_sums_profile_kernel."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))


def main():
    from kmap_amd import _ffi, visualization as V
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000
    rng = np.random.default_rng(3)
    kh = rng.integers(0, 4 ** 8, n).astype(np.uint32)
    lab = np.where(rng.random(n) < 0.41, 0, np.where(rng.random(n) < 0.29, 1, 2)).astype(np.int32)
    lab = np.sort(lab)
    lens = [8, 7]
    nb = np.empty((n, 20), np.int32)
    for g in range(3):                                   # neighbours mostly inside the own label, as after the real selection
        idx = np.nonzero(lab == g)[0]
        nb[idx] = rng.choice(idx, size=(len(idx), 20))
    kh_d, lab_d, nb_d = _ffi.DeviceBuffer.from_numpy(kh), _ffi.DeviceBuffer.from_numpy(lab), _ffi.DeviceBuffer.from_numpy(nb)
    lds = (n + 127) & ~127
    sums_d = _ffi.DeviceBuffer(n * lds * 2)
    for rep in range(5):
        _ffi.sync()
        t0 = time.perf_counter()
        V.knn_sums_kmers_dev(kh_d.ptr, lab_d.ptr, n, 8, lens, nb_d, 20, out=sums_d.ptr)
        _ffi.sync()
        print(f"knn_sums N={n}: {1e3 * (time.perf_counter() - t0):.3f} ms", flush=True)
    s = sums_d.to_numpy(np.uint16, (n, lds))
    print("checksum", int(s[:, :n].astype(np.uint64).sum()), int(s[123, 4567]), int(s[49999 % n, 0]))


if __name__ == "__main__":
    main()
