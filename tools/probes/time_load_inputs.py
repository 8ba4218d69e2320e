#!/usr/bin/env python3
"""How fast can input.bin.pkl (1.5 GB at C3) reach HBM?  pickle.load + pageable copy vs a memory map of the pickle's payload."""
import mmap
import os
import pickle
import sys
import tempfile
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np


def main():
    from kmap_amd import _ffi
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_510_000_000
    arr = np.random.default_rng(1).integers(0, 4, n, dtype=np.uint8)
    d = tempfile.mkdtemp(prefix="kmap_load_")
    path = os.path.join(d, "input.bin.pkl")
    with open(path, "wb") as fh:
        pickle.dump(arr, fh, protocol=4)
    from kmap_amd.kmer_count import locate_pickled_array
    off, dt, shape = locate_pickled_array(path, 1 << 20)
    print("payload offset", off, dt, shape)
    dev = _ffi.DeviceBuffer(n)
    _ffi.sync()
    for rep in range(3):
        t0 = time.perf_counter()
        with open(path, "rb") as fh:
            a = pickle.load(fh)
        t1 = time.perf_counter()
        _ffi.check(_ffi.lib().kmap_memcpy_h2d(dev.ptr, _ffi.ptr(a), n, None))
        _ffi.sync()
        t2 = time.perf_counter()
        del a
        t3 = time.perf_counter()
        print(f"pickle.load {t1 - t0:.3f}  h2d {t2 - t1:.3f}  free {t3 - t2:.3f}")
    for flags, name in ((mmap.MAP_PRIVATE, "private"), (mmap.MAP_SHARED, "shared"), (mmap.MAP_PRIVATE | mmap.MAP_POPULATE, "private+populate")):
        for rep in range(2):
            t0 = time.perf_counter()
            fd = os.open(path, os.O_RDONLY)
            mm = mmap.mmap(fd, 0, flags=flags, prot=mmap.PROT_READ)
            v = np.frombuffer(mm, np.uint8, count=n, offset=off)
            t1 = time.perf_counter()
            _ffi.check(_ffi.lib().kmap_memcpy_h2d(dev.ptr, v.ctypes.data, n, None))
            _ffi.sync()
            t2 = time.perf_counter()
            back = dev.to_numpy(np.uint8, (1 << 20,), offset=n - (1 << 20))
            ok = bool((back == arr[-(1 << 20):]).all())
            del v
            mm.close()
            os.close(fd)
            t3 = time.perf_counter()
            print(f"mmap {name}: map {t1 - t0:.3f}  h2d {t2 - t1:.3f}  unmap {t3 - t2:.3f}  ok={ok}")
    # chunked pread into one re-used pageable buffer
    for chunk in (16 << 20, 64 << 20):
        buf = np.empty(chunk, np.uint8)
        t0 = time.perf_counter()
        fd = os.open(path, os.O_RDONLY)
        pos = 0
        while pos < n:
            m = min(chunk, n - pos)
            got = os.preadv(fd, [memoryview(buf)[:m]], off + pos)
            assert got == m
            _ffi.check(_ffi.lib().kmap_memcpy_h2d(dev.ptr + pos, _ffi.ptr(buf), m, None))
            pos += m
        _ffi.sync()
        os.close(fd)
        print(f"pread chunks of {chunk >> 20} MiB: {time.perf_counter() - t0:.3f}")


if __name__ == "__main__":
    main()
