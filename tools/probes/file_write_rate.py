#!/usr/bin/env python3
"""what the result directory's file system takes: GB/s of write() from one thread, of pwrite() from several threads into one
file (preallocated or not) and into several files -- the ceiling of the k{k}.pkl writers (motif_discovery.TableSaver)"""
import os
import sys
import tempfile
import threading
import time

import numpy as np

GB = 1 << 30
total, chunk = 8 * GB, 128 << 20
buf = np.random.default_rng(0).integers(0, 255, chunk, dtype=np.uint8)
d = tempfile.mkdtemp(dir=sys.argv[1] if len(sys.argv) > 1 else None)
print("dir", d, os.popen(f"df -T {d} | tail -1").read().strip())


def one_thread():
    p = os.path.join(d, "a")
    t0 = time.perf_counter()
    with open(p, "wb") as fh:
        for _ in range(total // chunk):
            fh.write(memoryview(buf))
    dt = time.perf_counter() - t0
    os.unlink(p)
    return total / dt / 1e9


def many(nt, prealloc, files):
    paths = [os.path.join(d, f"b{i}") for i in range(nt if files else 1)]
    fds = [os.open(p, os.O_CREAT | os.O_WRONLY) for p in paths]
    if prealloc:
        for fd in fds:
            os.posix_fallocate(fd, 0, total // len(fds))
    per = total // nt

    def work(i):
        fd = fds[i if files else 0]
        base = 0 if files else i * per
        for o in range(0, per, chunk):
            os.pwrite(fd, memoryview(buf), base + o)
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(nt)]
    [t.start() for t in th]
    [t.join() for t in th]
    dt = time.perf_counter() - t0
    for fd in fds:
        os.close(fd)
    for p in paths:
        os.unlink(p)
    return total / dt / 1e9


print("write(), 1 thread: %.2f GB/s" % one_thread())
for nt in (2, 4, 8):
    print(f"pwrite, {nt} threads, one file: %.2f GB/s" % many(nt, False, False))
    print(f"pwrite, {nt} threads, one file, fallocate first: %.2f GB/s" % many(nt, True, False))
    print(f"pwrite, {nt} threads, {nt} files: %.2f GB/s" % many(nt, False, True))
os.rmdir(d)
