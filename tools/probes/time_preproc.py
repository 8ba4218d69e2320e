#!/usr/bin/env python3
"""`kmap preproc` on a synthetic FASTA of config C3's size (10 M x 150 bp reads, 1.6 GB of text): the native encoder
(kmap_fasta_open = count, kmap_fasta_read = encode into the caller's array) by thread count, and the whole verb.

    python tools/probes/time_preproc.py [n_reads] [read_len]
"""
import ctypes as C
import os
import shutil
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))


def write_fasta(path, n, L, seed=2):
    rng = np.random.default_rng(seed)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    with open(path, "wb") as fh:
        for r0 in range(0, n, 1_000_000):
            m = min(1_000_000, n - r0)
            rec = np.empty((m, 10 + L + 1), dtype=np.uint8)
            hdr = np.char.add(">r", np.char.zfill(np.arange(r0, r0 + m).astype(str), 7)).astype("S9")
            rec[:, :9] = np.frombuffer(hdr.tobytes(), dtype=np.uint8).reshape(m, 9)
            rec[:, 9] = 10
            rec[:, 10:10 + L] = lut[rng.integers(0, 4, size=(m, L), dtype=np.uint8)]
            rec[:, -1] = 10
            rec.tofile(fh)
    return os.path.getsize(path)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 150
    from kmap_amd import _ffi, kmer_count as kc
    lib = _ffi.lib()
    tmp = Path(tempfile.mkdtemp(prefix="kmap_preproc_"))
    try:
        fa = tmp / "reads.fa"
        t = time.perf_counter()
        size = write_fasta(fa, n, L)
        print(f"FASTA: {n} reads x {L} bp = {size / 1e9:.2f} GB written in {time.perf_counter() - t:.1f} s; host threads {os.cpu_count()}")
        for thr in ("1", "4", "16", "16"):
            os.environ["KMAP_IO_THREADS"] = thr
            h, nb, ns = C.c_void_p(), C.c_int64(0), C.c_int64(0)
            t0 = time.perf_counter()
            assert lib.kmap_fasta_open(str(fa).encode(), C.byref(h), C.byref(nb), C.byref(ns)) == 0
            t1 = time.perf_counter()
            arr = np.empty(nb.value, np.uint8)
            borders = np.empty((ns.value, 2), np.int64)
            assert lib.kmap_fasta_read(h, arr.ctypes.data_as(C.c_void_p), borders.ctypes.data_as(C.c_void_p)) == 0
            t2 = time.perf_counter()
            lib.kmap_fasta_close(h)
            print(f"threads {thr:>2}: open (count) {t1 - t0:.3f} s, read (encode) {t2 - t1:.3f} s -> {size / 1e9 / (t2 - t0):.2f} GB/s of text; "
                  f"{nb.value} bytes, {ns.value} reads, last border {borders[-1].tolist()}")
            del arr, borders
        os.environ.pop("KMAP_IO_THREADS", None)
        res = tmp / "res"
        t0 = time.perf_counter()
        kc._preproc(str(fa), str(res))
        print(f"_preproc (config, motif table, encode, two pickles): {time.perf_counter() - t0:.2f} s")
        t0 = time.perf_counter()
        a = kc.load_array_pickle(res / "input.bin.pkl")
        print(f"load_array_pickle: {time.perf_counter() - t0:.3f} s, {len(a)} bytes, first {a[:8].tolist()}")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
