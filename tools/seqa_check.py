#!/usr/bin/env python3
"""SEQ force kernel, producer / adder form against the classic forms (quad / pair / wide): bit identity of the gradient and
ms per force evaluation, on one GPU.  `python3 tools/seqa_check.py [--big]`"""
import os
import statistics
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from kmap_amd import _ffi, visualization as V   # noqa: E402


def forces(form, n, row0, nrows, sums_d, lds, lut, ld, reps):
    os.environ["KMAP_SEQ_FORM"] = form
    s = V.EmbedSession(n, 1, 0.01, V.EMBED_SEQ, row0=row0, nrows=nrows)
    os.environ.pop("KMAP_SEQ_FORM")
    _ffi.check(_ffi.lib().kmap_embed_set_prob_lut(s._h, sums_d.ptr, lds, _ffi.ptr(lut), len(lut)))
    s.set_coords(ld)
    g, l = _ffi.DeviceBuffer(2 * n * 4), _ffi.DeviceBuffer(8)
    g.zero()
    s.forces(g.ptr, l.ptr)
    _ffi.sync()
    out = g.to_numpy(np.float32, (2, n)), float(l.to_numpy(np.float64, (1,))[0])
    ms = []
    if reps:
        for _ in range(30):
            s.forces(g.ptr, l.ptr)
        evs = [_ffi.Event() for _ in range(reps + 1)]
        evs[0].record()
        for i in range(reps):
            s.forces(g.ptr, l.ptr)
            evs[i + 1].record()
        _ffi.sync()
        ms = [evs[i].elapsed_ms(evs[i + 1]) for i in range(reps)]
    s.close()
    g.free()
    l.free()
    return out, (statistics.median(ms) if ms else 0.0)


def fuzz(count, seed=1):
    """random shapes: n, row0, nrows (incl. 1 row, a tail shard, all rows), k (LUT length), coordinate scale (coincident points, far
    points, squared distances beyond 1e30); producer / adder form against the classic forms, bit for bit"""
    rng = np.random.default_rng(seed)
    bad = 0
    for t in range(count):
        n = int(rng.choice([rng.integers(40, 600), rng.integers(600, 5000), rng.integers(5000, 20000)]))
        mode = int(rng.integers(0, 4))
        if mode == 0:
            row0, nrows = 0, n
        elif mode == 1:
            nrows = int(rng.integers(1, min(n, 70) + 1))
            row0 = int(rng.integers(0, n - nrows + 1))
        elif mode == 2:
            nrows = int(rng.integers(1, n + 1))
            row0 = n - nrows
        else:
            row0 = int(rng.integers(0, n))
            nrows = int(rng.integers(1, n - row0 + 1))
        k = int(rng.choice([3, 6, 8, 12, 16, 20]))
        lut = V.hd_prob_lut(k, 20, 400 * k)
        lds = (n + 127) & ~127
        sums = rng.integers(0, len(lut), size=(nrows, lds), dtype=np.uint16)
        sums_d = _ffi.DeviceBuffer.from_numpy(sums)
        scale = float(rng.choice([0.01, 1.0, 5.0, 40.0, 1e3, 1e16]))
        ld = (rng.standard_normal((2, n)) * scale).astype(np.float32)
        if n > 6:
            ld[:, 5] = ld[:, 4]
        (gc, lc), _ = forces("classic", n, row0, nrows, sums_d, lds, lut, ld, 0)
        (ga, la), _ = forces("adder", n, row0, nrows, sums_d, lds, lut, ld, 0)
        same = np.array_equal(gc.view(np.uint32), ga.view(np.uint32))
        lok = (not np.isfinite(lc) and not np.isfinite(la)) or abs(la - lc) <= 1e-6 * abs(lc) + 1e-30
        if not (same and lok):
            bad += 1
            print(f"FAIL n={n} row0={row0} nrows={nrows} k={k} scale={scale}: bits {'same' if same else 'differ'} loss {lc!r} {la!r}", flush=True)
        sums_d.free()
        if (t + 1) % 25 == 0:
            print(f"fuzz {t + 1}/{count}: {bad} failures", flush=True)
    return bad


def main():
    if "--fuzz" in sys.argv:
        bad = fuzz(int(sys.argv[sys.argv.index("--fuzz") + 1]))
        print("FAILED" if bad else "fuzz ok")
        return 1 if bad else 0
    big = "--big" in sys.argv
    cases = [(1000, 100, 650), (5003, 0, 5003), (5000, 0, 5000), (4096 + 40, 7, 31), (16384 + 1029, 300, 16384 + 77)]
    if big:
        cases += [(50000, 0, 6250), (50000, 43750, 6250), (50000, 0, 50000), (49792, 0, 49792), (200000, 25000, 25000)]
    lut = V.hd_prob_lut(8, 20, 3200)
    bad = 0
    for n, row0, nrows in cases:
        rng = np.random.default_rng(n + row0)
        lds = (n + 127) & ~127
        if n * nrows <= 3e8:
            sums = rng.integers(0, 3201, size=(nrows, lds), dtype=np.uint16)
        else:                                         # big: a random block tiled (the values only feed the LUT)
            blk = rng.integers(0, 3201, size=(1024, lds), dtype=np.uint16)
            sums = np.concatenate([np.roll(blk, 17 * i, axis=1) for i in range((nrows + 1023) // 1024)])[:nrows]
        sums_d = _ffi.DeviceBuffer.from_numpy(sums)
        for scale in (5.0, 60.0):
            ld = (rng.standard_normal((2, n)) * scale).astype(np.float32)
            (gc, lc), tc = forces("classic", n, row0, nrows, sums_d, lds, lut, ld, 10)
            (ga, la), ta = forces("adder", n, row0, nrows, sums_d, lds, lut, ld, 10)
            same = np.array_equal(gc.view(np.uint32), ga.view(np.uint32))
            nz = bool(ga[:, row0:row0 + nrows].any())
            ok = same and nz and abs(la - lc) <= 1e-7 * abs(lc)
            bad += not ok
            print(f"n={n} row0={row0} nrows={nrows} scale={scale}: classic {tc:.4f} ms  adder {ta:.4f} ms  x{tc / max(ta, 1e-9):.2f}  "
                  f"bits {'same' if same else 'DIFFER (%d)' % int((gc.view(np.uint32) != ga.view(np.uint32)).sum())}  loss {lc:.9g} {la:.9g} {'ok' if ok else 'FAIL'}", flush=True)
        sums_d.free()
    print("FAILED" if bad else "all ok")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
