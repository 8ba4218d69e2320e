#!/usr/bin/env python3
"""SEQ force kernel time vs N around the wave-granularity steps (16 rows per wave, 4 waves per block, 256 CUs): is the kernel
paying for the last, nearly empty round of blocks?"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from kmap_amd import _ffi, visualization as V  # noqa: E402


def main():
    k = 8
    lut = V.hd_prob_lut(k, 20, 400 * k)
    for n in [int(a) for a in sys.argv[1:]] or [49152, 49216, 50000, 65536]:
        rng = np.random.default_rng(2)
        lds = (n + 127) & ~127
        sums = rng.integers(0, 400 * k, size=(64, lds), dtype=np.uint16)
        sums_d = _ffi.DeviceBuffer(n * lds * 2)
        for r0 in range(0, n, 64):          # any probabilities do: the kernel's time does not depend on them
            _ffi.check(_ffi.lib().kmap_memcpy_h2d(sums_d.ptr + r0 * lds * 2, _ffi.ptr(sums), min(64, n - r0) * lds * 2, None))
        ld, ph = V._init_draws(n, 10, 7)
        sess = V.EmbedSession(n, 10, 0.01, V.EMBED_SEQ)
        _ffi.check(_ffi.lib().kmap_embed_set_prob_lut(sess._h, sums_d.ptr, lds, _ffi.ptr(lut), len(lut)))
        sess.set_coords(ld, ph)
        sess.set_jitter(np.random.normal(0, 0.01, 4096))
        sess.step(20)
        _ffi.sync()
        e0, e1 = _ffi.Event(), _ffi.Event()
        e0.record()
        sess.step(20)
        e1.record()
        _ffi.sync()
        ms = e0.elapsed_ms(e1) / 20
        print(f"N={n}: {ms:.3f} ms/iter, {n / 64 / 256:.3f} blocks per CU, {ms / (n * n) * 1e9:.4f} ps/pair")
        sess.close()
        sums_d.free()


if __name__ == "__main__":
    main()
