#!/usr/bin/env python3
"""Debug helper: counts of one synthetic input with the fine partition on (this process) against the CPU restatement."""
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))


def main():
    from kmap_amd.kmer_count import DeviceCounts
    from kmap_amd.motif_discovery import DeviceSeq
    from oracle import oracle as O
    k = int(sys.argv[1])
    n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 12000
    rng = np.random.default_rng(7)
    lens = rng.integers(20, 160, size=n_reads)
    parts, borders, pos = [], [], 0
    for L in lens:
        parts.append(rng.integers(0, 4, size=L).astype(np.uint8))
        parts.append(np.array([255], np.uint8))
        borders.append((pos, pos + L))
        pos += L + 1
    seq = np.concatenate(parts)
    borders = np.array(borders, np.int64)
    ds, dc = DeviceSeq(seq, borders), DeviceCounts()
    ds.count(dc, k, dedupe=False, merge_revcom=False)
    u, c = dc.fetch()
    ou, oc = O.count_kmers(seq, borders, k, rep_mode=True, revcom_mode=False)
    print("k", k, "n", len(seq), "uniq", len(u), len(ou), "sum", int(c.sum()), int(oc.sum()))
    got = dict(zip(u.tolist(), c.tolist()))
    want = dict(zip(ou.tolist(), oc.tolist()))
    bad = [(key, want.get(key, 0), got.get(key, 0)) for key in sorted(set(got) | set(want)) if want.get(key, 0) != got.get(key, 0)]
    print("differing keys:", len(bad))
    h = O.comp_kmer_hash(seq, k)
    nb_bits = max(10, 2 * k - 16)
    low = 2 * k - nb_bits
    for key, w, g in bad[:40]:
        where = np.nonzero(h == key)[0]
        print(f"  key {key:#x} bucket {key >> low} low {key & ((1 << low) - 1):#x} want {w} got {g} positions {where[:4].tolist()} tiles {(where[:4] // 32768).tolist()}")


if __name__ == "__main__":
    main()
