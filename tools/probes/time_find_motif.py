#!/usr/bin/env python3
"""Wall-time split of find_motif's steps for one k at C3 scale (count, fetch, pickle, top-k, ball mass, mask, recount)."""
import argparse
import pickle
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", type=int, nargs="+", default=[14, 15, 16])
    ap.add_argument("--reads", type=int, default=10_000_000)
    args = ap.parse_args()
    from kmap_amd import _ffi, synth
    from kmap_amd.kmer_count import DeviceCounts, gen_motif_def_dict, read_default_config_file, revcom_hash
    from kmap_amd.motif_discovery import DeviceSeq
    seq, borders = synth.synth_reads(args.reads, 150, 2)
    ds = DeviceSeq(seq, borders)
    mdd = gen_motif_def_dict(read_default_config_file())
    dc = DeviceCounts()

    def T(label, fn):
        _ffi.sync()
        t0 = time.perf_counter()
        r = fn()
        _ffi.sync()
        print(f"  {label:28s} {1e3 * (time.perf_counter() - t0):9.1f} ms", flush=True)
        return r
    for k in args.k:
        print(f"k = {k}")
        ds.reset()
        T("count (dedupe, first round)", lambda: ds.count(dc, k, dedupe=True, merge_revcom=True))
        print(f"  n_uniq = {dc.n_uniq}")
        T("count again (dedupe)", lambda: ds.count(dc, k, dedupe=True, merge_revcom=True))
        T("count (no dedupe)", lambda: ds.count(dc, k, dedupe=False, merge_revcom=True))
        T("total", dc.total)
        u, c = T("fetch", dc.fetch)
        import ctypes as C
        st = _ffi.vp()
        _ffi.check(_ffi.lib().kmap_stream_create(C.byref(st)))
        u2, c2 = T("fetch (own stream, staged)", lambda: dc.fetch(stream=st.value))
        assert np.array_equal(u, u2) and np.array_equal(c, c2)
        del u2, c2
        with tempfile.TemporaryDirectory() as td:
            T("pickle dump (protocol 5)", lambda: pickle.dump([k, u, c], open(Path(td) / "k.pkl", "wb"), protocol=5))
        _, cand, _ = T("topk(5)", lambda: dc.topk(5))
        T("hamball_mass", lambda: dc.hamball_mass(cand, mdd[k].max_ham_dist, True))
        cons = np.array([cand[0], revcom_hash(cand[0], k)])
        T("mask", lambda: ds.mask(k, cons, np.array([mdd[k].max_ham_dist] * 2)))
        T("recount (no dedupe)", lambda: ds.count(dc, k, dedupe=False, merge_revcom=True))
        del u, c
    dc.close()
    ds.close()


if __name__ == "__main__":
    main()
