"""Seeded synthetic reads in the style of the reference's test generator (tests/kmap_tests.py:75-114), fixed length:
40 % of the reads carry motif A, 40 % motif B at a uniform position with 5 % per-base substitution, 20 % pure random;
no N.  Emitted directly as the preprocessed array contract (uint8 + 255 separators, (n_seq,2) int64 borders)."""
import pickle
from pathlib import Path

import numpy as np

MOTIF_A, MOTIF_B = "AATCGATAGC", "AGGACCTACGTAC"
_CODE = {"A": 0, "C": 1, "G": 2, "T": 3}


def synth_reads(n_reads, read_len, seed, motifs=(MOTIF_A, MOTIF_B), fractions=(0.4, 0.4), mutation_rate=0.05, chunk=1 << 20):
    rng = np.random.Generator(np.random.PCG64(seed))
    row = read_len + 1
    seq = np.empty(n_reads * row, dtype=np.uint8)
    view = seq.reshape(n_reads, row)
    bounds = np.cumsum([int(n_reads * f) for f in fractions])
    for lo in range(0, n_reads, chunk):
        hi = min(n_reads, lo + chunk)
        m = hi - lo
        block = rng.integers(0, 4, size=(m, read_len), dtype=np.uint8)
        start = 0
        for mi, motif in enumerate(motifs):
            a, b = max(lo, start), min(hi, int(bounds[mi]))
            start = int(bounds[mi])
            if b <= a:
                continue
            cnt, L = b - a, len(motif)
            pos = rng.integers(0, read_len - L + 1, size=cnt)
            keep = rng.random((cnt, L)) > mutation_rate
            codes = np.array([_CODE[c] for c in motif], dtype=np.uint8)
            rows = np.arange(a - lo, b - lo)[:, None]
            cols = pos[:, None] + np.arange(L)[None, :]
            cur = block[rows, cols]
            block[rows, cols] = np.where(keep, codes[None, :], cur)
        view[lo:hi, :read_len] = block
    view[:, read_len] = 255
    starts = np.arange(n_reads, dtype=np.int64) * row
    borders = np.stack([starts, starts + read_len], axis=1)
    return seq, borders


def synth_reads_dev(n_reads, read_len, seed, motifs=(MOTIF_A, MOTIF_B), fractions=(0.4, 0.4), mutation_rate=0.05, keep_raw=False):
    """The same kind of reads generated in HBM (csrc/synth.hip: counter-based, NOT the numpy stream of synth_reads) for the
    configurations that are too large to build on the host inside a benchmark (C5: 50 M x 300 bp = 15 GB).
    Returns a DeviceSeq (packed reads + borders resident); keep_raw=True returns (DeviceSeq, raw) where raw() fetches
    the uint8 bytes [lo, hi) of the array as generated (for spot checks by tests and bench.py) until raw.free() is called --
    the 1 B / position array otherwise leaves HBM as soon as it is packed."""
    import ctypes as C
    from . import _ffi
    from .motif_discovery import DeviceSeq
    n = n_reads * (read_len + 1)
    raw = _ffi.DeviceBuffer(max(n, 16))
    borders = _ffi.DeviceBuffer(max(n_reads, 1) * 16)
    codes = np.array([_CODE[c] for m in motifs for c in m], np.uint8)
    lens = np.array([len(m) for m in motifs], np.int32)
    fr = np.array(fractions, np.float64)
    _ffi.check(_ffi.lib().kmap_synth_reads_dev(raw.ptr, borders.ptr, n_reads, read_len, C.c_uint64(seed), _ffi.ptr(codes) if len(codes) else None,
                                               _ffi.ptr(lens) if len(lens) else None, _ffi.ptr(fr) if len(fr) else None, len(motifs),
                                               float(mutation_rate), None))
    if not keep_raw:
        return DeviceSeq.from_device(raw, n, borders, n_reads, read_len)
    copy = _ffi.DeviceBuffer(max(n, 16))
    _ffi.check(_ffi.lib().kmap_memcpy_d2d(copy.ptr, raw.ptr, n, None))
    ds = DeviceSeq.from_device(raw, n, borders, n_reads, read_len)

    class _Raw:
        def __call__(self, lo, hi):
            return copy.to_numpy(np.uint8, (hi - lo,), offset=lo)

        def free(self):
            copy.free()
    return ds, _Raw()


def write_res_dir(res_dir, seq, borders, overrides=None, fasta_name="synthetic.fa"):
    """Create a res_dir as `kmap preproc` would leave it (config.toml, motif_def_table.csv, the two pickles)."""
    from ._toml import dump_toml
    from .kmer_count import FileNameDict, MotifDef, dump_pickle_nocopy, gen_motif_def_dict, read_default_config_file
    res = Path(res_dir)
    res.mkdir(parents=True, exist_ok=True)
    cfg = read_default_config_file()
    cfg["general"]["input_fasta_file"] = str(res / fasta_name)
    cfg["general"]["res_dir"] = str(res)
    for sec, kv in (overrides or {}).items():
        cfg[sec].update(kv)
    dump_toml(cfg, res / FileNameDict["config_file"])
    table = gen_motif_def_dict(cfg)
    with open(res / FileNameDict["motif_def_file"], "w+") as fh:
        fh.write(MotifDef.get_field_names() + "\n")
        for k in sorted(k for k in table if isinstance(k, int)):
            fh.write(str(table[k]) + "\n")
    with open(res / FileNameDict["processed_fasta_file"], "wb") as fh:
        dump_pickle_nocopy(seq, fh, protocol=4)
    with open(res / FileNameDict["processed_fasta_seqboarder_file"], "wb") as fh:
        dump_pickle_nocopy(borders, fh, protocol=4)
    return cfg
