"""Host side of the k-mer counting path: the reference's operator interface
(/root/reference/src/kmap/kmer_count.py) re-exposed over the HIP C ABI.

Function names, argument meaning and error behaviour follow the reference so that its callers
(and tests written against it) run unchanged; every array operator launches a hand-written
gfx950 kernel through kmap_amd._ffi -- there is no numpy/CPU implementation of the kernels here.
Host-only pieces (dtype rules, scalar hash helpers, FASTA encoding, TOML/CSV contracts) are plain
Python like the reference's.
"""
import ctypes as C
import gzip
import pickle
from dataclasses import dataclass, fields
from importlib.resources import files
from pathlib import Path
from typing import Dict, List, Tuple

import numpy as np

from . import _ffi
from ._ffi import check, ptr

# file-name contract of the result directory (reference kmer_count.py:26-53)
FileNameDict = {
    "default_config_file": "default_config.toml",
    "config_file": "config.toml",
    "default_motif_def_file": "default_motif_def_table.csv",
    "motif_def_file": "motif_def_table.csv",
    "processed_fasta_file": "input.bin.pkl",
    "processed_fasta_seqboarder_file": "input.seqboarder.bin.pkl",
    "motif_pos_density_file": "motif_pos_density.np.pkl",
    "motif_pos_density_plot_dir": "motif_pos_density",
    "kmer_count_dir": "kmer_count",
    "conseq_similarity_dir": "conseq_similarity",
    "co_occur_dir": "co_occurence",
    "co_occur_dist_mat_file": "co_occurence_motif_dist_mat.tsv",
    "co_occur_dist_data_file": "co_occurence_motif_dist_data.txt",
    "co_occur_mat_file": "co_occurence_mat.tsv",
    "co_occur_mat_norm_file": "co_occurence_mat.norm.tsv",
    "co_occur_network_fig": "co_occur_network.pdf",
    "motif_occurence_file": "final.motif_occurence.csv",
    "hamball_dir": "hamming_balls",
    "candidate_conseq_file": "candidate_conseq.csv",
    "final_conseq_file": "final_conseq.txt",
    "final_conseq_info_file": "final_conseq.info.csv",
    "sample_kmer_pkl_file": "sample_kmers.pkl",
    "sample_kmer_txt_file": "sample_kmers.tsv",
    "sample_kmer_hamdist_mat_file": "sample_kmer_hamdist_mat.pkl",
    "ld_data_file": "low_dim_data.tsv",
    "ld_fig_file_stem": "ld_data",
}

MISSING_VAL = 255
_BASES = "ACGT"
_ENC = np.full(256, MISSING_VAL, dtype=np.uint8)
for _i, _b in enumerate(_BASES):
    _ENC[ord(_b)] = _i


# ---- dtype rules (reference kmer_count.py:351-370) ----------------------------------------------
def get_cnt_dtype(kmer_len: int):
    return np.int32 if kmer_len < 16 else np.int64


def get_hash_dtype(kmer_len):
    if 0 < kmer_len < 16:
        return np.uint32
    elif kmer_len < 32:
        return np.uint64
    raise Exception(f"max_kmer_len=31, kmer_len={kmer_len} is greater the maximum value.")


def get_invalid_hash(dtype):
    return dtype(np.iinfo(dtype).max)


# ---- scalar helpers (reference kmer_count.py:238-268,416-446,626-640) ---------------------------
def kmer2hash(kmer: str) -> np.uint64:
    assert len(kmer) < 32, "kmer should be shorted than 32 bases"
    kh = 0
    for base in kmer:
        kh = (kh << 2) | _BASES.index(base)
    return np.uint64(kh)


def hash2kmer(hashkey, k: int) -> str:
    hk = int(hashkey)
    return "".join(_BASES[(hk >> (2 * (k - 1 - i))) & 3] for i in range(k))


def hashes2kmers(hash_arr, k: int) -> np.ndarray:
    """hash2kmer for a whole array: numpy '<U{k}' strings"""
    h = np.asarray(hash_arr, dtype=np.uint64).reshape(-1, 1)
    shifts = (2 * np.arange(k - 1, -1, -1, dtype=np.uint64)).reshape(1, -1)
    codes = ((h >> shifts) & np.uint64(3)).astype(np.uint8)
    letters = np.frombuffer(b"ACGT", dtype="S1")[codes]
    return letters.view(f"S{k}").reshape(-1).astype(f"U{k}") if k else np.array([""] * len(h))


def revcom_hash(in_hash, kmer_len: int):
    dt = get_hash_dtype(kmer_len)
    com = ((1 << (2 * kmer_len)) - 1 - int(in_hash)) % (1 << (8 * np.dtype(dt).itemsize))
    out = 0
    for _ in range(kmer_len):
        out = (out << 2) | (com & 3)
        com >>= 2
    return dt(out)


def reverse_complement(seq):
    return seq[::-1].translate(str.maketrans("ACGT", "TGCA"))


def arr2dna(dna_np_arr: np.ndarray) -> str:
    lut = np.full(256, ord("?"), np.uint8)
    lut[:4] = np.frombuffer(b"ACGT", np.uint8)
    lut[MISSING_VAL] = ord("N")
    return lut[np.asarray(dna_np_arr, np.uint8)].tobytes().decode()


def dna2arr(dna_str, dtype=np.uint8, append_missing_val_flag=True) -> np.ndarray:
    """A/C/G/T -> 0..3, anything else -> 255, optional trailing 255 separator (upper case expected)."""
    codes = _ENC[np.frombuffer(dna_str.encode("latin-1"), dtype=np.uint8)].astype(dtype)
    if append_missing_val_flag:
        codes = np.append(codes, dtype(MISSING_VAL))
    return codes


# ---- array operators: HIP kernels ------------------------------------------------------------
def _as_hash(a, kmer_len):
    dt = get_hash_dtype(kmer_len)
    a = np.asarray(a)
    if a.dtype != dt:
        a = a.astype(dt)
    return np.ascontiguousarray(a), dt


def comp_kmer_hash(seq_np_arr: np.ndarray, kmer_len: int) -> np.ndarray:
    """k-mer hash at every array index (reference comp_kmer_hash_taichi, kmer_count.py:449-473)."""
    dt = get_hash_dtype(kmer_len)
    seq = np.ascontiguousarray(seq_np_arr, dtype=np.uint8)
    out = np.empty(len(seq), dtype=dt)
    fn = _ffi.lib().kmap_hash_kmers_u32 if dt == np.uint32 else _ffi.lib().kmap_hash_kmers_u64
    check(fn(ptr(seq), len(seq), kmer_len, ptr(out)))
    return out


comp_kmer_hash_taichi = comp_kmer_hash   # the reference's name for the same operator


def remove_duplicate_hash_per_seq(hash_arr: np.ndarray, boarder_mat: np.ndarray, invalid_hash=None) -> np.ndarray:
    """In place: per read keep the first occurrence of each hash (reference kmer_count.py:743-760)."""
    assert boarder_mat.shape[1] == 2
    assert hash_arr.dtype in (np.uint32, np.uint64) and hash_arr.flags.c_contiguous
    b = np.ascontiguousarray(boarder_mat, dtype=np.int64)
    fn = _ffi.lib().kmap_dedupe_per_read_u32 if hash_arr.dtype == np.uint32 else _ffi.lib().kmap_dedupe_per_read_u64
    check(fn(ptr(hash_arr), len(hash_arr), ptr(b), len(b)))
    return hash_arr


class DeviceCounts:
    """Owns a kmap_counts handle: unique k-mers + counts resident in HBM.
    Multi-GPU, large tables (distributed.make_dist_device_seq, bins owned by key range): the handle may hold only THIS rank's key
    range of the table (`_shard`, a distributed.CountShard); n_uniq is then the global number of unique k-mers, and total / topk /
    hamball_mass -- everything find_motif asks of the table (reference motif_discovery.py:648,661-673) -- are local partials
    combined by one tiny collective each, so no rank ever receives the whole (k-mer, count) list.  `_full` (optional, on the rank
    that writes k{k}.pkl): a second handle holding the gathered table."""

    def __init__(self):
        h = _ffi.vp()
        check(_ffi.lib().kmap_counts_create(C.byref(h)))
        self._h = h.value
        self.k = 0
        self.n_uniq = 0
        self._shard = None
        self._full = None

    def _unshard(self):
        self._shard = None
        if self._full is not None:
            self._full.close()
            self._full = None

    def run_hashes(self, hash_dev_ptr, n, k, merge_revcom, stream=None):
        self._unshard()
        nu = _ffi.i64(0)
        check(_ffi.lib().kmap_counts_run_hashes_dev(self._h, hash_dev_ptr, n, k, int(merge_revcom), C.byref(nu), stream))
        self.k, self.n_uniq = k, nu.value
        return self.n_uniq

    def run_seq(self, seq_dev_ptr, n, borders_dev_ptr, n_seq, k, dedupe, merge_revcom, stream=None):
        self._unshard()
        nu = _ffi.i64(0)
        check(_ffi.lib().kmap_counts_run_seq_dev(self._h, seq_dev_ptr, n, borders_dev_ptr, n_seq, k, int(dedupe),
                                                 int(merge_revcom), C.byref(nu), stream))
        self.k, self.n_uniq = k, nu.value
        return self.n_uniq

    def _fetch_local(self, n, stream=None):
        u = np.empty(n, get_hash_dtype(self.k))
        c = np.empty(n, get_cnt_dtype(self.k))
        if n:
            if stream is None:
                check(_ffi.lib().kmap_counts_fetch(self._h, ptr(u), ptr(c)))
            else:
                check(_ffi.lib().kmap_counts_fetch_stream(self._h, ptr(u), ptr(c), stream))
        return u, c

    def fetch(self, stream=None):
        """(unique hashes, counts) as the reference's numpy arrays.  stream: a non-default stream handle -> the copy runs there
        through pinned staging buffers (a background thread can then drain this table while the default stream keeps working).
        Sharded table: COLLECTIVE -- every rank must call it, and every rank takes the same path (the shards are all-gathered on
        the host) whether or not it also holds a gathered copy; the rank that does reads that copy with fetch_full()."""
        if self._shard is not None:
            return self._shard.gather_host(*self._fetch_local(self._shard.n_local))
        return self._fetch_local(self.n_uniq, stream)

    def fetch_full(self, stream=None):
        """local, no collective: the gathered copy on the rank that holds one (the writer of k{k}.pkl), else None"""
        return None if self._full is None else self._full.fetch(stream)

    def total(self):
        t = _ffi.i64(0)
        check(_ffi.lib().kmap_counts_total(self._h, C.byref(t)))
        return t.value if self._shard is None else self._shard.sum_int(t.value)

    def topk(self, top_k):
        """device top-k by count (largest first, ties by lowest index): (indices, hashes, counts)"""
        idx, kh, cnt = np.empty(top_k, np.int64), np.empty(top_k, np.uint64), np.empty(top_k, np.int64)
        m = _ffi.i32(0)
        check(_ffi.lib().kmap_counts_topk(self._h, top_k, ptr(idx), ptr(kh), ptr(cnt), C.byref(m)))
        if self._shard is not None:    # the ranks' candidates (<= world x top_k) merged by the same rule: count descending, index ascending
            idx, kh, cnt = self._shard.merge_topk(top_k, idx[:m.value], kh[:m.value], cnt[:m.value])
            return idx, kh.astype(get_hash_dtype(self.k)), cnt
        return idx[:m.value], kh[:m.value].astype(get_hash_dtype(self.k)), cnt[:m.value]

    def hamball_mass(self, cands, radius, revcom=True):
        cands = np.ascontiguousarray(cands, dtype=np.uint64)
        out = np.zeros(len(cands), np.float64)
        check(_ffi.lib().kmap_counts_hamball_mass(self._h, ptr(cands), len(cands), int(radius), int(revcom), ptr(out)))
        return out if self._shard is None else self._shard.sum_f64(out)   # integer-valued float64 partials: exact in any order

    def close(self):
        if self._full is not None:
            self._full.close()
            self._full = None
        if self._h:
            _ffi.lib().kmap_counts_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def count_uniq_hash(hash_arr: np.ndarray, kmer_len):
    """Sorted unique hashes + counts, invalid hash dropped (reference kmer_count.py:476-491)."""
    h, dt = _as_hash(hash_arr, kmer_len)
    dev = _ffi.DeviceBuffer.from_numpy(h)
    dc = DeviceCounts()
    try:
        dc.run_hashes(dev.ptr, len(h), kmer_len, merge_revcom=False)
        return dc.fetch()
    finally:
        dc.close()
        dev.free()


def _ham(kh_arr, consensus_kh, kmer_len, shift_bits, clen):
    h, dt = _as_hash(kh_arr, kmer_len)
    out = np.empty(len(h), dtype=np.uint8)
    fn = _ffi.lib().kmap_hamdist_1vN_u32 if dt == np.uint32 else _ffi.lib().kmap_hamdist_1vN_u64
    check(fn(ptr(h), len(h), int(dt(consensus_kh)), shift_bits, clen, ptr(out)))
    return out


def cal_hamming_dist(kh_arr: np.ndarray, consensus_kh, kmer_len: int) -> np.ndarray:
    """Hamming distance of every hash to one consensus (reference kmer_count.py:494-515)."""
    return _ham(kh_arr, consensus_kh, kmer_len, 0, kmer_len)


def cal_hamming_dist_head(kh_arr, consensus_kh, kmer_len: int, consensus_len: int) -> np.ndarray:
    """Consensus (<= k bases) against the first consensus_len bases (reference kmer_count.py:518-546)."""
    assert consensus_len <= kmer_len
    return _ham(kh_arr, consensus_kh, kmer_len, 2 * (kmer_len - consensus_len), consensus_len)


def cal_hamming_dist_tail(kh_arr, consensus_kh, kmer_len: int, consensus_len: int) -> np.ndarray:
    """Consensus against the last consensus_len bases (reference kmer_count.py:549-577)."""
    assert consensus_len <= kmer_len
    return _ham(kh_arr, consensus_kh, kmer_len, 0, consensus_len)


def mask_input(seq_np_arr: np.ndarray, kmer_len: int, consensus_kh_arr, max_hamball_dist_arr):
    """In place: overwrite every occurrence of the consensuses' Hamming balls with 255
    (reference kmer_count.py:580-610)."""
    assert seq_np_arr.dtype == np.uint8 and seq_np_arr.flags.c_contiguous
    if not seq_np_arr.flags.writeable:
        raise ValueError("mask_input works in place and needs a writeable array (copy a memory-mapped input first)")
    cons = np.ascontiguousarray(consensus_kh_arr, dtype=np.uint64)
    rad = np.ascontiguousarray(max_hamball_dist_arr, dtype=np.int32)
    assert len(cons) == len(rad)
    check(_ffi.lib().kmap_mask_hamball(ptr(seq_np_arr), len(seq_np_arr), kmer_len, ptr(cons), ptr(rad), len(cons)))
    return seq_np_arr


def get_revcom_hash_arr(in_hash_arr: np.ndarray, kmer_len: int) -> np.ndarray:
    """Reverse-complement hash of every entry (reference kmer_count.py:613-623)."""
    h, dt = _as_hash(in_hash_arr, kmer_len)
    out = np.empty_like(h)
    fn = _ffi.lib().kmap_revcom_u32 if dt == np.uint32 else _ffi.lib().kmap_revcom_u64
    check(fn(ptr(h), len(h), kmer_len, ptr(out)))
    return out


def merge_revcom(uniq_kmer_hash_arr: np.ndarray, uniq_kh_cnt_arr: np.ndarray, kmer_len: int,
                 keep_lower_hash_flag=True) -> Tuple:
    """Sum the counts of reverse-complement pairs and keep one key per pair (reference
    kmer_count.py:643-685, including: palindromes double, a kept key whose partner is absent is
    replaced in place without re-sorting).  Like the reference: revcom on device, set logic in numpy.
    (The fused device path -- DeviceCounts.run_seq(..., merge_revcom=True) -- does the same in bin space.)"""
    kh = np.asarray(uniq_kmer_hash_arr)
    cnt = np.asarray(uniq_kh_cnt_arr)
    rc = get_revcom_hash_arr(kh, kmer_len).astype(kh.dtype)
    pos = np.searchsorted(kh, rc)
    pos_c = np.minimum(pos, max(len(kh) - 1, 0))
    has_partner = (kh[pos_c] == rc) if len(kh) else np.zeros(0, bool)
    merged = cnt + np.where(has_partner, cnt[pos_c], 0).astype(cnt.dtype)
    worse = (kh > rc) if keep_lower_hash_flag else (kh < rc)
    keep = ~(has_partner & worse)
    out_kh = np.where(worse, rc, kh)[keep]
    return out_kh, merged[keep]


def mask_ham_ball(seq_np_arr: np.ndarray, motif_def_dict: dict, consensus_seq_list: List[str],
                  max_ham_dist_list: List[int] = ()) -> np.ndarray:
    """Mask user-supplied consensus Hamming balls, grouped by length (reference kmer_count.py:688-723)."""
    lens = [len(s) for s in consensus_seq_list]
    if len(max_ham_dist_list) == 0:
        max_ham_dist_list = [motif_def_dict[n].max_ham_dist for n in lens]
    assert len(max_ham_dist_list) == len(consensus_seq_list)
    for k in sorted(set(lens)):
        sel = [i for i, n in enumerate(lens) if n == k]
        seq_np_arr = mask_input(seq_np_arr, k, np.array([kmer2hash(consensus_seq_list[i]) for i in sel]),
                                np.array([max_ham_dist_list[i] for i in sel]))
    return seq_np_arr


# ---- motif definition table + config (reference kmer_count.py:104-136,221-235,726-740) -----------
@dataclass
class MotifDef:
    kmer_len: int
    p_uniform: float
    max_ham_dist: int
    ratio_mu: float
    ratio_std: float
    ratio_cutoff: float

    @classmethod
    def get_field_names(cls):
        return ",".join(f.name for f in fields(cls))

    def __str__(self):
        return ",".join(str(getattr(self, f.name)) for f in fields(self))


def norm_ppf(q, loc, scale):
    """scipy.stats.norm.ppf(q, loc=loc, scale=scale) for scalars, without importing scipy.stats (0.5 s in a fresh process, of which this
    package needs two functions): the same scipy.special call behind it (norm._ppf = special.ndtri; rv_continuous.ppf returns
    _ppf(q) * scale + loc, NaN for scale <= 0 / NaN arguments / q outside [0, 1]) -- bit-identical, asserted by test_host_logic.py"""
    from scipy.special import ndtri
    q, loc, scale = float(q), float(loc), float(scale)
    if not (scale > 0.0) or loc != loc or not (0.0 <= q <= 1.0):
        return np.float64(np.nan)
    return np.float64(ndtri(q) * scale + loc)


def norm_logsf(x, loc, scale):
    """scipy.stats.norm.logsf(x, loc=loc, scale=scale) for scalars: special.log_ndtr(-((x - loc) / scale)), NaN for scale <= 0 / NaN
    arguments (norm._logsf(x) = _norm_logcdf(-x) = special.log_ndtr(-x))"""
    from scipy.special import log_ndtr
    x, loc, scale = float(x), float(loc), float(scale)
    if not (scale > 0.0) or loc != loc or x != x:
        return np.float64(np.nan)
    z = (x - loc) / scale
    if z == float("-inf"):
        return np.float64(0.0)                      # scipy assigns log(1) = +0.0 below the support; log_ndtr(inf) is -0.0
    return np.float64(log_ndtr(-z))


_TABLE_NEED = ("kmer_len", "max_ham_dist", "p_uniform", "ratio_mu", "ratio_std")


def _read_motif_table_plain(motif_def_file):
    """The rows of a motif table as (kmer_len, max_ham_dist, p_uniform, ratio_mu, ratio_std) WITHOUT pandas -- or None when the file
    holds anything on which `pandas.read_csv` (the reference's reader, kmer_count.py:719-740) and Python's float() could disagree.
    pandas' C parser is not correctly rounded (30 % of 17-digit strings come back up to 13 ulp off, long runs of leading zeros lose
    digits), but a plain decimal of at most 15 digit characters is one exact integer divided by one exact power of ten in both (0
    mismatches in 2 M random tokens); the packaged table and the one `preproc` writes are of that kind.  Everything else -- exponents,
    longer numbers, other NA spellings, quotes, spaces, ragged rows -- returns None and the caller takes pandas."""
    import csv
    import re
    try:
        with open(motif_def_file, newline="") as fh:
            text = fh.read()
    except (OSError, UnicodeDecodeError):
        return None
    if '"' in text:                                   # quoted fields: pandas' business (csv.reader would unquote them silently)
        return None
    try:
        rows = list(csv.reader(text.splitlines(), quoting=csv.QUOTE_NONE))
    except csv.Error:
        return None
    if not rows:
        return None
    header = rows[0]
    if len(set(header)) != len(header) or any(c not in header for c in _TABLE_NEED):
        return None
    col = {c: header.index(c) for c in _TABLE_NEED}
    plain_int, plain_dec = re.compile(r"^[0-9]{1,9}$"), re.compile(r"^-?[0-9]*\.?[0-9]*$")
    out, fractional = [], False
    for r in rows[1:]:
        if not r:
            continue                                  # pandas skips blank lines
        if len(r) != len(header):
            return None
        vals = []
        for c in _TABLE_NEED:
            tok = r[col[c]]
            if c in ("kmer_len", "max_ham_dist"):
                if not plain_int.match(tok):
                    return None
                vals.append(int(tok))
            elif tok in ("", "nan", "NaN"):
                vals.append(float("nan"))
            else:
                n_digits = sum(ch.isdigit() for ch in tok)
                if not plain_dec.match(tok) or not 1 <= n_digits <= 15:
                    return None
                fractional = fractional or "." in tok
                vals.append(float(tok))
        out.append(tuple(vals))
    # pandas hands iterrows() one float64 row as soon as any column is float: with an all-integer table it would hand out integers
    return out if fractional else None


def init_motif_def_dict(motif_def_file, p_value_cutoff=1e-10) -> dict:
    """reference kmer_count.py:719-740 (`pd.read_csv(...).iterrows()` + `norm.ppf`).  A fresh process pays 0.35 s for importing pandas
    and 0.5 s for scipy.stats -- most of `scan_motif` / `preproc` at the reference's default input size -- so tables on which the plain
    reader provably agrees with pandas are read without it (_read_motif_table_plain), and the two normal-distribution functions come
    from scipy.special (norm_ppf / norm_logsf).  test_host_logic.py holds both to the pandas / scipy.stats path bit for bit."""
    table = {"p_value_cutoff": p_value_cutoff}
    rows = _read_motif_table_plain(motif_def_file)
    if rows is None:
        import pandas as pd
        rows = [(int(row["kmer_len"]), int(row["max_ham_dist"]), row["p_uniform"], row["ratio_mu"], row["ratio_std"])
                for _, row in pd.read_csv(motif_def_file).iterrows()]
    for k, max_ham_dist, p_uniform, ratio_mu, ratio_std in rows:
        cutoff = norm_ppf(1 - p_value_cutoff, loc=ratio_mu, scale=ratio_std)
        table[k] = MotifDef(k, p_uniform, max_ham_dist, ratio_mu, ratio_std, cutoff)
    return table


def _pkg_file(name):
    return files(__package__).joinpath(name)


def read_default_config_file(debug=False):
    from ._toml import load_toml
    cfg = load_toml(_pkg_file(FileNameDict["default_config_file"]))
    if debug:
        print(cfg)
    return cfg


def gen_motif_def_dict(config_dict: dict, debug=False) -> Dict:
    src = config_dict["motif_discovery"]["motif_def_file"]
    cutoff = config_dict["motif_discovery"]["p_value_cutoff"]
    if src == "default":
        table = init_motif_def_dict(_pkg_file(FileNameDict["default_motif_def_file"]), p_value_cutoff=cutoff)
    else:
        assert Path(src).exists()
        table = init_motif_def_dict(src, p_value_cutoff=cutoff)
    if debug:
        print(table)
    return table


# ---- FASTA -> uint8 array contract (reference kmer_count.py:182-218,308-347) ---------------------
def _iter_fasta_records(file_name):
    opener = gzip.open if str(file_name).endswith(".gz") else open
    with opener(file_name, "rt") as fh:
        header, chunks = None, []
        for line in fh:
            if line.startswith(">"):
                if header is not None:
                    yield header, "".join(chunks)
                header, chunks = line[1:].strip(), []
            elif header is not None:
                chunks.append("".join(line.split()))
        if header is not None:
            yield header, "".join(chunks)


def read_dnaseq_file(file_name, file_type="fasta"):
    """Yield one uint8 array (with trailing 255 separator) per FASTA record, sequence upper-cased."""
    assert file_type == "fasta"
    for _, seq in _iter_fasta_records(file_name):
        yield dna2arr(seq.upper(), append_missing_val_flag=True)


def encode_fasta(fasta_file):
    """Whole file (plain or .gz) -> (uint8 array with 255 separators, (n_seq,2) int [start, end-of-read)) by the native
    streaming parser of libkmap_hip (kmap_fasta_*); same output as concatenating read_dnaseq_file()."""
    h, nb, ns = _ffi.vp(), _ffi.i64(0), _ffi.i64(0)
    check(_ffi.lib().kmap_fasta_open(str(fasta_file).encode(), C.byref(h), C.byref(nb), C.byref(ns)))
    try:
        arr = np.empty(nb.value, dtype=np.uint8)
        borders = np.empty((ns.value, 2), dtype=np.int64)
        check(_ffi.lib().kmap_fasta_read(h.value, ptr(arr), ptr(borders)))
    finally:
        _ffi.lib().kmap_fasta_close(h.value)
    return arr, borders.astype(int)


def encode_fasta_py(fasta_file):
    """Pure-Python equivalent of encode_fasta (record by record, like the reference); used to cross-check the parser."""
    parts = list(read_dnaseq_file(fasta_file))
    lens = np.array([len(p) for p in parts], dtype=np.int64)
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64) if len(parts) else np.zeros(0, np.int64)
    arr = np.concatenate(parts) if parts else np.zeros(0, np.uint8)
    return arr, np.stack([starts, starts + lens - 1], axis=1).astype(int) if len(parts) else np.zeros((0, 2), int)


class _BlobAt(Exception):
    def __init__(self, offset, nbytes):
        self.offset, self.nbytes = offset, nbytes


class _OpReader:
    """file wrapper for pickletools.genops that stops at the first large bytes payload instead of reading it"""

    def __init__(self, fh, big):
        self.fh, self.big = fh, big

    def read(self, n):
        if n >= self.big:
            raise _BlobAt(self.fh.tell(), n)
        return self.fh.read(n)

    def readline(self):
        return self.fh.readline()

    def tell(self):
        return self.fh.tell()


_SKIP_OPS = ("PROTO", "FRAME", "MEMOIZE", "BINPUT", "LONG_BINPUT")
_INT_OPS = ("BININT", "BININT1", "BININT2", "LONG1")


def locate_pickled_array(path, min_bytes=1 << 20):
    """Where the data of a pickled ndarray sits inside its file: (offset, dtype, shape), or None when the file is anything but
    ONE C-ordered, little-endian / byte-sized, non-object ndarray pickled with protocol 3 or 4 (what `pickle.dump(arr, fh)` of the
    reference's preproc writes, kmer_count.py:333) whose >= min_bytes payload is one in-band bytes object.  The op stream around
    the payload is matched strictly; any deviation returns None and the caller unpickles the ordinary way."""
    import os
    import pickletools
    ops = []
    with open(path, "rb") as fh:
        try:
            for op, arg, _ in pickletools.genops(_OpReader(fh, max(int(min_bytes), 1024))):   # (shorter reads are opcodes and small arguments)
                if op.name not in _SKIP_OPS:
                    ops.append((op.name, arg))
            return None                                           # no large payload at all
        except _BlobAt as b:
            offset, nbytes = b.offset, b.nbytes
        except Exception:                                         # noqa: BLE001 -- not a pickle we understand
            return None
        size = os.fstat(fh.fileno()).st_size
        if offset + nbytes > size or size - (offset + nbytes) > 16:
            return None
        fh.seek(offset + nbytes)
        try:
            tail = [op.name for op, _, _ in pickletools.genops(fh.read()) if op.name not in _SKIP_OPS]
        except Exception:                                         # noqa: BLE001
            return None
    if tail != ["TUPLE", "BUILD", "STOP"]:
        return None
    names = [n for n, _ in ops]
    # ... MARK 1 <shape tuple> <dtype global> str NEWFALSE NEWTRUE TUPLE3 REDUCE MARK 3 order NONE NONE NONE -1 -1 0 TUPLE BUILD NEWFALSE | payload
    want_tail = ["TUPLE3", "REDUCE", "MARK", "INT", "STR", "NONE", "NONE", "NONE", "INT", "INT", "INT", "TUPLE", "BUILD", "NEWFALSE"]
    kinds = ["INT" if n in _INT_OPS else "STR" if n in ("SHORT_BINUNICODE", "BINUNICODE") else n for n in names]
    if len(kinds) < len(want_tail) + 12 or kinds[-len(want_tail):] != want_tail:
        return None
    state = [a for (n, a), kd in zip(ops[-len(want_tail):], want_tail) if kd in ("INT", "STR")]
    if state[0] != 3 or state[1] not in ("|", "<") or state[2:] != [-1, -1, 0]:
        return None
    i = len(ops) - len(want_tail)
    if kinds[i - 3:i] != ["STR", "NEWFALSE", "NEWTRUE"]:
        return None
    dtype_str = ops[i - 3][1]
    i -= 3
    if kinds[i - 1] == "GLOBAL" and ops[i - 1][1] == "numpy dtype":
        i -= 1
    elif kinds[i - 1] == "STACK_GLOBAL" and kinds[i - 2] == "STR" and ops[i - 2][1] == "dtype":
        i -= 3                                                    # module name (string or memo reference), "dtype", STACK_GLOBAL
    else:
        return None
    shape = []
    if kinds[i - 1] in ("TUPLE1", "TUPLE2", "TUPLE3"):
        nd = int(kinds[i - 1][-1])
        dims = ops[i - 1 - nd:i - 1]
        i -= 1 + nd
    else:
        return None                                               # 0-d and > 3-d arrays: not the input files
    if any(n not in _INT_OPS for n, _ in dims):
        return None
    shape = tuple(int(a) for _, a in dims)
    if kinds[i - 2:i] != ["MARK", "INT"] or ops[i - 1][1] != 1:
        return None
    head = " ".join(str(a) for _, a in ops[:i - 2] if isinstance(a, str))
    if "_reconstruct" not in head or "ndarray" not in head or "multiarray" not in head:
        return None
    try:
        dt = np.dtype(dtype_str)
    except TypeError:
        return None
    if dt.hasobject or dt.itemsize * int(np.prod(shape, dtype=np.int64)) != nbytes:
        return None
    return offset, dt, shape


MAP_PICKLE_MIN_BYTES = 64 << 20    # pickled arrays from this size on are memory-mapped instead of unpickled (load_array_pickle)


def load_array_pickle(path, min_bytes=None, populate=True):
    """The ndarray a pickle file holds.  Large plain arrays (locate_pickled_array) come back as a READ-ONLY view of a private
    memory map of the file -- no 1.5-GB copy through `pickle.load` at C3 (0.27 s + 0.11 s to free it); the upload then DMAs
    straight from the page cache.  Everything else is unpickled normally.
    populate: map the pages in one go (MAP_POPULATE) instead of one soft fault per 4-KiB page while the upload walks the
    array -- 370 000 faults at C3, most of the 0.24 s the first upload of a fresh process took; a rank of a sharded run that
    reads only its slice passes False."""
    import mmap
    import os
    if min_bytes is None:
        min_bytes = MAP_PICKLE_MIN_BYTES
    loc = None
    try:
        if os.path.getsize(path) >= min_bytes:
            loc = locate_pickled_array(path, min_bytes)
    except OSError:
        loc = None
    if loc is None:
        with open(path, "rb") as fh:
            return pickle.load(fh)
    offset, dt, shape = loc
    if offset % dt.itemsize:                                      # an unaligned view would be legal numpy but a trap for native readers
        with open(path, "rb") as fh:
            return pickle.load(fh)
    with open(path, "rb") as fh:
        flags = mmap.MAP_PRIVATE | (getattr(mmap, "MAP_POPULATE", 0) if populate else 0)
        mm = mmap.mmap(fh.fileno(), 0, flags=flags, prot=mmap.PROT_READ)
    return np.frombuffer(mm, dt, count=int(np.prod(shape, dtype=np.int64)), offset=offset).reshape(shape)


class _ArrayPayload:
    """stands for the data bytes of an ndarray inside the no-copy pickler"""

    def __init__(self, view):
        self.view = view


def _plain_big_array(obj):
    return (type(obj) is np.ndarray and obj.flags.c_contiguous and obj.ndim in (1, 2) and obj.dtype.kind in "ui" and obj.dtype.isnative
            and obj.nbytes >= (1 << 20))


class _NoCopyPickler(pickle._Pickler):
    """The Python pickler writing what `pickle.dump(obj, fh)` writes -- same opcodes, same framing (a large bytes object ends the
    current frame and goes out as header + payload) -- except that the payload of every large plain ndarray inside `obj` is the
    array's own memory instead of the `arr.tobytes()` copy numpy's __reduce__ makes for protocols below 5."""

    def __init__(self, fh, protocol):
        super().__init__(fh, protocol)
        self.dispatch = dict(pickle._Pickler.dispatch)
        self.dispatch[_ArrayPayload] = _NoCopyPickler._save_payload

    def reducer_override(self, obj):                                          # asked for every object that is not in the memo yet
        if not _plain_big_array(obj):
            return NotImplemented
        func, args, state = np.empty(0, obj.dtype).__reduce__()              # numpy's own constructor call and state layout
        return func, args, (state[0], obj.shape, state[2], False, _ArrayPayload(memoryview(obj).cast("B")))

    def _save_payload(self, obj):                                             # pickle._Pickler.save_bytes, payload by reference
        import struct
        n = obj.view.nbytes
        if n > 0xFFFFFFFF:
            self._write_large_bytes(pickle.BINBYTES8 + struct.pack("<Q", n), obj.view)
        else:
            self._write_large_bytes(pickle.BINBYTES + struct.pack("<I", n), obj.view)
        self.memoize(obj)


def dump_pickle_nocopy(obj, fh, protocol=None):
    """`pickle.dump(obj, fh)` (reference kmer_count.py:333,341, motif_discovery.py:331-345) without a second copy of the large arrays
    in `obj`: under protocol 4 numpy's __reduce__ hands the pickler `arr.tobytes()` -- 1.5 GB more memory and 0.3 s for C3's
    input.bin.pkl, 15 GB at C5's size.  Byte for byte the same file (test_host_logic.py compares); objects without such an array,
    and interpreters whose default protocol is not 4, take pickle.dump itself."""
    protocol = pickle.DEFAULT_PROTOCOL if protocol is None else protocol
    seq = obj if isinstance(obj, (list, tuple)) else (obj,)
    if protocol == 4 and any(_plain_big_array(x) for x in seq):
        _NoCopyPickler(fh, 4).dump(obj)
    else:
        pickle.dump(obj, fh, protocol=protocol)


dump_array_pickle = dump_pickle_nocopy


def proc_input(input_fasta_file: str, res_dir=".", out_bin_file_name: str = "input.bin.pkl",
               out_boarder_bin_file_name: str = "input.seqboarder.bin.pkl", debug=True):
    assert Path(input_fasta_file).exists()
    assert Path(res_dir).exists()
    assert out_bin_file_name.endswith(".pkl")
    arr, borders = encode_fasta(input_fasta_file)
    out = Path(res_dir) / out_bin_file_name
    if debug:
        print(f"Convert input file={input_fasta_file} into binary file {out}. buffer_size={len(arr) / 2 ** 30}GB.")
    with open(out, "wb") as fh:
        dump_array_pickle(arr, fh)
    with open(Path(res_dir) / out_boarder_bin_file_name, "wb") as fh:
        dump_array_pickle(borders, fh)
    print(f"input binary file {out} generated.\n")


def _preproc(fasta_file: str, res_dir=".", debug=False):
    """`kmap preproc` (reference kmer_count.py:139-179): config.toml, motif_def_table.csv, input pickles."""
    from ._toml import dump_toml, load_toml
    assert Path(fasta_file).exists()
    Path(res_dir).mkdir(exist_ok=True)
    cfg_path = Path(res_dir) / FileNameDict["config_file"]
    had_cfg = cfg_path.exists()
    cfg = load_toml(cfg_path) if had_cfg else read_default_config_file(debug=debug)
    if not had_cfg or cfg["general"].get("input_fasta_file") is None:
        cfg["general"]["input_fasta_file"] = fasta_file
        cfg["general"]["res_dir"] = res_dir
        dump_toml(cfg, cfg_path)
    # the encoder and the two pickle writers are native / system calls that release the interpreter: they run on a helper thread
    # beside the motif table's pandas / scipy imports (~0.8 s in a fresh process, most of what the verb took after the encoder's rebuild)
    import threading
    failed = []

    def _encode():
        try:
            proc_input(cfg["general"]["input_fasta_file"], cfg["general"]["res_dir"],
                       out_bin_file_name=FileNameDict["processed_fasta_file"],
                       out_boarder_bin_file_name=FileNameDict["processed_fasta_seqboarder_file"], debug=debug)
        except BaseException as e:   # noqa: BLE001 -- re-raised below
            failed.append(e)
    enc = threading.Thread(target=_encode)
    enc.start()
    try:
        table = gen_motif_def_dict(cfg, debug=debug)
        ks = sorted(k for k in table if isinstance(k, int))
        with open(Path(res_dir) / FileNameDict["motif_def_file"], "w+") as fh:
            fh.write(MotifDef.get_field_names() + "\n")
            for k in ks:
                fh.write(str(table[k]) + "\n")
    finally:
        enc.join()
    if failed:
        raise failed[0]
    return cfg, table
