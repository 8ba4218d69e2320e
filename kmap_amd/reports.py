"""Consumers of the occurrence hit list and of the counted k-mers.

Same names, arguments and return values as the reference's reporting helpers (motif_discovery.py):
`get_motif_seq_num` (:1345), `get_motif_pos_density` (:1255), `get_motif_co_occurence_mat` (:1189),
`write_co_occurence_mat` (:1165), `write_co_occurence_dist_arr` (:1143), `ex_hamball_kh_arr` (:924), `cal_cnt_mat` (:978),
`_ex_hamball` (:489).  The reference re-parses `*.motif_occurence.csv` row by row (one scipy `norm(...).pdf` per hit, one
`np.median` per cell); here every consumer also accepts the in-memory hit list the occurrence scan produced (`Occurrence`),
so `scan_motif` never re-reads its own CSV.  The density and the Hamming-ball extraction run on the GPU
(csrc/reports.hip); the co-occurrence statistics are exact integer / half-integer numpy reductions over the hit list.
Plotting (pdf figures, logos, networks) is outside the hot path and not provided.
"""
import csv
import ctypes as C
import pickle
from pathlib import Path
from typing import List

import numpy as np

from . import _ffi
from ._ffi import check, ptr
from .kmer_count import (FileNameDict, get_cnt_dtype, get_hash_dtype, hash2kmer, init_motif_def_dict, kmer2hash,
                         reverse_complement, revcom_hash)


class Occurrence:
    """The hit list behind one `*.motif_occurence.csv`: per consensus c `hits[c]` (int32 per row) and `pos[c]` (int32,
    concatenated in row order, ascending inside a cell), `seq_len` (int64 per row), `seq_ind` (int64 per row).
    Rows may include reads without any hit (they are ignored exactly like the rows the CSV omits)."""

    def __init__(self, hits: List[np.ndarray], pos: List[np.ndarray], seq_len, seq_ind=None):
        self.hits = [np.ascontiguousarray(h, np.int32) for h in hits]
        self.pos = [np.ascontiguousarray(p, np.int32) for p in pos]
        self.seq_len = np.ascontiguousarray(seq_len, np.int64)
        n = len(self.seq_len)
        self.seq_ind = np.arange(n, dtype=np.int64) if seq_ind is None else np.ascontiguousarray(seq_ind, np.int64)
        for h, p in zip(self.hits, self.pos):
            assert len(h) == n and int(h.sum(dtype=np.int64)) == len(p)
        self._offs = [None] * len(self.hits)

    @classmethod
    def from_per(cls, per, read_len):
        """from scan_motif_occurence's result (one row per read of the input)"""
        return cls([h for h, _ in per], [p for _, p in per], read_len)

    @classmethod
    def from_file(cls, path, n_conseq=None):
        """parse `seq_ind;loc,loc;...;seq_len` (gen_motif_occurence_file's format, motif_discovery.py:1396-1419)"""
        with open(path, "r", newline="") as fh:
            reader = csv.reader(fh, delimiter=";")
            header = next(reader)
            n_c = len(header) - 2
            if n_conseq is not None:
                assert len(header) == n_conseq + 2
            hits = [[] for _ in range(n_c)]
            pos = [[] for _ in range(n_c)]
            seq_len, seq_ind = [], []
            for row in reader:
                seq_ind.append(int(row[0]))
                seq_len.append(int(float(row[-1].strip())))
                for c in range(n_c):
                    cell = row[c + 1].strip()
                    locs = sorted(int(v) for v in cell.split(",")) if cell != "" else []
                    hits[c].append(len(locs))
                    pos[c].extend(locs)
        return cls([np.array(h, np.int32) for h in hits], [np.array(p, np.int32) for p in pos],
                   np.array(seq_len, np.int64), np.array(seq_ind, np.int64))

    @property
    def n_conseq(self):
        return len(self.hits)

    def offs(self, c):
        if self._offs[c] is None:
            o = np.zeros(len(self.seq_len) + 1, np.int64)
            np.cumsum(self.hits[c], dtype=np.int64, out=o[1:])
            self._offs[c] = o
        return self._offs[c]

    def medians(self, c):
        """np.median of every cell (NaN for an empty cell): the mean of the two middle locations of the sorted cell"""
        h, p = self.hits[c], self.pos[c]
        med = np.empty(len(h), np.float64)
        check(_ffi.lib().kmap_cell_medians_i32(ptr(h), ptr(p), len(h), len(p), ptr(med)))   # one native pass (host code)
        return med


def _as_occurrence(occ, n_conseq=None) -> Occurrence:
    if isinstance(occ, Occurrence):
        return occ
    if isinstance(occ, (str, Path)):
        return Occurrence.from_file(occ, n_conseq)
    return Occurrence([h for h, _ in occ], [p for _, p in occ], np.zeros(len(occ[0][0]), np.int64))   # bare `per` list


def get_motif_seq_num(occurence_file_path, motif_index: int):
    """(rows with the motif, total occurrences) -- reference motif_discovery.py:1345-1393"""
    if isinstance(occurence_file_path, list):          # bare hit list from scan_motif_occurence: no container needed
        entry = occurence_file_path[motif_index]
        if hasattr(entry, "n_reads_hit"):               # ScanHits: summary from the device, the list itself is not fetched
            return int(entry.n_reads_hit), int(entry.total)
        hits, pos = entry
        return int(np.count_nonzero(hits)), len(pos)    # the positions array holds exactly sum(hits) entries
    hits = _as_occurrence(occurence_file_path).hits[motif_index]
    return int(np.count_nonzero(hits)), int(hits.sum(dtype=np.int64))


def get_motif_pos_density(occurence_file_path, motif_index: int, kmer_len: int, x_step=0.01, x_arr=None):
    """(rows with the motif, occurrences, density over x_arr) -- reference motif_discovery.py:1255-1327.  GPU kernel; f64,
    agrees with the reference's sequential scipy sum to ~1e-13 relative (summation order over reads differs)."""
    occ = _as_occurrence(occurence_file_path)
    if x_arr is None:
        x_arr = np.arange(0, 1, x_step)
    x = np.ascontiguousarray(x_arr, np.float64)
    hits, pos, offs = occ.hits[motif_index], occ.pos[motif_index], occ.offs(motif_index)
    density = np.zeros(len(x), np.float64)
    pos_arg = pos if len(pos) else np.zeros(1, np.int32)
    check(_ffi.lib().kmap_pos_density(ptr(hits), ptr(offs), ptr(pos_arg), ptr(occ.seq_len), len(hits), int(kmer_len), ptr(x),
                                      len(x), float(x_step), ptr(density)))
    out = np.zeros_like(x_arr)                   # the reference accumulates into zeros_like(x_arr)
    out[...] = density
    return int(np.count_nonzero(hits)), int(hits.sum(dtype=np.int64)), out


def get_motif_co_occurence_mat(occurence_file_path, n_conseq: int, as_arrays=False):
    """(co-occurrence counts with per-motif row counts on the diagonal, median |distance| matrix, {(i,j): signed
    distances median_j - median_i in row order}) -- reference motif_discovery.py:1189-1253.  as_arrays: the distances stay float64
    arrays instead of the reference's lists (scan_motif's own caller: 3 M values per pair at C3)"""
    assert n_conseq > 0
    occ = _as_occurrence(occurence_file_path, n_conseq)
    assert occ.n_conseq == n_conseq
    present = [h > 0 for h in occ.hits]
    med = [occ.medians(c) for c in range(n_conseq)]
    res_mat = np.zeros((n_conseq, n_conseq), dtype=int)
    dist_mat = np.zeros((n_conseq, n_conseq), dtype=float)
    dist_dict = {}
    for i in range(n_conseq):
        for j in range(i + 1, n_conseq):
            both = present[i] & present[j]
            d = med[j][both] - med[i][both]
            dist_dict[(i, j)] = d if as_arrays else list(d)
            res_mat[i, j] = res_mat[j, i] = int(np.count_nonzero(both))
            dist_mat[i, j] = dist_mat[j, i] = 1e6 if len(d) == 0 else np.median(np.abs(d))
    np.fill_diagonal(res_mat, [int(np.count_nonzero(p)) for p in present])
    return res_mat, dist_mat, dist_dict


def write_co_occurence_dist_arr(output_file, dist_dict, conseq_list: List[str]):
    """reference motif_discovery.py:1143-1162"""
    names = [f"m{i}_{s}_{reverse_complement(s)}" for i, s in enumerate(conseq_list)]
    with open(output_file, "wb") as fh:
        for i, j in dist_dict:
            vals = np.ascontiguousarray(dist_dict[(i, j)], dtype=np.float64)
            if len(vals) == 0:
                continue
            fh.write((names[i] + "-" + names[j] + "\n").encode())
            fh.flush()
            # "\t".join(f"{n:.2f}" for n in vals) + "\n", formatted and written natively (1.2 s of Python per 3 M values)
            check(_ffi.lib().kmap_write_f2_tsv_line(fh.fileno(), ptr(vals), len(vals)))


def write_co_occurence_mat(output_file, dist_mat: np.ndarray, conseq_list: List[str]):
    """reference motif_discovery.py:1165-1186"""
    assert len(conseq_list) == len(dist_mat)
    rc_names = [f"m{i}_{reverse_complement(s)}" for i, s in enumerate(conseq_list)]
    with open(output_file, "w") as fh:
        fh.write("\t".join(["RC"] + [f"m{i}_{s}" for i, s in enumerate(conseq_list)]) + "\n")
        for i, arr in enumerate(dist_mat):
            arr = np.around(arr, decimals=2)
            fh.write(rc_names[i] + "\t" + "\t".join(str(x) for x in arr) + "\n")


# ---- Hamming-ball extraction -----------------------------------------------------------------------------------
def _hamball_extract(uniq_kh_arr, uniq_kh_cnt_arr, kmer_len, conseq_kh, max_ham_dist, revcom_mode, want_mat=True):
    u = np.ascontiguousarray(uniq_kh_arr, get_hash_dtype(kmer_len))
    c = np.ascontiguousarray(uniq_kh_cnt_arr, get_cnt_dtype(kmer_len))
    assert len(u) == len(c)
    out_u, out_c = np.empty(max(len(u), 1), u.dtype), np.empty(max(len(u), 1), c.dtype)
    mat = np.zeros((4, kmer_len), np.int64)
    n_out = _ffi.i64(0)
    check(_ffi.lib().kmap_hamball_extract(ptr(u), ptr(c), len(u), kmer_len, int(conseq_kh), int(max_ham_dist), int(bool(revcom_mode)),
                                          ptr(out_u), ptr(out_c), C.byref(n_out), ptr(mat) if want_mat else None))
    return out_u[:n_out.value], out_c[:n_out.value], mat


def _hamball_extract_resident(dc, kmer_len, conseq_kh, max_ham_dist, revcom_mode):
    """the ball over the table a DeviceCounts handle holds in HBM (two calls: how many members, then the members)"""
    n_out = _ffi.i64(0)
    args = (dc._h, int(conseq_kh), int(max_ham_dist), int(bool(revcom_mode)))
    check(_ffi.lib().kmap_counts_hamball_extract(*args, 0, None, None, C.byref(n_out), None))
    n = n_out.value
    u, c = np.empty(n, get_hash_dtype(kmer_len)), np.empty(n, get_cnt_dtype(kmer_len))
    if n:
        check(_ffi.lib().kmap_counts_hamball_extract(*args, n, ptr(u), ptr(c), C.byref(n_out), None))
        assert n_out.value == n
    return u, c


def ex_hamball_kh_arr(res_dir: str, conseq: str, max_ham_dist: int = -1, motif_def_file: str = None, revcom_mode=True, resident=None):
    """(hashes, counts) of the counted k-mers inside the Hamming ball of `conseq`, reverse-complement members re-oriented
    to the consensus -- reference motif_discovery.py:924-975.  resident: a DeviceCounts handle that still holds the whole table of
    k{len(conseq)}.pkl in HBM (scan_motif's own call): the file is not read back"""
    conseq = conseq.upper()
    assert all(e in ("A", "C", "G", "T") for e in conseq)
    kmer_len = len(conseq)
    conseq_kh = kmer2hash(conseq)
    if revcom_mode:
        assert conseq_kh <= revcom_hash(conseq_kh, kmer_len)
    assert Path(motif_def_file).exists()
    assert Path(res_dir).exists()
    if max_ham_dist == -1:
        max_ham_dist = init_motif_def_dict(motif_def_file)[kmer_len].max_ham_dist
    if resident is not None and resident.k == kmer_len and getattr(resident, "_shard", None) is None:
        return _hamball_extract_resident(resident, kmer_len, conseq_kh, max_ham_dist, revcom_mode)
    with open(Path(res_dir) / FileNameDict["kmer_count_dir"] / f"k{kmer_len}.pkl", "rb") as fh:
        res_list = pickle.load(fh)
    assert res_list[0] == kmer_len
    u, c, _ = _hamball_extract(res_list[1], res_list[2], kmer_len, conseq_kh, max_ham_dist, revcom_mode, want_mat=False)
    return u, c


def cal_cnt_mat(uniq_kh_arr, uniq_kh_cnt_arr, kmer_len):
    """4 x kmer_len base-count matrix weighted by the k-mer counts -- reference motif_discovery.py:978-986"""
    _, _, mat = _hamball_extract(uniq_kh_arr, uniq_kh_cnt_arr, kmer_len, 0, kmer_len, False)   # radius k: every k-mer
    return mat.astype(int)


def _ex_hamball(res_dir: str, conseq: str, return_type: str, output_file: str, max_ham_dist: int = -1, resident=None):
    """`kmap ex_hamball`: write the ball as hash,count / kmer,count lines or as the count matrix -- reference :489-530.
    resident: see ex_hamball_kh_arr"""
    from ._toml import load_toml
    config_file_path = Path(res_dir) / FileNameDict["config_file"]
    assert config_file_path.exists()
    config_dict = load_toml(config_file_path)
    assert return_type in ("hash", "kmer", "matrix")
    motif_def_file_path = Path(res_dir) / FileNameDict["motif_def_file"]
    revcom_mode = config_dict["kmer_count"]["revcom_mode"]
    uniq_kh_arr, uniq_kh_cnt_arr = ex_hamball_kh_arr(res_dir, conseq, max_ham_dist, motif_def_file_path, revcom_mode, resident=resident)
    kmer_len = len(conseq)
    with open(output_file, "w+") as fh:
        if return_type == "hash":
            fh.write("".join(f"{kh},{cnt}\n" for kh, cnt in zip(uniq_kh_arr.tolist(), uniq_kh_cnt_arr.tolist())))
        elif return_type == "kmer":
            fh.write("".join(f"{hash2kmer(kh, kmer_len)},{cnt}\n" for kh, cnt in zip(uniq_kh_arr, uniq_kh_cnt_arr.tolist())))
        else:
            np.savetxt(fh, cal_cnt_mat(uniq_kh_arr, uniq_kh_cnt_arr, kmer_len), delimiter=",", fmt="%d")
    print(f"Extract Hamming ball [type={return_type}] save in {output_file}.")
