"""All-pairs Hamming matrix of sampled k-mers on the GPU (reference cal_samp_kmer_hamdist_mat,
motif_discovery.py:759-808, plus the block expansion :705-757)."""
import numpy as np

from . import _ffi
from ._ffi import check, ptr
from .kmer_count import get_hash_dtype


def pitch_for(n):
    """Row pitch (bytes) of the device-resident uint8 matrix.  From one 4-KiB column block up: a multiple of 4 KiB with an
    ODD number of 4-KiB chunks per row, which is what the tiled Hamming kernel needs to give every XCD its own residue
    class of chunks (6.5+ TB/s of stores instead of 5.3-5.8; csrc/hamdist_matrix.hip).  Below that: 256-byte aligned rows
    (every 1-KiB wave store covers whole 128-byte lines: 5.3 TB/s vs 3.3 TB/s with a 16-byte pitch)."""
    return int(_ffi.lib().kmap_hamdist_pitch(int(n)))


def hamdist_matrix_dev(kh_dev_ptr, label_dev_ptr, n, k, conseq_lens, out_dev_ptr, ld, row0=0, nrows=None, stream=None):
    """Launch the matrix kernel on device-resident inputs; writes rows [row0,row0+nrows) to out_dev_ptr."""
    nrows = n - row0 if nrows is None else nrows
    clen = np.ascontiguousarray(conseq_lens, dtype=np.int32)
    fn = _ffi.lib().kmap_hamdist_matrix_u32_dev if get_hash_dtype(k) == np.uint32 else _ffi.lib().kmap_hamdist_matrix_u64_dev
    check(fn(kh_dev_ptr, label_dev_ptr, n, k, ptr(clen) if len(clen) else None, len(clen), row0, nrows, out_dev_ptr, ld,
             stream))


def hamdist_matrix_u8(kh, label, k, conseq_lens):
    """Host convenience: N hashes + N labels -> dense N x N uint8 Hamming matrix."""
    kh = np.ascontiguousarray(kh, dtype=np.uint64)
    label = np.ascontiguousarray(label, dtype=np.int32)
    clen = np.ascontiguousarray(conseq_lens, dtype=np.int32)
    n = len(kh)
    out = np.empty((n, n), dtype=np.uint8)
    check(_ffi.lib().kmap_hamdist_matrix_u8(ptr(kh), ptr(label), n, k, ptr(clen) if len(clen) else None, len(clen), ptr(out)))
    return out


def _convert_to_block_arr(arr, block_size_arr):
    """Repeat arr[i] block_size_arr[i] times (reference motif_discovery.py:733-757)."""
    block_size_arr = np.asarray(block_size_arr)
    assert np.issubdtype(block_size_arr.dtype, np.integer)
    assert np.all(block_size_arr > 0)
    assert len(arr) == len(block_size_arr)
    return np.repeat(np.asarray(arr), block_size_arr)


def cal_samp_kmer_hamdist_mat(samp_kh_arr, samp_cnts, samp_label_arr, conseq_list, kmer_len, uniq_dist_flag=False):
    """Drop-in for the reference operator: int64 matrix, unique (uniq_dist_flag) or expanded by counts."""
    assert len(samp_kh_arr) == len(np.unique(samp_kh_arr))
    for conseq in conseq_list:
        assert len(conseq) <= kmer_len
    lens = [len(c) for c in conseq_list]
    if uniq_dist_flag:
        return hamdist_matrix_u8(samp_kh_arr, samp_label_arr, kmer_len, lens).astype(int)
    kh = _convert_to_block_arr(samp_kh_arr, samp_cnts)
    lab = _convert_to_block_arr(samp_label_arr, samp_cnts)
    return hamdist_matrix_u8(kh, lab, kmer_len, lens).astype(int)
