"""Row-sharded multi-GPU execution of the Hamming / smoothing / embedding stages (one process per GPU,
torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).

Sharding: every rank holds all N hashes (<= 1.6 MB) and owns row blocks of the N x N problem.  The
Hamming matrix and the neighbour sums need no data-path collective; the embedding loop exchanges, per iteration, the
2 x N gradient and one float64 loss partial (two all-reduces).  Two layouts:
  * SEQ, and FAST below N = 16384: one contiguous row block per rank, every rank evaluates all columns of its rows and
    fills only its rows of the gradient (sum = concatenation, exact);
  * FAST from N = 16384: the symmetric kernel -- each unordered pair once -- with the 256-row blocks dealt out cyclically
    (rank r owns blocks r, r + world, ...: the upper-triangle work of a block shrinks with its index); a rank's gradient
    buffer then holds partial sums for all points and the all-reduce is a true sum.
Best-list / early-stop / jitter logic runs redundantly and identically on every rank (apply kernel), so no further
broadcast is needed.
"""
import os

import numpy as np


def row_partition(n, world, rank):
    """Contiguous balanced row blocks: the first n % world ranks get one extra row."""
    base, extra = divmod(int(n), int(world))
    row0 = rank * base + min(rank, extra)
    return row0, base + (1 if rank < extra else 0)


class DistEmbedLoop:
    """Drives a (row-sharded) embedding session: forces -> all-reduce(grad, loss) -> apply.

    `session` needs forces(grad_ptr, loss_ptr), apply(grad_ptr, loss_ptr); grad_t / loss_t are torch tensors
    (2 x N float32, 1 x float64) on the session's device.  forces() writes only the rows this rank owns, and
    the in-place all-reduce leaves the other ranks' rows in the buffer, so the buffer is zeroed before every
    forces(): the SUM all-reduce is then a concatenation (x + 0 + ... + 0, exact)."""

    def __init__(self, session, grad_t, loss_t, dist=None, group=None):
        self.s, self.g, self.l, self.dist, self.group = session, grad_t, loss_t, dist, group
        self.gp, self.lp = grad_t.data_ptr(), loss_t.data_ptr()
        self.n_collectives = 0

    def step(self, n_iter):
        d = self.dist
        sharded = d is not None and d.get_world_size(self.group) > 1
        for _ in range(n_iter):
            if sharded:
                self.g.zero_()
            self.s.forces(self.gp, self.lp)
            if sharded:
                # rank-ordered ring sum; identical bits on every rank.  x + 0 + ... + 0 is exact for the gradient.
                d.all_reduce(self.g, op=d.ReduceOp.SUM, group=self.group)
                d.all_reduce(self.l, op=d.ReduceOp.SUM, group=self.group)
                self.n_collectives += 2
            self.s.apply(self.gp, self.lp)


def kmap_from_kmers_distributed(samp_kh, samp_cnts, samp_label, conseq_list, kmer_len, n_neighbour=20, n_max_iter=2500,
                                learning_rate=0.01, n_best_result=10, random_seed=None, mode=None, trace=None):
    """Multi-GPU version of visualization.kmap_from_kmers.  Call from every rank after
    torch.distributed.init_process_group("nccl") and torch.cuda.set_device(local_rank)."""
    import torch
    import torch.distributed as dist
    from . import _ffi
    from ._ffi import check, ptr
    from .hamdist import hamdist_matrix_dev, pitch_for
    from .kmer_count import get_hash_dtype
    from . import visualization as vz

    rank, world = dist.get_rank(), dist.get_world_size()
    kh = np.repeat(np.asarray(samp_kh), samp_cnts).astype(get_hash_dtype(kmer_len))
    lab = np.repeat(np.asarray(samp_label), samp_cnts).astype(np.int32)
    n = len(kh)
    mode = vz.default_mode(n) if mode is None else mode
    row0, nrows = row_partition(n, world, rank)
    lens = [len(c) for c in conseq_list]
    ldd = pitch_for(n)
    kh_d, lab_d = _ffi.DeviceBuffer.from_numpy(kh), _ffi.DeviceBuffer.from_numpy(lab)
    # full D on every GPU (N^2 bytes; 40 GB at N = 200 k fits the 288 GB part): neighbour rows are arbitrary
    D_d = _ffi.DeviceBuffer(n * ldd)
    hamdist_matrix_dev(kh_d.ptr, lab_d.ptr, n, kmer_len, lens, D_d.ptr, ldd)
    if vz.knn_mode(n) == "device":
        # every rank selects all rows on its own GPU (N x 20 ints; cheaper than gathering them)
        nb = vz.knn_select_dev(D_d.ptr, ldd, n, n_neighbour)
    else:
        # drop-in neighbour choice for the local rows on the host (numpy argpartition), then all-gather
        rows = np.empty((max(nrows, 1), n), np.uint8)
        if nrows:
            check(_ffi.lib().kmap_memcpy2d_d2h(ptr(rows), n, D_d.ptr + row0 * ldd, ldd, n, nrows, None))
        nb_local = np.argpartition(rows[:nrows].astype(np.int64), n_neighbour, axis=1)[:, :n_neighbour].astype(np.int32)
        parts = [None] * world
        dist.all_gather_object(parts, nb_local)
        nb = np.concatenate(parts)
    # FAST at N >= 16384: symmetric kernel, each unordered pair once; rank r owns the 256-row blocks r, r + world, ...
    # (cyclic), its gradient buffer holds partial sums for ALL points and the all-reduce adds the ranks' buffers.
    # Otherwise: contiguous row blocks, every rank evaluates all columns of its rows (SEQ keeps the reference's row order).
    cyclic = (mode == vz.EMBED_FAST and n >= 16384 and world > 1 and os.environ.get("KMAP_DIST_CYCLIC", "1") != "0")
    if cyclic:
        blocks = vz.cyclic_blocks(n, world, rank)
        lds = (n + 127) & ~127
        blk_bytes = vz.CYCLIC_BLOCK_ROWS * lds * 2
        sums_d = _ffi.DeviceBuffer(max(len(blocks), 1) * blk_bytes)
        for b, (r0, nr) in enumerate(blocks):
            dst = sums_d.ptr + b * blk_bytes
            if vz.knn_sums_kmers_dev(kh_d.ptr, lab_d.ptr, n, kmer_len, lens, nb, n_neighbour, row0=r0, nrows=nr, out=dst) is None:
                vz.knn_sums_dev(D_d.ptr, ldd, nb, n, n_neighbour, row0=r0, nrows=nr, out=dst)
    else:
        res = vz.knn_sums_kmers_dev(kh_d.ptr, lab_d.ptr, n, kmer_len, lens, nb, n_neighbour, row0=row0, nrows=nrows)
        sums_d, lds = res if res is not None else vz.knn_sums_dev(D_d.ptr, ldd, nb, n, n_neighbour, row0=row0, nrows=nrows)
    if isinstance(nb, _ffi.DeviceBuffer):
        nb.free()
    for b in (D_d, kh_d, lab_d):
        b.free()
    lut = vz.hd_prob_lut(kmer_len, n_neighbour, n_neighbour * n_neighbour * kmer_len)
    ld_data, placeholders = vz._init_draws(n, n_best_result, random_seed)      # same seed -> same draws on every rank
    if cyclic:
        sess = vz.EmbedSession(n, n_best_result, learning_rate, vz.EMBED_FAST, cyclic=(world, rank))
    else:
        sess = vz.EmbedSession(n, n_best_result, learning_rate, mode, row0=row0, nrows=nrows)
    try:
        sess.set_prob_lut(sums_d, lds, lut)
        sess.set_coords(ld_data, placeholders)
        grad_t = torch.zeros((2, n), dtype=torch.float32, device="cuda")
        loss_t = torch.zeros(1, dtype=torch.float64, device="cuda")
        loop = DistEmbedLoop(sess, grad_t, loss_t, dist)
        vz._run_loop(sess, n_max_iter, step_fn=loop.step, trace=trace)
        return sess.best(), lab
    finally:
        sess.close()


# ---- read-sharded k-mer counting / masking / scanning -----------------------------------------------------------
class _DevArray:
    """Zero-copy view of library-owned device memory for torch (via __cuda_array_interface__)."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def read_partition(borders, world, rank):
    """Contiguous read ranges cut at read borders, balanced by read count: returns (first_read, n_reads)."""
    return row_partition(len(borders), world, rank)


def make_dist_device_seq(seq_np_arr, boarder_mat, dist, group=None):
    """A DeviceSeq holding only this rank's reads whose count()/scan() results are global:
    count = local histogram -> all-reduce(SUM) of the 4^k uint32 bins -> identical compaction on every rank
    (k <= 16; per-read dedupe, masking and scanning are local to a read, hence to a rank).  Because every rank then
    sees the same counts, find_motif(dev_seq=...) makes the same decisions everywhere without further exchange."""
    import ctypes as C
    import torch
    from . import _ffi
    from ._ffi import check
    from .motif_discovery import DeviceSeq

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    borders = np.ascontiguousarray(boarder_mat, dtype=np.int64).reshape(-1, 2)
    r0, nr = read_partition(borders, world, rank)
    if nr:
        lo, hi = int(borders[r0, 0]), int(borders[r0 + nr - 1, 1]) + 1          # include the last read's separator
        hi = min(hi, len(seq_np_arr))
    else:
        lo = hi = 0
    local_seq = np.ascontiguousarray(seq_np_arr[lo:hi])
    local_borders = borders[r0:r0 + nr] - lo

    class DistDeviceSeq(DeviceSeq):
        first_read, n_local_reads, n_all_reads = r0, nr, len(borders)

        def count(self, dc, k, dedupe, merge_revcom, use_work=True):
            if k > 16:
                raise ValueError("sharded counting all-reduces the 4^k histogram and needs k <= 16")
            inval = self.inval_work if use_work else self.inval_orig
            check(_ffi.lib().kmap_counts_hist_packed_dev(dc._h, self.codes.ptr, inval.ptr, self.n, self.borders.ptr,
                                                         self.n_seq, k, int(dedupe), None))
            p, nb = _ffi.vp(), _ffi.i64(0)
            check(_ffi.lib().kmap_counts_bins(dc._h, C.byref(p), C.byref(nb)))
            bins = torch.as_tensor(_DevArray(p.value, 4 ** k, "<i4"), device="cuda")   # int32 sum wraps like uint32
            torch.cuda.synchronize()
            dist.all_reduce(bins, op=dist.ReduceOp.SUM, group=group)
            torch.cuda.synchronize()
            nu = _ffi.i64(0)
            check(_ffi.lib().kmap_counts_finish(dc._h, k, int(merge_revcom), C.byref(nu), None))
            dc.k, dc.n_uniq = k, nu.value
            return dc.n_uniq

        def scan(self, k, consensus_kh, radius, revcom):
            hits, pos = DeviceSeq.scan(self, k, consensus_kh, radius, revcom)
            parts = [None] * world
            dist.all_gather_object(parts, (hits, pos), group=group)                    # read order = rank order
            return np.concatenate([h for h, _ in parts]), np.concatenate([q for _, q in parts])

    ds = DistDeviceSeq(local_seq, local_borders)
    all_len = (borders[:, 1] - borders[:, 0]).astype(np.int64)
    ds.read_len_global = all_len
    return ds
