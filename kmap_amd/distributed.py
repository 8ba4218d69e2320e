"""Row-sharded multi-GPU execution of the Hamming / smoothing / embedding stages (one process per GPU,
torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).

Sharding: every rank holds all N hashes (<= 1.6 MB) and owns a contiguous block of rows of
the N x N problem.  The Hamming matrix and the neighbour sums need no data-path collective; the embedding
loop exchanges, per iteration, the 2 x N gradient (each rank fills only its rows; sum = concatenation, exact)
and one float64 loss partial.  Best-list / early-stop / jitter logic then runs redundantly and identically on
every rank (apply kernel), so no further broadcast is needed.
"""
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
    mode = vz.default_mode() if mode is None else mode
    kh = np.repeat(np.asarray(samp_kh), samp_cnts).astype(get_hash_dtype(kmer_len))
    lab = np.repeat(np.asarray(samp_label), samp_cnts).astype(np.int32)
    n = len(kh)
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
    sums_d, lds = vz.knn_sums_dev(D_d.ptr, ldd, nb, n, n_neighbour, row0=row0, nrows=nrows)
    if isinstance(nb, _ffi.DeviceBuffer):
        nb.free()
    for b in (D_d, kh_d, lab_d):
        b.free()
    lut = vz.hd_prob_lut(kmer_len, n_neighbour, n_neighbour * n_neighbour * kmer_len)
    ld_data, placeholders = vz._init_draws(n, n_best_result, random_seed)      # same seed -> same draws on every rank
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
