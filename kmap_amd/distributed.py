"""Row-sharded multi-GPU execution of the Hamming / smoothing / embedding stages and read-sharded counting / scanning
(one process per GPU, torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).

Sharding: every rank holds all N hashes (<= 1.6 MB) and owns row blocks of the N x N problem:
  * Hamming matrix: a rank computes ONLY its rows of D (N^2 / G bytes of HBM per rank) and picks the 20 neighbours of those
    rows from them; the N x 20 neighbour table is completed by one all-gather of device tensors (RCCL).
  * neighbour sums: from the k-mers' base-count profiles (csrc/knn_profile.hip), which need the neighbour table but no D
    rows, for exactly the rows the rank's embedding session owns.  (k > 16 or > 4 short consensuses: the matrix-based
    fallback needs arbitrary rows of D, so D is then computed in full on every rank.)
  * embedding loop: per iteration the 2 x N gradient and one float64 loss partial are all-reduced.  Two layouts:
      - SEQ, and FAST below N = 16384: one contiguous row block per rank, every rank evaluates all columns of its rows and
        fills only its rows of the gradient (sum = concatenation, exact);
      - FAST from N = 16384: the symmetric kernel -- each unordered pair once -- with the 256-row blocks dealt out cyclically
        (rank r owns blocks r, r + world, ...: the upper-triangle work of a block shrinks with its index); a rank's gradient
        buffer then holds partial sums for all points and the all-reduce is a true sum.
    Best-list / early-stop / jitter logic runs redundantly and identically on every rank (apply kernel): no broadcast.
  * reads (counting, masking, occurrence scan): contiguous read ranges; the 4^k-bin histogram is all-reduced in place,
    scan hits are all-gathered in read order (padded tensors; device tensors on RCCL).
The library launches on the null stream, which is torch's default stream, so collectives are ordered after the kernels that
produce their operands (and before the ones that consume them) on the device: no host synchronisation brackets them.
"""
import os

import numpy as np


def row_partition(n, world, rank):
    """Contiguous balanced row blocks: the first n % world ranks get one extra row."""
    base, extra = divmod(int(n), int(world))
    row0 = rank * base + min(rank, extra)
    return row0, base + (1 if rank < extra else 0)


def _coll_device(dist, group=None):
    """tensors of collectives live on the GPU for RCCL ("nccl") and on the host for gloo (CPU rehearsals, one-GPU tests)"""
    return "cuda" if dist.get_backend(group) == "nccl" else "cpu"


def all_gather_concat(dist, arr, lengths=None, group=None):
    """Concatenation, in rank order, of every rank's array (same dtype and trailing shape, different leading lengths) through
    one all_gather of padded tensors -- device tensors on RCCL, host tensors on gloo; no pickling.  `lengths` (leading length
    per rank) is exchanged first when the caller does not know it."""
    import torch
    world = dist.get_world_size(group)
    dev = _coll_device(dist, group)
    arr = np.ascontiguousarray(arr)
    if lengths is None:
        mine = torch.tensor([arr.shape[0]], dtype=torch.int64, device=dev)
        parts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(parts, mine, group=group)
        lengths = [int(p.item()) for p in parts]
    cap = max(max(lengths), 1)
    pad = np.zeros((cap,) + arr.shape[1:], arr.dtype)
    pad[:arr.shape[0]] = arr
    view = pad.view(np.uint8).reshape(cap, -1)                      # bytes: torch has no uint16 / uint32 collectives
    mine = torch.from_numpy(view).to(dev)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    out = [p.cpu().numpy().view(arr.dtype).reshape((cap,) + arr.shape[1:])[:lengths[r]] for r, p in enumerate(parts)]
    return np.concatenate(out)


class _DevArray:
    """Zero-copy view of library-owned device memory for torch (via __cuda_array_interface__)."""

    def __init__(self, ptr, shape, typestr):
        shape = (int(shape),) if np.isscalar(shape) else tuple(int(s) for s in shape)
        self.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (int(ptr), False), "version": 2}


def all_gather_rows_dev(dist, local_dev, n, world, rank, row_elems, group=None):
    """All ranks' contiguous row blocks (row_partition) of an int32 [n, row_elems] device table -> the whole table on this
    rank's device.  RCCL: all_gather of padded device tensors, the result stays in HBM; gloo: staged through the host.
    Returns a DeviceBuffer-like object (`.ptr`, `.free()`)."""
    import torch
    from . import _ffi
    r0, nr = row_partition(n, world, rank)
    if _coll_device(dist, group) != "cuda":
        rows = local_dev.to_numpy(np.int32, (nr, row_elems)) if nr else np.zeros((0, row_elems), np.int32)
        full = all_gather_concat(dist, rows, [row_partition(n, world, r)[1] for r in range(world)], group)
        return _ffi.DeviceBuffer.from_numpy(full)
    cap = row_partition(n, world, 0)[1]                               # rank 0 holds the longest block
    mine = torch.zeros((cap, row_elems), dtype=torch.int32, device="cuda")
    if nr:
        mine[:nr].copy_(torch.as_tensor(_DevArray(local_dev.ptr, (nr, row_elems), "<i4"), device="cuda"))
    parts = torch.empty((world, cap, row_elems), dtype=torch.int32, device="cuda")
    dist.all_gather_into_tensor(parts, mine, group=group)
    full = torch.cat([parts[r, :row_partition(n, world, r)[1]] for r in range(world)]).contiguous()
    return _ffi.DeviceView(full.data_ptr(), full.numel() * 4, keep=full)


def broadcast_seed(dist, random_seed, group=None):
    """`random_seed = "default"` (None) means OS entropy in the reference (np.random.seed(None)); under a multi-rank launch
    every rank must still start from the same coordinates, placeholders and jitter stream, so rank 0 draws the seed and
    broadcasts it."""
    if random_seed is not None or dist is None or dist.get_world_size(group) <= 1:
        return random_seed
    import torch
    t = torch.zeros(1, dtype=torch.int64, device=_coll_device(dist, group))
    if dist.get_rank(group) == 0:
        t[0] = int(np.random.SeedSequence().entropy) & 0xFFFFFFFF
    dist.broadcast(t, 0, group=group)
    return int(t.item())


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
                                learning_rate=0.01, n_best_result=10, random_seed=None, mode=None, trace=None,
                                neighbor_inds_mat=None):
    """Multi-GPU version of visualization.kmap_from_kmers.  Call from every rank after
    torch.distributed.init_process_group("nccl") and torch.cuda.set_device(local_rank).
    neighbor_inds_mat: optional full (N, n_neighbour) table to use instead of the selection (tests inject it)."""
    import torch
    import torch.distributed as dist
    from . import _ffi
    from ._ffi import check, ptr
    from .hamdist import hamdist_matrix_dev, pitch_for
    from .kmer_count import get_hash_dtype
    from . import visualization as vz

    rank, world = dist.get_rank(), dist.get_world_size()
    random_seed = broadcast_seed(dist, random_seed)
    kh = np.repeat(np.asarray(samp_kh), samp_cnts).astype(get_hash_dtype(kmer_len))
    lab = np.repeat(np.asarray(samp_label), samp_cnts).astype(np.int32)
    n = len(kh)
    mode = vz.default_mode(n) if mode is None else mode
    row0, nrows = row_partition(n, world, rank)
    lens = [len(c) for c in conseq_list]
    ldd = pitch_for(n)
    kh_d, lab_d = _ffi.DeviceBuffer.from_numpy(kh), _ffi.DeviceBuffer.from_numpy(lab)
    hbm = {"d_rows": nrows, "d_bytes": nrows * ldd}
    # this rank's rows of D only (N^2 / G bytes): enough to choose the neighbours of these rows
    D_d = _ffi.DeviceBuffer(max(nrows, 1) * ldd)
    if nrows:
        hamdist_matrix_dev(kh_d.ptr, lab_d.ptr, n, kmer_len, lens, D_d.ptr, ldd, row0=row0, nrows=nrows)
    if neighbor_inds_mat is not None:
        nb = _ffi.DeviceBuffer.from_numpy(np.ascontiguousarray(neighbor_inds_mat, np.int32))
    else:
        if vz.knn_mode(n) == "device":
            nb_local = vz.knn_select_dev(D_d.ptr, ldd, n, n_neighbour, row0=0, nrows=nrows)     # D_d holds local rows from 0
        else:
            # drop-in neighbour choice for the local rows on the host (numpy argpartition on int64 rows)
            rows = np.empty((max(nrows, 1), n), np.uint8)
            if nrows:
                check(_ffi.lib().kmap_memcpy2d_d2h(ptr(rows), n, D_d.ptr, ldd, n, nrows, None))
            sel = np.argpartition(rows[:nrows].astype(np.int64), n_neighbour, axis=1)[:, :n_neighbour].astype(np.int32)
            nb_local = _ffi.DeviceBuffer.from_numpy(sel) if nrows else _ffi.DeviceBuffer(16)
        nb = all_gather_rows_dev(dist, nb_local, n, world, rank, n_neighbour)
        nb_local.free()
    # FAST at N >= 16384: symmetric kernel, each unordered pair once; rank r owns the 256-row blocks r, r + world, ...
    # (cyclic), its gradient buffer holds partial sums for ALL points and the all-reduce adds the ranks' buffers.
    # Otherwise: contiguous row blocks, every rank evaluates all columns of its rows (SEQ keeps the reference's row order).
    cyclic = (mode == vz.EMBED_FAST and n >= 16384 and world > 1 and os.environ.get("KMAP_DIST_CYCLIC", "1") != "0")
    blocks = vz.cyclic_blocks(n, world, rank) if cyclic else [(row0, nrows)]
    lds = (n + 127) & ~127
    blk_bytes = (vz.CYCLIC_BLOCK_ROWS if cyclic else max(nrows, 1)) * lds * 2
    sums_d = _ffi.DeviceBuffer(max(len(blocks), 1) * blk_bytes)
    full_D = None
    for b, (r0, nr) in enumerate(blocks):
        dst = sums_d.ptr + b * blk_bytes
        if nr and vz.knn_sums_kmers_dev(kh_d.ptr, lab_d.ptr, n, kmer_len, lens, nb, n_neighbour, row0=r0, nrows=nr, out=dst) is None:
            if full_D is None:     # profile kernel does not cover this request: the matrix-based sums gather arbitrary rows of D
                D_d.free()
                full_D = _ffi.DeviceBuffer(n * ldd)
                hamdist_matrix_dev(kh_d.ptr, lab_d.ptr, n, kmer_len, lens, full_D.ptr, ldd)
                hbm["d_rows"], hbm["d_bytes"] = n, n * ldd
            vz.knn_sums_dev(full_D.ptr, ldd, nb, n, n_neighbour, row0=r0, nrows=nr, out=dst)
    _ffi.sync()
    nb.free()
    for buf in (D_d, kh_d, lab_d, full_D):
        if buf is not None:
            buf.free()
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
        if trace is not None:
            trace["hbm"] = hbm
            trace["seed"] = random_seed
        return sess.best(), lab
    finally:
        sess.close()


# ---- read-sharded k-mer counting / masking / scanning -----------------------------------------------------------
def read_partition(borders, world, rank):
    """Contiguous read ranges cut at read borders, balanced by read count: returns (first_read, n_reads)."""
    return row_partition(len(borders), world, rank)


def make_dist_device_seq(seq_np_arr, boarder_mat, dist, group=None):
    """A DeviceSeq holding only this rank's reads whose count()/scan() results are global:
    count = local histogram -> all-reduce(SUM) of the 4^k uint32 bins -> identical compaction on every rank
    (k <= 16; per-read dedupe, masking and scanning are local to a read, hence to a rank).  Because every rank then
    sees the same counts, find_motif(dev_seq=...) makes the same decisions everywhere without further exchange.
    scan() returns the hits of ALL reads (all-gathered in read order); `out_n_seq` / `out_read_len` describe the reads those
    results cover (all of them), `n_seq` / `read_len` stay the local shard the kernels run on."""
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
    reads_of = [read_partition(borders, world, r)[1] for r in range(world)]

    class DistDeviceSeq(DeviceSeq):
        first_read, n_local_reads, n_all_reads = r0, nr, len(borders)
        scan_lazy = None        # hit lists are all-gathered through the host: no device-resident variant

        def count(self, dc, k, dedupe, merge_revcom, use_work=True):
            if k > 16:
                raise ValueError("sharded counting all-reduces the 4^k histogram and needs k <= 16")
            inval = self.inval_work if use_work else self.inval_orig
            check(_ffi.lib().kmap_counts_hist_packed_dev(dc._h, self.codes.ptr, inval.ptr, self.n, self.borders.ptr,
                                                         self.n_seq, k, int(dedupe), None))
            p, nb = _ffi.vp(), _ffi.i64(0)
            check(_ffi.lib().kmap_counts_bins(dc._h, C.byref(p), C.byref(nb)))
            bins = torch.as_tensor(_DevArray(p.value, 4 ** k, "<i4"), device="cuda")   # int32 sum wraps like uint32
            dist.all_reduce(bins, op=dist.ReduceOp.SUM, group=group)      # stream-ordered after the histogram kernels
            nu = _ffi.i64(0)
            check(_ffi.lib().kmap_counts_finish(dc._h, k, int(merge_revcom), C.byref(nu), None))
            dc.k, dc.n_uniq = k, nu.value
            return dc.n_uniq

        def scan(self, k, consensus_kh, radius, revcom):
            hits, pos = DeviceSeq.scan(self, k, consensus_kh, radius, revcom)
            all_hits = all_gather_concat(dist, hits, reads_of, group)                   # read order = rank order
            return all_hits, all_gather_concat(dist, pos, None, group)

    ds = DistDeviceSeq(local_seq, local_borders)
    ds.out_n_seq = len(borders)
    ds.out_read_len = (borders[:, 1] - borders[:, 0]).astype(np.int64)
    return ds
