"""Row-sharded multi-GPU execution of the Hamming / smoothing / embedding stages and read-sharded counting / scanning
(one process per GPU, torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).

Sharding: every rank holds all N hashes (<= 1.6 MB) and owns row blocks of the N x N problem:
  * Hamming matrix: a rank computes ONLY its rows of D (N^2 / G bytes of HBM per rank) and picks the 20 neighbours of those
    rows from them; the N x 20 neighbour table is completed by one all-gather of device tensors (RCCL).
  * neighbour sums: from the k-mers' base-count profiles (csrc/knn_profile.hip), which need the neighbour table but no D
    rows, for exactly the rows the rank's embedding session owns.  (k > 16 or > 4 short consensuses: the matrix-based
    fallback needs arbitrary rows of D, so D is then computed in full on every rank.)
  * embedding loop: ONE float32 all-reduce per iteration of [gradient 2 x N | the loss partial as integer limbs]
    (DistEmbedLoop).  Two layouts:
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


def barrier(dist, group=None):
    """every rank of the group has arrived: a one-element SUM all-reduce + its read-back -- what a barrier is on RCCL anyway.  torch's
    own dist.barrier() on a gloo group never returned once the key-range count's concurrent asynchronous reduces had run (all ranks
    stood in it while all-reduces around it kept completing: the five-rank rehearsal of bench.py, round 6)."""
    import torch
    t = torch.zeros(1, dtype=torch.int32, device=_coll_device(dist, group))
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return int(t.item())


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


class CountShard:
    """What a DeviceCounts handle that holds only this rank's KEY RANGE of a count table needs to answer find_motif's questions
    (reference motif_discovery.py:648,661-673) without any rank receiving the whole (k-mer, count) list: the sizes of all ranks'
    shards (shards are in key order = rank order, so a k-mer's index in the whole sorted table is its shard's offset + its local
    index), and three tiny collectives -- a SUM of one integer (total count), a SUM of top_k float64 masses per trial, an
    all-gather of <= top_k candidates per rank.  All ranks take identical decisions because all see identical merged values."""

    def __init__(self, dist, group, sizes):
        self.dist, self.group = dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.sizes = [int(v) for v in sizes]
        self.n_local = self.sizes[self.rank]
        self.offset = int(sum(self.sizes[:self.rank]))
        self.n_global = int(sum(self.sizes))
        self.dev = _coll_device(dist, group)

    def sum_int(self, v):
        import torch
        t = torch.tensor([int(v)], dtype=torch.int64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return int(t.item())

    def sum_f64(self, arr):
        """element-wise sum over the ranks of integer-valued float64 partials (exact below 2^53 in any order)"""
        import torch
        t = torch.from_numpy(np.ascontiguousarray(arr, np.float64)).to(self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy()

    def merge_topk(self, top_k, idx, kh, cnt):
        """every rank's local top-k (count descending, local index ascending) -> the top-k of the whole table by the same rule
        (count descending, index in the whole sorted table ascending): identical on every rank"""
        import torch
        mine = np.full((top_k, 3), -1, np.int64)
        m = len(idx)
        mine[:m, 0] = np.asarray(cnt, np.int64)
        mine[:m, 1] = np.asarray(idx, np.int64) + self.offset
        mine[:m, 2] = np.asarray(kh, np.uint64).view(np.int64)
        t = torch.from_numpy(mine.reshape(-1)).to(self.dev)
        parts = torch.empty(self.world * top_k * 3, dtype=torch.int64, device=self.dev)
        self.dist.all_gather_into_tensor(parts, t, group=self.group)
        allc = parts.cpu().numpy().reshape(-1, 3)
        allc = allc[allc[:, 0] > 0]                                   # kmap_counts_topk never returns a non-positive count
        order = np.lexsort((allc[:, 1], -allc[:, 0]))[:top_k]
        sel = allc[order]
        return sel[:, 1].copy(), sel[:, 2].copy().view(np.uint64), sel[:, 0].copy()

    def gather_host(self, u, c):
        """the whole table on every rank's host (the top_k > 16 path of find_motif: numpy's argpartition on the full arrays)"""
        return all_gather_concat(self.dist, u, self.sizes, self.group), all_gather_concat(self.dist, c, self.sizes, self.group)


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


MSG_EXTRA = 8          # floats behind the 2 x N gradient in an iteration's message (csrc/embed.hip MSG_EXTRA)
_TWO48 = float(1 << 48)


def loss_to_limbs(v):
    """Host mirror of the device encoding (embed.hip loss_to_limbs): a rank's float64 loss partial as an exact integer -- 48.48
    fixed point cut into six 16-bit limbs stored as float32, limb 6 = "not representable" flag, limb 7 = padding.  A float32 SUM
    all-reduce adds the limbs of up to 256 ranks without rounding, in any order."""
    out = np.zeros(MSG_EXTRA, np.float32)
    v = float(v)
    if not (v >= 0.0) or not (v < float(1 << 47)):
        out[6] = 1.0
        return out
    hi = int(v)
    lo = int((v - hi) * _TWO48)
    for i in range(3):
        out[i] = (lo >> (16 * i)) & 0xFFFF
        out[3 + i] = (hi >> (16 * i)) & 0xFFFF
    return out


def loss_from_limbs(tail):
    """the summed limbs back to float64 (embed.hip loss_from_limbs); NaN if any rank raised the flag"""
    if float(tail[6]) != 0.0:
        return float("nan")
    lo = sum(int(tail[i]) << (16 * i) for i in range(3))
    hi = sum(int(tail[3 + i]) << (16 * i) for i in range(3))
    hi += lo >> 48
    lo &= (1 << 48) - 1
    return float(hi) + float(lo) / _TWO48


def assert_default_stream():
    """The library launches on the null stream; ProcessGroupNCCL orders a collective against torch's CURRENT stream.  The two
    are the same only while the caller has not switched streams (torch.cuda.stream(...)): otherwise the collective could run
    before the kernels that fill its operand -- silently.  Cheap host-side check, no device work."""
    import torch
    if torch.cuda.is_available() and torch.cuda.current_stream() != torch.cuda.default_stream():
        raise RuntimeError("kmap_amd.distributed: call with torch's default stream current (the HIP library launches on the null "
                           "stream; collectives on another stream would not be ordered after its kernels)")


PEER_HANDLE_BYTES = 64
PEER_BUS_ID_BYTES = 32


class PeerTimeout(RuntimeError):
    """a wait of the peer-direct apply kernel ran into its bound: the segment's iterations were not applied consistently"""


class PeerExchangeError(RuntimeError):
    """the peer-direct exchange cannot be set up on this node (no peer access, an area that cannot be mapped, a store that did not
    arrive): raised on EVERY rank of the group together, before any iteration has run -- KMAP_DIST_EXCHANGE=auto falls back to the
    all-reduce on it, =direct lets it through"""


class PeerExchange:
    """The peer-direct transport of the iteration message (kmap_hip.h kmap_peer_*): every rank exports a fine-grained receive area
    through a HIP IPC handle, the handles are all-gathered once, and from then on an iteration is three kernels and no library
    call: forces into a local message -> push (the message into slot [rank] of every rank's area over the xGMI links + a release
    store of the iteration number) -> apply (waits for the world flags, adds the slots in rank order).  Works between the GPUs of
    a node and between processes that share one GPU (the rehearsal on a one-GPU box).
    Set-up is validated before it is trusted: (1) every rank's device must be this rank's own or one hipDeviceCanAccessPeer allows;
    (2) every area must map; (3) a handshake -- each rank stores a tagged word into every area, all verify their own area.  After
    each of the three steps the ranks all-reduce a success flag, so a failure anywhere raises PeerExchangeError everywhere (no rank
    is left waiting in a collective) with the handle destroyed.  KMAP_PEER_TIMEOUT_MS: bound of the apply kernel's wait (10 000)."""

    def __init__(self, session, n, dist, group=None, _corrupt_handle_of=None):
        import ctypes as C
        import torch
        from . import _ffi
        self.s, self.dist, self.group = session, dist, group
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        lib = _ffi.lib()
        dev = _coll_device(dist, group)
        self._p = None
        why = []

        def agree(ok, what):
            """all-reduce(MIN) of this rank's success: every rank learns whether ANY rank failed the step"""
            t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
            if int(t.item()) == 0:
                self._destroy_now()
                raise PeerExchangeError(f"peer exchange: {what} failed on " + ("this rank: " + "; ".join(why) if not ok else "another rank"))

        def attempt(fn, *args):
            rc = fn(*args)
            if rc != 0:
                why.append(_ffi.last_error())
            return rc == 0

        # KMAP_PEER_TIMEOUT_MS is validated HERE, inside the first agreed step: a value that only one rank has, or that only one
        # rank cannot parse, must fail on every rank together, not leave the others in a barrier
        ms_env, timeout_ms = os.environ.get("KMAP_PEER_TIMEOUT_MS"), None
        if ms_env:
            try:
                timeout_ms = int(ms_env)
                if timeout_ms <= 0:
                    raise ValueError("must be positive")
            except ValueError as e:
                why.append(f"KMAP_PEER_TIMEOUT_MS={ms_env!r}: {e}")
        h = _ffi.vp()
        ok = not why and attempt(lib.kmap_peer_create, C.byref(h), world, rank, int(lib.kmap_embed_msg_floats(n)))
        self._p = h.value if ok else None
        mine = np.zeros(PEER_HANDLE_BYTES + PEER_BUS_ID_BYTES, np.uint8)       # [IPC handle | PCI bus id]
        if ok:
            ok = attempt(lib.kmap_peer_handle, self._p, _ffi.ptr(mine)) and attempt(lib.kmap_peer_bus_id, mine[PEER_HANDLE_BYTES:].ctypes.data)
        agree(ok, "creating / exporting a receive area")
        rec = PEER_HANDLE_BYTES + PEER_BUS_ID_BYTES
        allh = torch.empty(world * rec, dtype=torch.uint8, device=dev)
        dist.all_gather_into_tensor(allh, torch.from_numpy(mine).to(dev), group=group)
        allh = np.ascontiguousarray(allh.cpu().numpy()).reshape(world, rec)
        self._handles = np.ascontiguousarray(allh[:, :PEER_HANDLE_BYTES])
        if _corrupt_handle_of is not None:                                     # fault injection (tests): a handle that maps nothing
            self._handles[_corrupt_handle_of] = 0xA5
        # (1) can this device store into every rank's device at all?
        for q in range(world):
            can = _ffi.i32(0)
            bus = np.ascontiguousarray(allh[q, PEER_HANDLE_BYTES:])
            if not attempt(lib.kmap_peer_can_access, bus.ctypes.data, C.byref(can)) or not can.value:
                ok = False
                why.append(f"no peer access to rank {q}'s device {bytes(bus).split(bytes(1))[0].decode(errors='replace')}")
        agree(ok, "the peer-access check (hipDeviceCanAccessPeer)")
        # (2) map every area
        ok = attempt(lib.kmap_peer_connect, self._p, _ffi.ptr(self._handles))
        agree(ok, "mapping the receive areas (hipIpcOpenMemHandle)")
        # (3) handshake: a tagged word from every rank into every area, then every rank reads its own area back
        token = 0x6B6D6170 << 16                                               # same on every rank
        ok = attempt(lib.kmap_peer_hello_push, self._p, token)
        agree(ok, "the handshake stores")          # doubles as the barrier between "everybody has stored" and "everybody checks"
        miss = _ffi.i32(0)
        ok = attempt(lib.kmap_peer_hello_check, self._p, token, C.byref(miss)) and miss.value == 0
        if miss.value:
            why.append(f"{miss.value} of {world} handshake words did not arrive in this rank's area")
        agree(ok, "the handshake check")
        ok = timeout_ms is None or attempt(lib.kmap_peer_set_timeout_ms, self._p, timeout_ms)
        agree(ok, "setting the wait bound (KMAP_PEER_TIMEOUT_MS)")   # doubles as the barrier: nobody pushes before every area is mapped everywhere

    def _destroy_now(self):
        from . import _ffi
        if self._p:
            _ffi.lib().kmap_peer_destroy(self._p)
            self._p = None

    def step(self, n_iter):
        from . import _ffi
        _ffi.check(_ffi.lib().kmap_embed_step_peer(self.s._h, self._p, int(n_iter), None))

    def check(self):
        """after a segment: did a wait run into its bound (a peer that never pushed)?  Blocks until the issued iterations ran."""
        import ctypes as C
        from . import _ffi
        t, it = _ffi.i32(0), _ffi.i64(0)
        _ffi.check(_ffi.lib().kmap_peer_status(self._p, C.byref(t), C.byref(it)))
        if t.value:
            raise PeerTimeout(f"peer exchange: a rank's message did not arrive within the wait bound (after {it.value} iterations issued); "
                              "the coordinates of this run are not to be used")

    def close(self):
        from . import _ffi
        if self._p:
            barrier(self.dist, self.group)        # no rank unmaps an area a peer may still push into
            _ffi.lib().kmap_peer_destroy(self._p)
            self._p = None


class DistEmbedLoop:
    """Drives a (row-sharded) embedding session with ONE collective per iteration:
        forces_msg -> all_reduce(SUM, float32 message) -> apply_msg.
    The message (kmap_hip.h, kmap_embed_forces_msg) is [gradient 2 x N | loss limbs]: a rank writes only its own entries, the
    apply kernel of a row-sharded session zeroes what it has read, so the sum is a concatenation (x + 0 + ... + 0, exact) without
    a memset launch; the loss travels as integer limbs whose float32 sums are exact in any order, so every rank decodes the
    same total and takes the same stop / snapshot decisions.  `session` needs forces_msg(ptr) / apply_msg(ptr); msg_t is a
    zero-initialised float32 torch tensor of 2 N + MSG_EXTRA elements on the session's device.
    always_collective: issue the all-reduce even on a one-rank group (bench.py measures the loop's overhead that way)."""

    def __init__(self, session, msg_t, dist=None, group=None, always_collective=False, peer: "PeerExchange" = None):
        self.s, self.m, self.dist, self.group = session, msg_t, dist, group
        self.mp = msg_t.data_ptr()
        self.n_collectives = 0
        self.coll = dist is not None and (always_collective or dist.get_world_size(group) > 1)
        self.peer = peer                          # peer-direct transport instead of the all-reduce (KMAP_DIST_EXCHANGE=direct)
        if self.coll and msg_t.is_cuda:
            assert_default_stream()

    def step(self, n_iter):
        d, s, m, mp, g = self.dist, self.s, self.m, self.mp, self.group
        if self.peer is not None:
            self.peer.step(n_iter)                # the whole segment is issued by ONE native call
            self.peer.check()
            return
        if not self.coll:
            for _ in range(n_iter):
                s.forces_msg(mp)
                s.apply_msg(mp)
            return
        for _ in range(n_iter):
            s.forces_msg(mp)
            d.all_reduce(m, op=d.ReduceOp.SUM, group=g)      # stream-ordered between the two kernels: no host sync
            s.apply_msg(mp)
        self.n_collectives += n_iter

    def profile(self, n_iter):
        """n_iter iterations with device events around the three phases -> mean ms per iteration of each (events on torch's
        current stream = the library's stream).  The event records perturb the loop a little: use step() for the headline time."""
        import torch
        if self.peer is not None:                 # three kernels issued by one native call: only the whole iteration can be timed
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self.peer.step(n_iter)
            e1.record()
            self.peer.check()
            torch.cuda.synchronize()
            return {"iteration_ms": e0.elapsed_time(e1) / n_iter}
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(n_iter)]
        d = self.dist
        for e in ev:
            e[0].record()
            self.s.forces_msg(self.mp)
            e[1].record()
            if self.coll:
                d.all_reduce(self.m, op=d.ReduceOp.SUM, group=self.group)
            e[2].record()
            self.s.apply_msg(self.mp)
            e[3].record()
        if self.coll:
            self.n_collectives += n_iter
        torch.cuda.synchronize()
        ph = [sum(e[i].elapsed_time(e[i + 1]) for e in ev) / n_iter for i in range(3)]
        total = ev[0][0].elapsed_time(ev[-1][3]) / n_iter
        return {"forces_ms": ph[0], "collective_ms": ph[1], "apply_ms": ph[2], "iteration_ms": total}


def kmap_from_kmers_distributed(samp_kh, samp_cnts, samp_label, conseq_list, kmer_len, n_neighbour=20, n_max_iter=2500,
                                learning_rate=0.01, n_best_result=10, random_seed=None, mode=None, trace=None,
                                neighbor_inds_mat=None, always_collective=False, profile_iters=0, exchange=None):
    """Multi-GPU version of visualization.kmap_from_kmers.  Call from every rank after
    torch.distributed.init_process_group("nccl") and torch.cuda.set_device(local_rank).
    neighbor_inds_mat: optional full (N, n_neighbour) table to use instead of the selection (tests inject it).
    always_collective / profile_iters: bench.py's instruments (all-reduce even on a one-rank group; that many leading
    iterations with events around forces / collective / apply, reported in trace["phases"]).
    exchange: "rccl" (one all-reduce per iteration, the default), "direct" (PeerExchange: peer-to-peer stores + flags, no
    library call between iterations; a set-up that fails its validation raises PeerExchangeError on every rank) or "auto" (direct
    where the validation passes, otherwise the all-reduce + one warning); None = the KMAP_DIST_EXCHANGE environment variable."""
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
            sel = vz.knn_select_numpy(D_d.ptr, ldd, n, n_neighbour, nrows=nrows).astype(np.int32)
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
        if nr and vz.knn_sums_kmers_dev(kh_d.ptr, lab_d.ptr, n, kmer_len, lens, nb, n_neighbour, row0=r0, nrows=nr, out=dst,
                                        natural_diag=(mode == vz.EMBED_SEQ and not cyclic)) is None:
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
    rowmap_d, stored = None, nrows
    if not cyclic and mode == vz.EMBED_SEQ and nrows:
        sums_d, rowmap_d, stored = vz.dedupe_sums_rows(sums_d, nrows, lds, n=n)     # the repeated rows of this rank's block, stored once
    try:
        sess.set_prob_lut(sums_d, lds, lut, rowmap_d, stored)
        sess.set_coords(ld_data, placeholders)
        msg_t = torch.zeros(2 * n + MSG_EXTRA, dtype=torch.float32, device="cuda")
        exchange = os.environ.get("KMAP_DIST_EXCHANGE", "rccl").lower() if exchange is None else exchange
        peer = None
        if exchange in ("direct", "auto"):
            try:
                peer = PeerExchange(sess, n, dist)
            except PeerExchangeError as e:     # raised on every rank together
                if exchange == "direct":
                    raise
                import warnings
                warnings.warn(f"{e} -- falling back to one all-reduce per iteration (KMAP_DIST_EXCHANGE=auto)")
        loop = DistEmbedLoop(sess, msg_t, dist, always_collective=always_collective, peer=peer)
        prof = {}

        def step_fn(seg):              # the first profile_iters iterations run with events around their three phases
            k = 0 if prof else min(profile_iters, seg)
            if k:
                prof.update(loop.profile(k))
            loop.step(seg - k)
        vz._run_loop(sess, n_max_iter, step_fn=step_fn if profile_iters else loop.step, trace=trace)
        if trace is not None:
            trace["hbm"] = hbm
            trace["seed"] = random_seed
            trace["collectives"] = loop.n_collectives
            trace["exchange"] = "direct" if peer is not None else "all_reduce"
            if profile_iters:
                trace["phases"] = prof
        return sess.best(), lab
    finally:
        if peer is not None:
            peer.close()
        sess.close()


# ---- read-sharded k-mer counting / masking / scanning -----------------------------------------------------------
def read_partition(borders, world, rank):
    """Contiguous read ranges cut at read borders, balanced by read count: returns (first_read, n_reads)."""
    return row_partition(len(borders), world, rank)


def _gathered_hits_cls():
    from .motif_discovery import ScanHits

    class GatheredHits(ScanHits):
        """The hit list of ALL reads of a read-sharded scan, gathered by device collectives (RCCL) and still resident in HBM:
        `parts_hits[r, :reads_of[r]]` / `parts_pos[r, :totals[r]]` are rank r's shard.  Same interface as ScanHits: the summary
        numbers are known at once, the arrays reach the host on first use (in practice: rank 0's CSV writer thread)."""

        def __init__(self, parts_hits, parts_pos, reads_of, totals, n_reads_hit, max_hits, device):
            import threading
            self._ph, self._pp, self._reads_of, self._totals, self._dev = parts_hits, parts_pos, reads_of, totals, device
            self.n_seq, self.total, self.n_reads_hit, self.max_hits = int(sum(reads_of)), int(sum(totals)), int(n_reads_hit), int(max_hits)
            self._host, self._owner, self._handle = None, None, None
            self._lock = threading.Lock()

        def _cat(self, narrow):
            import torch
            with torch.cuda.device(self._dev):          # torch's current device is per thread (CSV writer threads)
                st = torch.cuda.Stream()                 # own stream: neither waits for nor blocks the launching thread
                st.wait_stream(torch.cuda.default_stream())
                with torch.cuda.stream(st):
                    hits = torch.cat([self._ph[r, :m] for r, m in enumerate(self._reads_of)])
                    if narrow:
                        hits = hits.clamp(max=255).to(torch.uint8)
                    pos = torch.cat([self._pp[r, :m] for r, m in enumerate(self._totals)]) if self.total else torch.zeros(0, dtype=torch.int32)
                    out = hits.cpu().numpy(), pos.cpu().numpy()
                st.synchronize()
            self._ph = self._pp = None
            return out

        def host(self):
            with self._lock:
                if self._host is None:
                    if self._ph is None:
                        raise RuntimeError("GatheredHits: the list was already handed to a CSV writer (host_u8)")
                    self._host = list(self._cat(False))
                return self._host

        @property
        def unfetched(self):
            return self._host is None and self._ph is not None

        def host_u8(self):
            with self._lock:
                assert self._host is None and self._ph is not None and self.max_hits <= 255
                return self._cat(True)

        def release(self):
            """a rank that will never write this list (everyone but the owner of the occurrence files) drops the gathered device
            tensors at once -- world x (reads + positions) int32 per consensus otherwise stay in HBM until the list object dies;
            the summary numbers (total, n_reads_hit, max_hits) stay"""
            with self._lock:
                self._ph = self._pp = None

        def __del__(self):
            pass

    return GatheredHits


KEY_SPACE_MIN_K = 13     # from here on a multi-rank count owns KEY RANGES over all reads instead of read ranges + a table collective
                         # (C3 at G = 8, one rank's pass: k = 12 3.1 ms against 0.74 ms + an all-reduce of 64 MiB; k = 13 about even;
                         # k = 14 2.8 ms against 2.1 ms + an all-reduce of 1 GiB, >= 12 ms on a ring: tools/probes/keyspace_proxy.py)


def make_dist_device_seq(seq_np_arr, boarder_mat, dist, group=None, shard_counts=None, key_space=None):
    """A DeviceSeq holding only this rank's reads whose count()/scan() results are global:
    count = local histogram -> all-reduce(SUM) of the 4^k uint32 bins -> identical compaction on every rank
    (k <= 16; per-read dedupe, masking and scanning are local to a read, hence to a rank).  Because every rank then
    sees the same counts, find_motif(dev_seq=...) makes the same decisions everywhere without further exchange.
    shard_counts (True / False force it for 11 <= k <= 16; None = for k >= 15 when the reads of all ranks hold fewer than 4^k / 4
    windows): the bins are owned by key range instead -- every rank receives only the summed counts of ITS slice of the table (one
    SUM-reduce per slice), compacts that slice, and the (k-mer, count) shards are all-gathered in rank = key order.  Per rank and
    (G-1)/G: 4^k * 4 B of slices + 4^k B of presence + 12 B per distinct k-mer, against 4^k * 8 B for the all-reduce: it pays while
    the distinct k-mers number fewer than a quarter of the bins -- and above TOPK_DEVICE_MIN unique k-mers the last term is not paid
    at all: the table stays sharded (DeviceCounts._shard / CountShard) and find_motif works on local partials.  The reverse-complement merge pairs
    bins of different slices: it runs BEFORE the reduction on each rank's local table, steered by an all-reduced presence map
    (half a byte per bin) so that merged(sum over ranks) == sum over ranks(merged); see include/kmap_hip.h.
    key_space (None = for KEY_SPACE_MIN_K <= k <= 16 on more than one rank; True / False force it for 11 <= k <= 16; KMAP_DIST_KEYSPACE=0 / 1
    overrides): counting by KEY SPACE -- every rank ALSO holds all packed reads (uploaded on the first such count: 0.625 B / position) and
    computes positions [4^k r / G, 4^k (r + 1) / G) of the table from the windows that decide them alone
    (kmap_counts_run_packed_range_dev: a window whose k-mer, or else its reverse complement, lies in the range) -- no table bytes are
    exchanged at all, and the histogram passes are those of a table 2 / G the size over 2 / G of the windows.  What is exchanged
    afterwards is what the key-range form exchanges: the shard sizes, and the (k-mer, count) shards only when the table is small
    enough not to stay sharded (CountShard).  Read-sharded counting all-reduces 4^k x 4 B per count pass (1 GiB at k = 14: >= 12 ms on
    a ring over xGMI) and its table passes do not shrink with the ranks; masking is replayed on the full copy (a mask pass over all
    reads per masked consensus, not sharded).
    scan() returns the hits of ALL reads (all-gathered in read order); `out_n_seq` / `out_read_len` describe the reads those
    results cover (all of them), `n_seq` / `read_len` stay the local shard the kernels run on.
    On RCCL the hit lists never pass through the host on their way to the collective: the shards are gathered as device
    tensors (scan_lazy -> GatheredHits) and only the rank that writes the occurrence file fetches them.
    The arrays may be views of memory-mapped pickles (kmer_count.load_array_pickle): a rank touches its own slice only."""
    import ctypes as C
    import torch
    from . import _ffi
    from ._ffi import check
    from .motif_discovery import TOPK_DEVICE_MIN, DeviceSeq

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    on_dev = _coll_device(dist, group) == "cuda"
    if on_dev:
        assert_default_stream()
    borders = boarder_mat if isinstance(boarder_mat, np.ndarray) and boarder_mat.dtype == np.int64 and boarder_mat.ndim == 2 \
        else np.ascontiguousarray(boarder_mat, dtype=np.int64).reshape(-1, 2)
    r0, nr = read_partition(borders, world, rank)
    if nr:
        lo, hi = int(borders[r0, 0]), int(borders[r0 + nr - 1, 1]) + 1          # include the last read's separator
        hi = min(hi, len(seq_np_arr))
    else:
        lo = hi = 0
    local_seq = np.ascontiguousarray(seq_np_arr[lo:hi])
    local_borders = borders[r0:r0 + nr] - lo
    reads_of = [read_partition(borders, world, r)[1] for r in range(world)]
    n_all_positions = len(seq_np_arr)
    GatheredHits = _gathered_hits_cls() if on_dev else None

    class DistDeviceSeq(DeviceSeq):
        first_read, n_local_reads, n_all_reads = r0, nr, len(borders)
        _all_read_len = None
        keep_sharded = None         # key-range counting: None = keep the table sharded above TOPK_DEVICE_MIN unique k-mers; True / False force it
        full_table_rank = None      # ... and gather the whole table on this rank as well (the writer of k{k}.pkl)

        @property
        def out_read_len(self):
            """lengths of ALL reads (the reads scan() results cover), computed on first use: only the rank that writes the
            occurrence file ever asks (160 MB of borders at C3)"""
            if self._all_read_len is None:
                self._all_read_len = (borders[:, 1] - borders[:, 0]).astype(np.int64)
            return self._all_read_len

        @out_read_len.setter
        def out_read_len(self, value):      # DeviceSeq.__init__ assigns the local shard's lengths: not what scan() covers here
            pass

        # masks are LOGGED and applied on demand: to the shard before a read-sharded count of the working mask, to the full copy before
        # a key-space count -- a k that counts by key space never pays the shard's mask passes, and vice versa
        _full, _mask_log, _done_shard, _done_full = None, None, 0, 0

        def _full_seq(self):
            """all reads, packed, on THIS rank (key-space counting): uploaded on first use; pending masks applied"""
            if self._full is None:
                self._full = DeviceSeq(seq_np_arr, borders)
                self._done_full = 0
            log = self._mask_log or []
            for k_, cons_, rad_ in log[self._done_full:]:     # the full copy has the reference's own cross-separator behaviour
                DeviceSeq.mask(self._full, k_, cons_, rad_)
            self._done_full = len(log)
            return self._full

        def _sync_shard(self):
            log = self._mask_log or []
            for k_, cons_, rad_ in log[self._done_shard:]:
                self._mask_shard(k_, cons_, rad_)
            self._done_shard = len(log)

        def reset(self):
            DeviceSeq.reset(self)
            self._mask_log, self._done_shard, self._done_full = [], 0, 0
            if self._full is not None:
                self._full.reset()

        def close(self):
            if self._full is not None:
                self._full.close()
                self._full = None
            DeviceSeq.close(self)

        def download(self):
            self._sync_shard()
            return DeviceSeq.download(self)

        def mask(self, k, consensus_kh_arr, max_ham_dist_arr):
            if self._mask_log is None:
                self._mask_log = []
            self._mask_log.append((k, np.array(consensus_kh_arr, dtype=np.uint64), np.array(max_ham_dist_arr, dtype=np.int32)))

        def _mask_shard(self, k, consensus_kh_arr, max_ham_dist_arr):
            """mask_input on this rank's reads, plus the one effect that crosses a shard boundary: a window that touches a separator
            has the reference's all-ones hash ("compared like any value", kmer_count.py:580-610), so a consensus within its radius
            of the all-T k-mer also masks the window that STARTS AT the separator in front of this shard -- on the rank before --
            and with it the first k - 1 positions here."""
            DeviceSeq.mask(self, k, consensus_kh_arr, max_ham_dist_arr)
            if self.first_read == 0 or self.n == 0:
                return
            kmask = (1 << (2 * k)) - 1
            hit = False
            for c, r in zip(np.asarray(consensus_kh_arr).tolist(), np.asarray(max_ham_dist_arr).tolist()):
                x = (kmask ^ int(c)) & kmask
                hit = hit or bin((x | (x >> 1)) & 0x5555555555555555).count("1") <= int(r)
            if not hit:
                return
            # positions [0, min(k - 1, n)) become invalid: a tiny kernel on the stream DeviceSeq.mask's kernels were queued on
            # (the library's null stream), so it is ordered behind them without a host round trip
            check(_ffi.lib().kmap_inval_set_prefix_dev(self.inval_work.ptr, min(k - 1, self.n), None))

        def count(self, dc, k, dedupe, merge_revcom, use_work=True, gather_full=False):
            """gather_full: this call's table is the one k{k}.pkl is written from (find_motif's first round) -- if it stays
            sharded, rank `full_table_rank` also receives the whole of it.  The masked re-counts of the later rounds never do:
            12 B per unique k-mer over the links and a second multi-GB table on the writer rank, per round, for nothing."""
            if k > 16:
                raise ValueError("sharded counting all-reduces the 4^k histogram and needs k <= 16")
            if on_dev:
                assert_default_stream()
            ks = key_space
            if os.environ.get("KMAP_DIST_KEYSPACE") in ("0", "1"):
                ks = os.environ["KMAP_DIST_KEYSPACE"] == "1"
            if 11 <= k <= 16 and ((ks is None and k >= KEY_SPACE_MIN_K and world > 1 and shard_counts is None) or ks):
                return self._count_by_key_space(dc, k, dedupe, merge_revcom, use_work, gather_full)
            if use_work:
                self._sync_shard()
            inval = self.inval_work if use_work else self.inval_orig
            dc._unshard()
            check(_ffi.lib().kmap_counts_hist_packed_dev(dc._h, self.codes.ptr, inval.ptr, self.n, self.borders.ptr,
                                                         self.n_seq, k, int(dedupe), None))
            p, nb = _ffi.vp(), _ffi.i64(0)
            check(_ffi.lib().kmap_counts_bins(dc._h, C.byref(p), C.byref(nb)))
            bins = torch.as_tensor(_DevArray(p.value, 4 ** k, "<i4"), device="cuda")   # int32 sum wraps like uint32
            by_range = (k >= 15 and 4 * n_all_positions < 4 ** k) if shard_counts is None else bool(shard_counts)
            if by_range and 11 <= k <= 16 and world <= 15 and (world > 1 or shard_counts):   # one rank: only when forced (tests)
                return self._count_by_key_range(dc, k, merge_revcom, bins, gather_full)
            dist.all_reduce(bins, op=dist.ReduceOp.SUM, group=group)      # stream-ordered after the histogram kernels
            nu = _ffi.i64(0)
            check(_ffi.lib().kmap_counts_finish(dc._h, k, int(merge_revcom), C.byref(nu), None))
            dc.k, dc.n_uniq = k, nu.value
            return dc.n_uniq

        def _count_by_key_range(self, dc, k, merge_revcom, bins, gather_full=False):
            """bins: this rank's local 4^k table (device tensor view).  Every collective below works on device tensors (gloo
            stages them through the host itself)."""
            lib = _ffi.lib()
            n_bins = 4 ** k
            if merge_revcom:
                nib = torch.empty(n_bins // 2, dtype=torch.uint8, device="cuda")
                check(lib.kmap_counts_presence_dev(dc._h, k, nib.data_ptr(), None))
                dist.all_reduce(nib, op=dist.ReduceOp.SUM, group=group)   # <= 15 ranks: a nibble cannot carry
                check(lib.kmap_counts_merge_presence_dev(dc._h, k, nib.data_ptr(), None))
                del nib
            bounds = [(n_bins * r // world) & ~7 for r in range(world)] + [n_bins]
            owner = [dist.get_global_rank(group, r) if group is not None else r for r in range(world)]
            works = [dist.reduce(bins[bounds[r]:bounds[r + 1]], dst=owner[r], op=dist.ReduceOp.SUM, group=group, async_op=True)
                     for r in range(world)]
            for w in works:
                w.wait()
            nu = _ffi.i64(0)
            check(lib.kmap_counts_finish_range(dc._h, k, int(bool(merge_revcom)), bounds[rank], bounds[rank + 1] - bounds[rank],
                                               C.byref(nu), None))
            return self._finish_shards(dc, k, nu, gather_full)

        def _count_by_key_space(self, dc, k, dedupe, merge_revcom, use_work, gather_full):
            """this rank's key range of the table from ALL reads (no table collective); then the shard bookkeeping of the key-range form"""
            full = self._full_seq()
            n_bins = 4 ** k
            bounds = [(n_bins * r // world) & ~7 for r in range(world)] + [n_bins]
            inval = full.inval_work if use_work else full.inval_orig
            nu = _ffi.i64(0)
            dc._unshard()
            check(_ffi.lib().kmap_counts_run_packed_range_dev(dc._h, full.codes.ptr, inval.ptr, full.n, full.borders.ptr, full.n_seq, k,
                                                              int(dedupe), int(bool(merge_revcom)), bounds[rank],
                                                              bounds[rank + 1] - bounds[rank], C.byref(nu), None))
            return self._finish_shards(dc, k, nu, gather_full)

        def _finish_shards(self, dc, k, nu, gather_full):
            """dc holds this rank's shard (nu entries) of a table cut by key range"""
            lib = _ffi.lib()
            # the shards, concatenated in rank order, are the table every rank would have compacted from the all-reduced bins
            sizes = torch.empty(world, dtype=torch.int64, device="cuda")
            dist.all_gather_into_tensor(sizes, torch.tensor([nu.value], dtype=torch.int64, device="cuda"), group=group)
            sizes = [int(v) for v in sizes.cpu().numpy()]
            total = int(sum(sizes))
            dc._unshard()
            from . import _policy
            # KMAP_EXACT / general.exact: find_motif calls np.argpartition on the whole fetched table -> every rank holds it
            keep = self.keep_sharded if self.keep_sharded is not None else (total > TOPK_DEVICE_MIN and not _policy.exact())
            if keep:
                # the table STAYS sharded: find_motif's top-k and Hamming-ball masses are local partials + tiny collectives
                # (CountShard); only the rank that writes k{k}.pkl -- if any -- receives the other ranks' shards
                if gather_full and self.full_table_rank is not None:
                    full = self._gather_table(dc, k, sizes, nu.value, dst=self.full_table_rank)
                    if full is not None:
                        dc._full = full
                dc._shard = CountShard(dist, group, sizes)
                dc.k, dc.n_uniq = k, total
                return total
            all_u, all_c = self._gather_table_tensors(dc, k, sizes, nu.value, dst=None)
            torch.cuda.current_stream().synchronize()                     # adopt copies on the null stream's side of the library
            check(lib.kmap_counts_adopt_dev(dc._h, all_u.data_ptr() if total else None, all_c.data_ptr() if total else None, total, k))
            dc.k, dc.n_uniq = k, total
            return total

        def _gather_table_tensors(self, dc, k, sizes, n_mine, dst):
            """this rank's shard [uniq | cnt] and the others' -> (all_u, all_c) device tensors in key order; dst = None: on every
            rank (all-gather), else only on rank `dst` (the others return (None, None)).  One padded buffer per rank carries
            keys and counts together."""
            lib = _ffi.lib()
            up, cp, n_tab = _ffi.vp(), _ffi.vp(), _ffi.i64(0)
            check(lib.kmap_counts_table_dev(dc._h, C.byref(up), C.byref(cp), C.byref(n_tab)))
            kt, kdt, kw = ("<i4", torch.int32, 1) if k < 16 else ("<i8", torch.int64, 2)       # key width in int32 words
            cap = max(max(sizes), 1)
            mine = torch.zeros((kw + 1) * cap, dtype=torch.int32, device="cuda")
            if n_mine:
                mine[:kw * n_mine].copy_(torch.as_tensor(_DevArray(up.value, kw * n_mine, "<i4"), device="cuda"))
                mine[kw * cap:kw * cap + n_mine].copy_(torch.as_tensor(_DevArray(cp.value, n_mine, "<i4"), device="cuda"))
            if dst is None:
                flat = torch.empty(world * (kw + 1) * cap, dtype=torch.int32, device="cuda")   # flat: gloo takes no 2-D output
                dist.all_gather_into_tensor(flat, mine, group=group)
                parts = flat.view(world, (kw + 1) * cap)
            else:
                glist = [torch.empty_like(mine) for _ in range(world)] if rank == dst else None
                if on_dev:
                    dist.gather(mine, glist, dst=dist.get_global_rank(group, dst) if group is not None else dst, group=group)
                else:                                                    # gloo gathers host tensors
                    hl = [torch.empty(mine.shape, dtype=torch.int32) for _ in range(world)] if rank == dst else None
                    dist.gather(mine.cpu(), hl, dst=dist.get_global_rank(group, dst) if group is not None else dst, group=group)
                    if rank == dst:
                        glist = [t.cuda() for t in hl]
                if rank != dst:
                    return None, None
                parts = glist                                            # indexed per rank below: no stacked copy
            all_u32 = torch.cat([parts[r][:kw * sizes[r]] for r in range(world)]).contiguous()
            all_c = torch.cat([parts[r][kw * cap:kw * cap + sizes[r]] for r in range(world)]).contiguous()
            all_u = all_u32.view(kdt) if kw == 2 else all_u32
            return all_u, all_c

        def _gather_table(self, dc, k, sizes, n_mine, dst):
            """the whole table as a NEW DeviceCounts on rank dst (None elsewhere)"""
            from .kmer_count import DeviceCounts
            all_u, all_c = self._gather_table_tensors(dc, k, sizes, n_mine, dst)
            if all_u is None:
                return None
            full = DeviceCounts()
            total = int(all_c.numel())
            torch.cuda.current_stream().synchronize()
            check(_ffi.lib().kmap_counts_adopt_dev(full._h, all_u.data_ptr() if total else None, all_c.data_ptr() if total else None, total, k))
            full.k, full.n_uniq = k, total
            return full

        def _scan_gathered(self, k, consensus_kh, radius, revcom):
            """local scan -> device gather of the shards -> GatheredHits (RCCL only)"""
            assert_default_stream()
            lib = _ffi.lib()
            if self._scan is None:
                h = _ffi.vp()
                check(lib.kmap_scan_create(C.byref(h)))
                self._scan = h.value
                self.declare_layout(self._scan)
            tot, nhit, mx = _ffi.i64(0), _ffi.i64(0), _ffi.i32(0)
            check(lib.kmap_scan_run_packed_dev(self._scan, self.codes.ptr, self.inval_orig.ptr, self.n, self.borders.ptr, self.n_seq, k,
                                               int(consensus_kh), int(radius), int(revcom), C.byref(tot), self.planes.ptr, None))
            check(lib.kmap_scan_summary(self._scan, C.byref(nhit), C.byref(mx), None))
            hp, pp = _ffi.vp(), _ffi.vp()
            check(lib.kmap_scan_result_dev(self._scan, C.byref(hp), C.byref(pp), None, None))
            meta = torch.tensor([tot.value, nhit.value, mx.value], dtype=torch.int64, device="cuda")
            metas = torch.empty((world, 3), dtype=torch.int64, device="cuda")
            dist.all_gather_into_tensor(metas, meta, group=group)
            metas = metas.cpu().numpy()                                  # 24 bytes per rank: the only host hop
            totals = [int(t) for t in metas[:, 0]]
            cap_r, cap_p = max(max(reads_of), 1), max(max(totals), 1)
            mine_h = torch.zeros(cap_r, dtype=torch.int32, device="cuda")
            if self.n_seq:
                mine_h[:self.n_seq].copy_(torch.as_tensor(_DevArray(hp.value, self.n_seq, "<i4"), device="cuda"))
            mine_p = torch.zeros(cap_p, dtype=torch.int32, device="cuda")
            if tot.value:
                mine_p[:tot.value].copy_(torch.as_tensor(_DevArray(pp.value, tot.value, "<i4"), device="cuda"))
            parts_h = torch.empty((world, cap_r), dtype=torch.int32, device="cuda")
            parts_p = torch.empty((world, cap_p), dtype=torch.int32, device="cuda")
            dist.all_gather_into_tensor(parts_h, mine_h, group=group)
            dist.all_gather_into_tensor(parts_p, mine_p, group=group)
            return GatheredHits(parts_h, parts_p, reads_of, totals, int(metas[:, 1].sum()), int(metas[:, 2].max()), torch.cuda.current_device())

        def scan(self, k, consensus_kh, radius, revcom):
            if on_dev:
                return tuple(self._scan_gathered(k, consensus_kh, radius, revcom).host())
            hits, pos = DeviceSeq.scan(self, k, consensus_kh, radius, revcom)
            all_hits = all_gather_concat(dist, hits, reads_of, group)                   # read order = rank order
            return all_hits, all_gather_concat(dist, pos, None, group)

    # gloo rehearsals: hit lists are exchanged as host tensors, no device-resident variant
    DistDeviceSeq.scan_lazy = DistDeviceSeq._scan_gathered if on_dev else None
    ds = DistDeviceSeq(local_seq, local_borders)
    ds.out_n_seq = len(borders)
    return ds
