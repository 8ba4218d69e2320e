"""Host side of the embedding path: the reference's visualization.py interface
(/root/reference/src/kmap/visualization.py) over the HIP C ABI.

`kmap()` / `umap()` / `knn_smooth()` keep the reference's names, arguments and RNG protocol
(np.random.seed, init draw, n_best placeholder draws, jitter draws in stream order); the
arithmetic runs in kmap_amd/csrc/embed.hip with all state resident in HBM.  No CPU fallback.
"""
import ctypes as C
import pickle
from pathlib import Path

import numpy as np

from . import _ffi
from ._ffi import check, ptr
from .hamdist import hamdist_matrix_dev, pitch_for
from .kmer_count import FileNameDict, get_hash_dtype

EMBED_FAST, EMBED_SEQ = 0, 1


def default_mode(n=None):
    """Embedding arithmetic when the caller does not choose: SEQ -- the reference's own arithmetic (visualization.py:296-317,
    taichi_core.py:305-326: IEEE f32, row sums in ascending j, no FMA; coordinates within 1e-5 of the reference's numpy-f32
    execution) -- at EVERY N.  FAST (each unordered pair once, wavefront-parallel row sums: same per-pair values, different
    rounding of the row sums, which gradient descent amplifies -- 1e-3 relative on the loss after 200 iterations at N = 96:
    statistically equivalent embeddings, not the same digits) is opt-in: `visualization.embed_mode = "fast"` in config.toml
    or KMAP_EMBED_MODE=fast (the environment wins).  n is accepted for the callers' convenience and plays no part."""
    from . import _policy
    return EMBED_FAST if _policy.embed_mode() == "fast" else EMBED_SEQ


STAGE_TIMES = {}          # cumulative wall-clock per stage (tools/e2e.py, bench.py report it)
TRACE_SINK = None         # a dict set by a test / bench.py receives the embedding loop's trace of the next `_visualize_kmers` (losses, state, loop_s)


class _stage:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        import time
        self.t0 = time.perf_counter()

    def __exit__(self, *exc):
        import time
        _ffi.sync()
        STAGE_TIMES[self.name] = STAGE_TIMES.get(self.name, 0.0) + time.perf_counter() - self.t0


KNN_NUMPY_MAX_N = 65536   # up to here the neighbours are the reference's own np.argpartition call (0.17 s of host time at C3's N = 50 000)


def knn_mode(n):
    """'numpy' (the reference's np.argpartition on int64 rows, visualization.py:100: drop-in, tie order numpy / ISA specific; the rows
    are streamed back from the device matrix and partitioned on host threads) up to N = 65 536 -- every BASELINE config a single GPU
    embeds -- and 'device' (smallest distance, then lowest index) above, where 40+ GB of int64 rows would pass through the host.
    KMAP_KNN overrides; KMAP_EXACT=1 / config general.exact selects 'numpy' at every size."""
    import os
    from . import _policy
    forced = os.environ.get("KMAP_KNN", "").lower()
    if forced in ("numpy", "device"):
        return forced
    return "numpy" if (n <= KNN_NUMPY_MAX_N or _policy.exact()) else "device"


def knn_select_numpy(D_dev_ptr, ldd, n, n_nb, nrows=None):
    """The reference's neighbour choice (visualization.py:100: np.argpartition on int64 rows) for rows [0, nrows) of a uint8
    device matrix, streamed back in row blocks: the copy of block b + 1 runs while worker threads partition block b (introselect
    runs outside the GIL; rows are independent, so the result equals one call on the whole matrix).
    The matrix repeats every sampled k-mer count times (motif_discovery.py:759-772), the row of a repeated k-mer is the row above
    it byte for byte (C3: 17 554 distinct rows among 50 000), and np.argpartition of equal arrays is the same array: the device
    flags the rows that differ from the row above (a comparison of the bytes, nothing is assumed about the sample), compacts
    them, and only they are copied and partitioned; a repeated row takes its predecessor's result."""
    from concurrent.futures import ThreadPoolExecutor
    import os
    nrows = n if nrows is None else nrows
    if nrows == 0:
        return np.zeros((0, n_nb), np.int64)
    lib = _ffi.lib()
    fresh_d = _ffi.DeviceBuffer(nrows)
    try:
        check(lib.kmap_rows_fresh_u8_dev(D_dev_ptr, ldd, n, 0, nrows, fresh_d.ptr, None))
        fresh = fresh_d.to_numpy(np.uint8, (nrows,)).astype(bool)
    finally:
        fresh_d.free()
    idx = np.flatnonzero(fresh).astype(np.int32)
    n_part = len(idx)
    src, pitch, comp_d = D_dev_ptr, ldd, None
    if n_part < nrows:                             # repeated rows: partition the compacted distinct ones
        pitch = (n + 15) & ~15
        idx_d = _ffi.DeviceBuffer.from_numpy(idx)
        comp_d = _ffi.DeviceBuffer(n_part * pitch)
        try:
            check(lib.kmap_gather_rows_u8_dev(D_dev_ptr, ldd, n, idx_d.ptr, n_part, comp_d.ptr, pitch, None))
            _ffi.sync()
        finally:
            idx_d.free()
        src = comp_d.ptr
    blk = max(1, min(n_part, (32 << 20) // max(n, 1)))
    sub = max(1, (2 << 20) // max(n, 1))          # rows per argpartition call: their int64 copy (16 MB) stays in the last-level cache
    res = np.empty((n_part, n_nb), np.int64)

    def part(r0, rows):
        for a in range(0, len(rows), sub):
            piece = rows[a:a + sub]
            res[r0 + a:r0 + a + len(piece)] = np.argpartition(piece.astype(np.int64), n_nb, axis=1)[:, :n_nb]

    try:
        with ThreadPoolExecutor(max(1, min(16, (os.cpu_count() or 2) - 1))) as pool:
            jobs = []
            for r0 in range(0, n_part, blk):
                r1 = min(n_part, r0 + blk)
                rows = np.empty((r1 - r0, n), np.uint8)
                check(lib.kmap_memcpy2d_d2h(ptr(rows), n, src + r0 * pitch, pitch, n, r1 - r0, None))
                jobs.append(pool.submit(part, r0, rows))
                while len(jobs) > 24:                 # bound the blocks in flight (32 MB each + their int64 copies)
                    jobs.pop(0).result()
            for j in jobs:
                j.result()
    finally:
        if comp_d is not None:
            comp_d.free()
    return res if n_part == nrows else res[np.cumsum(fresh) - 1]


def knn_select_dev(D_dev_ptr, ldd, n, n_nb, row0=0, nrows=None, stream=None):
    """Device k-NN selection -> (DeviceBuffer int32 [nrows, n_nb])."""
    nrows = n - row0 if nrows is None else nrows
    nb_d = _ffi.DeviceBuffer(max(nrows, 1) * n_nb * 4)
    check(_ffi.lib().kmap_knn_select_u8_dev(D_dev_ptr, ldd, n, n_nb, row0, nrows, nb_d.ptr, stream))
    return nb_d


_LUT_CAP = 12416          # floats of LUT the force kernels cache in LDS (embed.hip F_LUT_LDS)
_JITTER_CHUNK = 4096      # normals pre-drawn per refill
_SEGMENT = 256            # iterations between host polls of the device loop state


# ---- smoothing ------------------------------------------------------------------------------------
def _pad_cols(a, ld):
    if a.shape[1] == ld and a.flags.c_contiguous:
        return a
    out = np.zeros((a.shape[0], ld), a.dtype)
    out[:, :a.shape[1]] = a
    return out


def _is_small_int_matrix(m):
    return np.issubdtype(m.dtype, np.integer) and m.size > 0 and m.min() >= 0 and m.max() <= 255


def knn_sums_dev(D_dev_ptr, ldd, nb, n, n_nb, row0=0, nrows=None, stream=None, out=None):
    """Device: integer neighbour sums of rows [row0,row0+nrows) -> DeviceBuffer of uint16 [nrows x lds]
    (out: device address of a caller-owned block to fill instead; the returned buffer is then None)."""
    nrows = n - row0 if nrows is None else nrows
    lds = (n + 127) & ~127
    own = not isinstance(nb, _ffi.DeviceBuffer)
    nb_d = _ffi.DeviceBuffer.from_numpy(np.ascontiguousarray(nb, np.int32)) if own else nb
    sums_d = _ffi.DeviceBuffer(max(nrows, 1) * lds * 2) if out is None else None
    check(_ffi.lib().kmap_knn_sums_u8_dev(D_dev_ptr, ldd, nb_d.ptr, n, n_nb, row0, nrows, sums_d.ptr if out is None else out, lds,
                                          stream))
    _ffi.sync(stream)
    if own:
        nb_d.free()
    return sums_d, lds


def dedupe_sums_rows(sums_d, nrows, lds, min_gain=0.9, free_input=True, n=None):
    """Store the repeated rows of a neighbour-sum matrix once.  A sample repeats its k-mers in runs (motif_discovery.py:759-772), the
    sums row of a repeated k-mer equals the row above it, and a SEQ session reads its rows through a map (kmap_embed_set_row_map): the
    wave's loads then touch one row per run instead of one per point (C3: 17 554 stored rows for 50 000 points; force evaluation
    1.56 -> 1.3x ms).  The device compares the bytes; nothing is assumed about the sample.
    -> (sums_d', rowmap_d, stored_rows); rowmap_d is None (and sums_d' is sums_d) when fewer than 1 - min_gain of the rows repeat.
    The input buffer is freed when a compacted copy replaces it (free_input).  n: the payload columns of a row (default: the whole
    pitch): only they are compared -- the sums kernels never write the pad columns [n, lds), and what a recycled allocation holds
    there must not make equal rows look different."""
    lib = _ffi.lib()
    fresh_d = _ffi.DeviceBuffer(max(nrows, 1))
    try:
        check(lib.kmap_rows_fresh_u8_dev(sums_d.ptr, lds * 2, (lds if n is None else n) * 2, 0, nrows, fresh_d.ptr, None))
        fresh = fresh_d.to_numpy(np.uint8, (nrows,)).astype(bool)
    finally:
        fresh_d.free()
    stored = int(fresh.sum())
    if nrows == 0 or stored > min_gain * nrows:
        return sums_d, None, nrows
    idx_d = _ffi.DeviceBuffer.from_numpy(np.flatnonzero(fresh).astype(np.int32))
    comp_d = _ffi.DeviceBuffer(stored * lds * 2)
    try:
        check(lib.kmap_gather_rows_u8_dev(sums_d.ptr, lds * 2, lds * 2, idx_d.ptr, stored, comp_d.ptr, lds * 2, None))
        _ffi.sync()
    finally:
        idx_d.free()
    if free_input:
        sums_d.free()
    rowmap_d = _ffi.DeviceBuffer.from_numpy((np.cumsum(fresh) - 1).astype(np.int32))
    return comp_d, rowmap_d, stored


CYCLIC_BLOCK_ROWS = 256   # row block of the symmetric FAST kernel (SY_R in csrc/embed.hip)


def cyclic_blocks(n, world, rank):
    """global row ranges [(row0, nrows), ...] of the 256-row blocks rank, rank + world, ... (EmbedSession(cyclic=...))"""
    nb = int(_ffi.lib().kmap_embed_cyclic_blocks(n, world, rank))
    out = []
    for b in range(nb):
        r0 = (rank + world * b) * CYCLIC_BLOCK_ROWS
        out.append((r0, min(CYCLIC_BLOCK_ROWS, n - r0)))
    return out


KNN_NATURAL_DIAG = 1 << 30      # kmap_hip.h KMAP_KNN_NATURAL_DIAG


def knn_sums_kmers_dev(kh_dev_ptr, lab_dev_ptr, n, kmer_len, conseq_lens, nb, n_nb, row0=0, nrows=None, stream=None, out=None, natural_diag=False):
    """Same sums as knn_sums_dev but from the k-mers themselves (base-count profiles, csrc/knn_profile.hip): no matrix is
    read.  Returns None when the profile kernel does not cover the request (k > 16, more than 4 short consensuses).
    natural_diag (the embedding's own calls): S[i][i] keeps the formula's value instead of the reference's 0 -- no force or loss term
    reads the diagonal, and the rows of a repeated k-mer then agree byte for byte wherever they are compared (dedupe_sums_rows: C4's
    200 000 rows are 31 298 stored rows instead of 98 177)."""
    nrows = n - row0 if nrows is None else nrows
    lds = (n + 127) & ~127
    clen = np.ascontiguousarray(conseq_lens, dtype=np.int32)
    own = not isinstance(nb, _ffi.DeviceBuffer)
    nb_d = _ffi.DeviceBuffer.from_numpy(np.ascontiguousarray(nb, np.int32)) if own else nb
    sums_d = _ffi.DeviceBuffer(max(nrows, 1) * lds * 2) if out is None else None
    dst = sums_d.ptr if out is None else out              # out: device address of a caller-owned [nrows x lds] uint16 block
    fn = _ffi.lib().kmap_knn_sums_kmers_u32_dev if get_hash_dtype(kmer_len) == np.uint32 else _ffi.lib().kmap_knn_sums_kmers_u64_dev
    rc = fn(kh_dev_ptr, lab_dev_ptr, n, kmer_len, ptr(clen) if len(clen) else None, len(clen), nb_d.ptr,
            n_nb | (KNN_NATURAL_DIAG if natural_diag else 0), row0, nrows, dst, lds, stream)
    if own:
        nb_d.free()
    if rc == -4:                      # KMAP_E_UNSUP
        if sums_d is not None:
            sums_d.free()
        return None
    check(rc)
    _ffi.sync(stream)
    return sums_d, lds


def knn_smooth(dist_mat: np.ndarray, n_neighbour: int, neighbor_inds_mat=None) -> np.ndarray:
    """Smoothed distance matrix, float32 (reference visualization.py:90-109).
    neighbor_inds_mat: optional (N, n_neighbour) indices to use instead of np.argpartition's choice
    (its tie order depends on numpy version / CPU ISA; tests inject the reference's matrix)."""
    n = len(dist_mat)
    if neighbor_inds_mat is None:
        neighbor_inds_mat = np.argpartition(dist_mat, n_neighbour, axis=1)[:, :n_neighbour]
    nb = np.ascontiguousarray(neighbor_inds_mat, np.int32)
    if _is_small_int_matrix(dist_mat):
        ldd = pitch_for(n)
        D_d = _ffi.DeviceBuffer.from_numpy(_pad_cols(np.ascontiguousarray(dist_mat, np.uint8), ldd))
        sums_d, lds = knn_sums_dev(D_d.ptr, ldd, nb, n, n_neighbour)
        sums = sums_d.to_numpy(np.uint16, (n, lds))[:, :n]
        D_d.free()
        sums_d.free()
        # exact integer sums -> f32, then the two f32 divisions of the kernel (taichi_core.py:236)
        return (sums.astype(np.float32) / np.float32(n_neighbour)) / np.float32(n_neighbour)
    D = np.ascontiguousarray(dist_mat, np.float32)
    S = np.empty((n, n), np.float32)
    check(_ffi.lib().kmap_knn_smooth_f32(ptr(D), ptr(nb), n, n_neighbour, ptr(S)))
    return S


def sigmoid(dist_mat, max_val=16.0, change_point=10.0, scale_factor=3.0):
    """reference visualization.py:199-212 (numpy expression, evaluated on the host)."""
    assert max_val > change_point > 0
    assert scale_factor > 0
    return max_val / (1 + np.exp(-scale_factor * (dist_mat - change_point)))


def hd_prob_lut(kmer_len, n_neighbour, max_sum):
    """p for every possible integer neighbour sum s: exp(-sigmoid(f32(s)/n_nb/n_nb)/0.5) as float32,
    evaluated by numpy with the reference's own expression chain (visualization.py:262,289)."""
    s = np.arange(max_sum + 1, dtype=np.float32)
    S = (s / np.float32(n_neighbour)) / np.float32(n_neighbour)
    T = sigmoid(S, 16.0, change_point=kmer_len / 2, scale_factor=0.2 * kmer_len - 0.2)
    return np.exp(-T / 0.5).astype("float32")


# ---- L3 operators ------------------------------------------------------------------------------------
def cal_ld_prob_mat_taichi(ld_data: np.ndarray, iter_mat=None):
    """q matrix incl. the [1e-3, 1-1e-3] clip (reference visualization.py:235-256)."""
    assert ld_data.shape[0] == 2 and ld_data.dtype == np.float32
    n = ld_data.shape[1]
    q = np.empty((n, n), np.float32)
    check(_ffi.lib().kmap_ld_prob_mat_f32(ptr(np.ascontiguousarray(ld_data)), n, ptr(q)))
    return q


def cross_entropy_taichi(hd_prob_mat, ld_prob_mat, iter_mat=None):
    """2 * sum_{i<j} CE (reference visualization.py:162-176); float64 accumulation on device."""
    assert hd_prob_mat.dtype == np.float32 and ld_prob_mat.dtype == np.float32
    out = C.c_float(0)
    check(_ffi.lib().kmap_cross_entropy_f32(ptr(np.ascontiguousarray(hd_prob_mat)), ptr(np.ascontiguousarray(ld_prob_mat)),
                                            len(hd_prob_mat), C.byref(out)))
    return np.float32(out.value)


def gradient_loss_taichi(hd_prob_mat, ld_prob_mat, ld_data, debug=False):
    """4 * row sums of q/(1-q)*(p-q)*(y_i-y_j), sequential f32 (reference visualization.py:131-145)."""
    assert ld_data.dtype == np.float32 and len(ld_data) == 2
    n = len(hd_prob_mat)
    g = np.empty((2, n), np.float32)
    check(_ffi.lib().kmap_gradient_loss_f32(ptr(np.ascontiguousarray(hd_prob_mat, np.float32)),
                                            ptr(np.ascontiguousarray(ld_prob_mat, np.float32)),
                                            ptr(np.ascontiguousarray(ld_data)), n, ptr(g)))
    return g


# ---- device-resident loop ------------------------------------------------------------------------------
class EmbedSession:
    """Owns a kmap_embed handle (coordinates, probabilities, snapshots and loop state in HBM)."""

    def __init__(self, n, n_best=10, learning_rate=0.01, mode=EMBED_FAST, row0=0, nrows=None, cyclic=None):
        """cyclic=(world, rank): FAST session that owns the 256-row blocks rank, rank + world, ... and evaluates every
        unordered pair once (multi-GPU; the probability rows are passed block after block, see cyclic_blocks)."""
        h = _ffi.vp()
        nrows = n - row0 if nrows is None else nrows
        if cyclic is not None:
            check(_ffi.lib().kmap_embed_create_cyclic(C.byref(h), n, int(cyclic[0]), int(cyclic[1]), n_best, learning_rate))
        else:
            check(_ffi.lib().kmap_embed_create(C.byref(h), n, row0, nrows, n_best, learning_rate, mode))
        self._h, self.n, self.n_best = h.value, n, n_best
        self._keep = []

    def set_prob_f32(self, p_dev, ld):
        self._keep.append(p_dev)
        check(_ffi.lib().kmap_embed_set_prob_f32(self._h, p_dev.ptr, ld))

    def set_prob_lut(self, sums_dev, ld, lut, rowmap=None, src_rows=None):
        """sums_dev: uint16 [rows x ld] on the device (kept alive by the session).  rowmap (SEQ sessions): int32 device buffer,
        session row -> row of sums_dev, for matrices whose repeated rows are stored once (`dedupe_sums_rows`)."""
        self._keep.append(sums_dev)
        lut = np.ascontiguousarray(lut, np.float32)
        check(_ffi.lib().kmap_embed_set_prob_lut(self._h, sums_dev.ptr, ld, ptr(lut), len(lut)))
        if rowmap is not None:
            self._keep.append(rowmap)
            check(_ffi.lib().kmap_embed_set_row_map(self._h, rowmap.ptr, int(src_rows)))

    def set_coords(self, coords, placeholders=None):
        coords = np.ascontiguousarray(coords, np.float32)
        ph = None if placeholders is None else np.ascontiguousarray(placeholders, np.float32)
        check(_ffi.lib().kmap_embed_set_coords(self._h, ptr(coords), ptr(ph)))

    def set_jitter(self, normals):
        normals = np.ascontiguousarray(normals, np.float64)
        check(_ffi.lib().kmap_embed_set_jitter(self._h, ptr(normals), len(normals)))

    def step(self, n_iter, stream=None):
        check(_ffi.lib().kmap_embed_step(self._h, n_iter, stream))

    def forces(self, grad_ptr=None, loss_ptr=None, stream=None):
        check(_ffi.lib().kmap_embed_forces(self._h, grad_ptr, loss_ptr, stream))

    def apply(self, grad_ptr=None, loss_ptr=None, stream=None):
        check(_ffi.lib().kmap_embed_apply(self._h, grad_ptr, loss_ptr, stream))

    def forces_msg(self, msg_ptr, stream=None):
        """this rank's contribution to the iteration in ONE float buffer (gradient entries + loss limbs, kmap_hip.h)"""
        check(_ffi.lib().kmap_embed_forces_msg(self._h, msg_ptr, stream))

    def apply_msg(self, msg_ptr, stream=None):
        check(_ffi.lib().kmap_embed_apply_msg(self._h, msg_ptr, stream))

    def state(self, stream=None):
        it, st, ll, bl, ju = _ffi.i64(0), _ffi.i32(0), _ffi.f32(0), _ffi.f32(0), _ffi.i32(0)
        check(_ffi.lib().kmap_embed_state(self._h, C.byref(it), C.byref(st), C.byref(ll), C.byref(bl), C.byref(ju), stream))
        return {"iters": it.value, "stopped": bool(st.value), "last_loss": ll.value, "best_loss": bl.value,
                "jitter_used": ju.value}

    def coords(self, stream=None):
        out = np.empty((2, self.n), np.float32)
        check(_ffi.lib().kmap_embed_get_coords(self._h, ptr(out), stream))
        return out

    def best(self, stream=None):
        out = np.empty((2, self.n), np.float32)
        check(_ffi.lib().kmap_embed_get_best(self._h, ptr(out), stream))
        return out

    def losses(self, max_n=1 << 16, stream=None):
        out = np.empty(max_n, np.float32)
        m = _ffi.i64(0)
        check(_ffi.lib().kmap_embed_get_losses(self._h, ptr(out), max_n, C.byref(m), stream))
        return out[:m.value].copy()

    def coords_dev_ptr(self):
        return _ffi.lib().kmap_embed_coords_dev(self._h)

    def close(self):
        if self._h:
            _ffi.lib().kmap_embed_destroy(self._h)
            self._h = None
        for b in self._keep:
            b.free()
        self._keep = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _init_draws(n_data, n_best_result, random_seed):
    """RNG protocol of the reference's umap (visualization.py:281,292-293)."""
    np.random.seed(random_seed)
    ld_data = np.random.randn(2, n_data).astype("float32")
    placeholders = np.stack([np.random.randn(2, n_data).astype("float32") for _ in range(n_best_result)]) \
        if n_best_result > 0 else np.zeros((0, 2, n_data), np.float32)
    return ld_data, placeholders


def _run_loop(sess, n_max_iter, step_fn=None, debug=False, trace=None):
    """Drive the device loop in segments; jitter normals are pre-drawn from numpy's global stream in
    order, and the stream is left exactly where the reference would leave it (one draw per jitter hit)."""
    import time
    rng_state = np.random.get_state()
    pool = np.zeros(0, np.float64)
    info = sess.state()
    _ffi.sync()
    t_loop = time.perf_counter()
    while info["iters"] < n_max_iter and not info["stopped"]:
        seg = min(_SEGMENT, n_max_iter - info["iters"])
        if len(pool) - info["jitter_used"] < 2 * seg:
            pool = np.concatenate([pool, np.random.normal(0, 0.01, _JITTER_CHUNK)])
            sess.set_jitter(pool)
        if step_fn is None:
            sess.step(seg)
        else:
            step_fn(seg)
        info = sess.state()
        if debug:
            print(f"i_iter= {info['iters']} loss= {info['last_loss']}")
    _ffi.sync()
    t_loop = time.perf_counter() - t_loop
    np.random.set_state(rng_state)
    if info["jitter_used"]:
        np.random.normal(0, 0.01, info["jitter_used"])
    if trace is not None:
        trace["loop_s"] = t_loop                    # wall time of the iterations alone (device-synchronised on both sides)
        trace["losses"] = sess.losses()
        trace["state"] = info
        trace["last_coords"] = sess.coords()
    return info


def umap(hd_dist_mat: np.ndarray, n_max_iter=2500, learning_rate=0.01, n_best_result=10, random_seed=None, debug=True,
         mode=None, trace=None) -> np.ndarray:
    """Drop-in for the reference's umap (visualization.py:270-326): transformed distance matrix in,
    lowest-loss 2 x N embedding out.  mode=EMBED_SEQ reproduces the reference's f32 summation order."""
    n = len(hd_dist_mat)
    mode = default_mode(n) if mode is None else mode
    ld_data, placeholders = _init_draws(n, n_best_result, random_seed)
    hd_prob_mat = np.exp(-hd_dist_mat / 0.5).astype("float32")          # sigma0 = 0.5 (:284,289)
    if n_max_iter <= 0 or n == 0:
        return placeholders[0] if len(placeholders) else ld_data
    ld = (n + 63) & ~63
    p_dev = _ffi.DeviceBuffer.from_numpy(_pad_cols(hd_prob_mat, ld))
    sess = EmbedSession(n, n_best_result, learning_rate, mode)
    try:
        sess.set_prob_f32(p_dev, ld)
        sess.set_coords(ld_data, placeholders)
        _run_loop(sess, n_max_iter, debug=debug, trace=trace)
        return sess.best()
    finally:
        sess.close()


def kmap(hamdist_mat: np.ndarray, kmer_len: int, n_neighbour=20, n_max_iter=2500, learning_rate=0.01, n_best_result=10,
         random_seed=None, debug=True, mode=None, neighbor_inds_mat=None, trace=None) -> np.ndarray:
    """Drop-in for the reference's kmap (visualization.py:259-267).  Integer distance matrices (the
    Hamming matrices scan_motif writes) stay on the device end to end: uint8 D -> uint16 neighbour sums
    -> LUT probabilities -> embedding loop; other matrices go through the float operators."""
    n = len(hamdist_mat)
    mode = default_mode(n) if mode is None else mode
    if _is_small_int_matrix(hamdist_mat) and n_neighbour * n_neighbour * int(hamdist_mat.max()) + 1 <= _LUT_CAP \
            and n_max_iter > 0:
        if neighbor_inds_mat is None:
            neighbor_inds_mat = np.argpartition(hamdist_mat, n_neighbour, axis=1)[:, :n_neighbour]   # :100
        ldd = pitch_for(n)
        D_d = _ffi.DeviceBuffer.from_numpy(_pad_cols(np.ascontiguousarray(hamdist_mat, np.uint8), ldd))
        sums_d, lds = knn_sums_dev(D_d.ptr, ldd, neighbor_inds_mat, n, n_neighbour)
        D_d.free()
        print("distance smoothing finished.")
        lut = hd_prob_lut(kmer_len, n_neighbour, n_neighbour * n_neighbour * int(hamdist_mat.max()))
        ld_data, placeholders = _init_draws(n, n_best_result, random_seed)
        rowmap_d, stored = None, n
        if mode == EMBED_SEQ:
            sums_d, rowmap_d, stored = dedupe_sums_rows(sums_d, n, lds, n=n)
        sess = EmbedSession(n, n_best_result, learning_rate, mode)
        try:
            sess.set_prob_lut(sums_d, lds, lut, rowmap_d, stored)
            sess.set_coords(ld_data, placeholders)
            _run_loop(sess, n_max_iter, debug=debug, trace=trace)
            out = sess.best()
        finally:
            sess.close()
        print("optimization finished.")
        return out
    trans = knn_smooth(hamdist_mat, n_neighbour, neighbor_inds_mat)
    trans = sigmoid(trans, 16.0, change_point=kmer_len / 2, scale_factor=0.2 * kmer_len - 0.2)
    print("distance smoothing finished.")
    out = umap(trans, n_max_iter=n_max_iter, learning_rate=learning_rate, n_best_result=n_best_result,
               random_seed=random_seed, debug=debug, mode=mode, trace=trace)
    print("optimization finished.")
    return out


def kmap_from_kmers(samp_kh, samp_cnts, samp_label, conseq_list, kmer_len, n_neighbour=20, n_max_iter=2500,
                    learning_rate=0.01, n_best_result=10, random_seed=None, debug=False, mode=None,
                    neighbor_inds_mat=None, trace=None):
    """Hot path used by visualize_kmers when sample_kmers.pkl is available: the Hamming matrix is computed
    on the device from the sampled hashes (never materialised as int64 on the host)."""
    kh = np.repeat(np.asarray(samp_kh), samp_cnts).astype(get_hash_dtype(kmer_len))
    mode = default_mode(len(kh)) if mode is None else mode
    lab = np.repeat(np.asarray(samp_label), samp_cnts).astype(np.int32)
    n = len(kh)
    lens = [len(c) for c in conseq_list]
    ldd = pitch_for(n)
    with _stage("hamdist_matrix"):
        kh_d, lab_d = _ffi.DeviceBuffer.from_numpy(kh), _ffi.DeviceBuffer.from_numpy(lab)
        D_d = _ffi.DeviceBuffer(n * ldd)
        hamdist_matrix_dev(kh_d.ptr, lab_d.ptr, n, kmer_len, lens, D_d.ptr, ldd)
    with _stage("knn_select"):
        if neighbor_inds_mat is None and knn_mode(n) == "device":
            neighbor_inds_mat = knn_select_dev(D_d.ptr, ldd, n, n_neighbour)
        elif neighbor_inds_mat is None:
            # drop-in neighbour choice: numpy argpartition on int64 rows, streamed back in row blocks
            neighbor_inds_mat = knn_select_numpy(D_d.ptr, ldd, n, n_neighbour)
    if trace is not None and not isinstance(neighbor_inds_mat, _ffi.DeviceBuffer):
        trace["nb"] = np.asarray(neighbor_inds_mat)            # the host-chosen neighbours (numpy mode / injected)
    with _stage("knn_sums"):
        res = knn_sums_kmers_dev(kh_d.ptr, lab_d.ptr, n, kmer_len, lens, neighbor_inds_mat, n_neighbour, natural_diag=mode == EMBED_SEQ)
        sums_d, lds = res if res is not None else knn_sums_dev(D_d.ptr, ldd, neighbor_inds_mat, n, n_neighbour)
    if isinstance(neighbor_inds_mat, _ffi.DeviceBuffer):
        neighbor_inds_mat.free()
    for b in (D_d, kh_d, lab_d):
        b.free()
    lut = hd_prob_lut(kmer_len, n_neighbour, n_neighbour * n_neighbour * kmer_len)
    ld_data, placeholders = _init_draws(n, n_best_result, random_seed)
    rowmap_d, stored = None, n
    if mode == EMBED_SEQ:
        with _stage("dedupe_sums"):
            sums_d, rowmap_d, stored = dedupe_sums_rows(sums_d, n, lds, n=n)
    sess = EmbedSession(n, n_best_result, learning_rate, mode)
    try:
        sess.set_prob_lut(sums_d, lds, lut, rowmap_d, stored)
        sess.set_coords(ld_data, placeholders)
        with _stage("embed_loop"):
            _run_loop(sess, n_max_iter, debug=debug, trace=trace)
        return sess.best(), lab
    finally:
        sess.close()


# ---- `kmap visualize_kmers` --------------------------------------------------------------------------
def _dist_context():
    """Multi-GPU launch (`python -m torch.distributed.run --nproc-per-node G -m kmap_amd visualize_kmers ...`): returns
    (dist, rank, owns_group) with the process group initialised on this rank's GPU, or (None, 0, False) single-process.
    KMAP_DIST_BACKEND / KMAP_DIST_SAME_GPU exist for rehearsals on a one-GPU box (gloo, every rank on GPU 0)."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 or os.environ.get("KMAP_DIST_DISABLE") == "1":   # KMAP_DIST_DISABLE=1: this rank runs the verb on its own
        return None, 0, False
    import torch
    import torch.distributed as dist
    dev = 0 if os.environ.get("KMAP_DIST_SAME_GPU") else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(dev)
    check(_ffi.lib().kmap_set_device(dev))
    owns = not dist.is_initialized()
    if owns:
        backend = os.environ.get("KMAP_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend)
    return dist, dist.get_rank(), owns


def _visualize_kmers(res_dir: str, debug=False, mode=None, neighbor_inds_mat=None):
    """reference visualization.py:36-87: config.toml + sample_kmer_hamdist_mat.pkl -> low_dim_data.tsv.
    Under a torch.distributed launch the compact hand-off is embedded row-sharded over all ranks (rank 0 writes the file);
    a dense int64 matrix (N <= 16384) is embedded by rank 0 alone.
    neighbor_inds_mat: optional (N, n_neighbour) neighbour table replacing the selection (np.argpartition's choice among
    ties is numpy / ISA specific; parity tests inject the table the reference run used)."""
    dist, rank, owns_group = _dist_context()
    out = _visualize_kmers_impl(res_dir, debug, mode, dist, rank, neighbor_inds_mat)
    # success path only: a rank that raised must not park in a barrier its peers may never reach -- it re-raises, exits
    # non-zero and the launcher tears the other ranks down
    if dist is not None:
        from .distributed import barrier as _dist_barrier
        _dist_barrier(dist)
        if owns_group:
            dist.destroy_process_group()
    return out


def _visualize_kmers_impl(res_dir, debug, mode, dist, rank, neighbor_inds_mat=None):
    from ._toml import load_toml
    cfg_path = Path(res_dir) / FileNameDict["config_file"]
    assert cfg_path.exists()
    cfg = load_toml(cfg_path)
    from . import _policy
    _policy.apply_config(cfg)          # optional keys general.exact / visualization.embed_mode
    if not debug:
        debug = cfg["general"]["debug"]
    vz = cfg["visualization"]
    random_seed = vz["random_seed"]
    if random_seed == "default":
        random_seed = None
    else:
        assert isinstance(random_seed, (int, float))
    warm = None
    if dist is None:
        # a fresh process pays 0.1 - 0.2 s for loading the library, the HIP runtime and the first allocation: beside the unpickling of
        # the hand-over (200 MB of int64 at the reference's default size) instead of behind it
        import threading

        def _warm():
            try:
                b = _ffi.DeviceBuffer.from_numpy(np.zeros(1 << 16, np.uint8))
                _ffi.sync()
                b.free()
            except Exception:     # noqa: BLE001 -- warm-up only: a real problem surfaces on the main thread
                pass
        warm = threading.Thread(target=_warm)
        warm.start()
    try:
        with open(Path(res_dir) / FileNameDict["sample_kmer_hamdist_mat_file"], "rb") as fh:
            kmer_len, hamdist_mat, label_arr = pickle.load(fh)
    finally:
        if warm is not None:
            warm.join()
    if hamdist_mat is None:
        # compact hand-off written by scan_motif above the int64-matrix size threshold
        with open(Path(res_dir) / FileNameDict["sample_kmer_pkl_file"], "rb") as fh:
            samp_kh, samp_cnts, samp_label, conseq_list = pickle.load(fh)
        if dist is not None:
            from .distributed import kmap_from_kmers_distributed
            ld_data, _ = kmap_from_kmers_distributed(samp_kh, samp_cnts, samp_label, conseq_list, kmer_len,
                                                     n_neighbour=vz["n_neighbour"], n_max_iter=vz["n_max_iter"],
                                                     learning_rate=vz["learning_rate"], n_best_result=vz["n_best_result"],
                                                     random_seed=random_seed, mode=mode, neighbor_inds_mat=neighbor_inds_mat)
        else:
            ld_data, _ = kmap_from_kmers(samp_kh, samp_cnts, samp_label, conseq_list, kmer_len, n_neighbour=vz["n_neighbour"],
                                         n_max_iter=vz["n_max_iter"], learning_rate=vz["learning_rate"],
                                         n_best_result=vz["n_best_result"], random_seed=random_seed, debug=debug, mode=mode,
                                         neighbor_inds_mat=neighbor_inds_mat, trace=TRACE_SINK)
            if TRACE_SINK is not None:
                TRACE_SINK["best"] = np.array(ld_data)
    elif rank != 0:
        return None      # dense hand-off: rank 0 embeds alone
    else:
        ld_data = kmap(hamdist_mat, kmer_len, n_neighbour=vz["n_neighbour"], n_max_iter=vz["n_max_iter"],
                       learning_rate=vz["learning_rate"], n_best_result=vz["n_best_result"], random_seed=random_seed,
                       debug=debug, mode=mode, neighbor_inds_mat=neighbor_inds_mat)
    if rank != 0:
        return ld_data
    # the reference's rows `f"{x:3.3f}\t{y:3.3f}\t{label}"` (visualization.py:65-72), formatted by ONE C-level % call
    n_pts = len(label_arr)
    flat = np.empty((n_pts, 3), object)
    flat[:, 0] = np.asarray(ld_data[0]).tolist()
    flat[:, 1] = np.asarray(ld_data[1]).tolist()
    flat[:, 2] = [int(v) for v in np.asarray(label_arr).tolist()]
    with open(Path(res_dir) / FileNameDict["ld_data_file"], "w+") as fh:
        fh.write("x\ty\tlabel\n" + ("%3.3f\t%3.3f\t%d\n" * n_pts) % tuple(flat.ravel().tolist()))
    print("Dimensionality reduction finished. Low dimensional embeddings generated.")
    if vz.get("gen_fig_flag"):
        print("gen_fig_flag: plotting is outside the GPU hot path of kmap_amd; low_dim_data.tsv holds the embedding.")
    return ld_data
