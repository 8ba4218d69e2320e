"""GPU parity of the smoothing + embedding path against golden traces produced by the reference
and against the CPU oracle.  Integer neighbour sums: bit-exact.  Float path tolerances are stated
per test (north star: 1e-5 abs on coordinates at a fixed seed)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def V():
    import kmap_amd.visualization as vz
    return vz


def test_knn_smooth_golden(V, golden):
    e, s = golden("embed_ops.npz"), golden("scan_testfa.npz")
    S = V.knn_smooth(s["hamdist_mat_u8"].astype(np.int64), int(e["n_nb"]), neighbor_inds_mat=e["nb"])
    assert S.dtype == np.float32
    np.testing.assert_array_equal(S, e["S"])                       # integer sums are exact in f32
    S2 = V.knn_smooth(e["small_D"], 4, neighbor_inds_mat=e["small_nb"])
    np.testing.assert_array_equal(S2, e["small_S"])
    # reference's own test shape (kmap_tests.py:579-612): vs an f64 triple loop within 1e-4
    D, nb = e["small_D"].astype(float), e["small_nb"]
    ref = np.array([[D[np.ix_(nb[i], nb[j])].sum() / 16 if i != j else 0 for j in range(10)] for i in range(10)])
    assert np.all(np.abs(S2 - ref) < 1e-4)


def test_knn_smooth_float_matrix_vs_oracle(V):
    from oracle import oracle as O
    rng = np.random.default_rng(4)
    for n, n_nb in ((37, 5), (200, 20)):
        x = rng.standard_normal((n, 3))
        D = np.sqrt(((x[:, None] - x[None]) ** 2).sum(-1))          # a Euclidean (non-integer) matrix
        nb = np.argpartition(D, n_nb, axis=1)[:, :n_nb]
        np.testing.assert_array_equal(V.knn_smooth(D, n_nb, neighbor_inds_mat=nb), O.knn_smooth(D, n_nb, nb=nb))


def test_knn_sums_chunked_rows_vs_oracle(V):
    """N above one LDS chunk is exercised with a row range (multi-GPU decomposition) at moderate N."""
    from oracle import oracle as O
    from kmap_amd import _ffi
    from kmap_amd.hamdist import pitch_for
    rng = np.random.default_rng(9)
    n, k, n_nb = 3000, 12, 20
    kh = rng.integers(0, 4 ** k, size=n, dtype=np.uint64)
    D = O.hamdist_matrix_u8(kh, np.zeros(n, np.int32), k, [k])
    nb = np.argpartition(D.astype(np.int64), n_nb, axis=1)[:, :n_nb]
    ldd = pitch_for(n)
    Dp = np.zeros((n, ldd), np.uint8)
    Dp[:, :n] = D
    D_d = _ffi.DeviceBuffer.from_numpy(Dp)
    rows = slice(1000, 1700)
    sums_d, lds = V.knn_sums_dev(D_d.ptr, ldd, nb, n, n_nb, row0=rows.start, nrows=rows.stop - rows.start)
    got = sums_d.to_numpy(np.uint16, (rows.stop - rows.start, lds))[:, :n]
    A = np.zeros((n, n), np.int32)
    np.add.at(A, (np.repeat(np.arange(n), n_nb), nb.ravel()), 1)
    want = (A[rows] @ D.astype(np.int32)) @ A.T
    want[np.arange(rows.stop - rows.start), np.arange(rows.start, rows.stop)] = 0
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("kernel", ["1", "2"])
def test_device_knn_selection_vs_oracle_rule(kernel):
    """Device k-NN selection (smallest distance, then lowest index) == stable argsort, as sets per row -- for the one-pass kernel
    (KMAP_KNN_SELECT=1: running selection under a bound from a sample of the row; the default of long rows) and the two-pass one
    (=2: histogram, then indices).  The switch is read once per process: the cases run in a child process."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = str(Path(__file__).resolve().parent.parent)
    code = "import sys; sys.path.insert(0, sys.argv[1]); import tests.test_gpu_embed as T; T._knn_select_cases(); print('cases ok')"
    r = subprocess.run([sys.executable, "-c", code, root], env=dict(os.environ, KMAP_KNN_SELECT=kernel), cwd=root, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0 and "cases ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def _knn_select_cases():
    from kmap_amd import visualization as V
    from oracle import oracle as O
    from kmap_amd import _ffi
    from kmap_amd.hamdist import pitch_for
    rng = np.random.default_rng(17)
    for n, k, n_nb in ((65, 8, 20), (1000, 8, 20), (3001, 14, 7), (200, 31, 200)):
        kh = rng.integers(0, 4 ** k, size=n, dtype=np.uint64)
        kh[:: 7] = kh[0]                                   # duplicates: distance-0 ties beyond the diagonal
        D = O.hamdist_matrix_u8(kh, np.zeros(n, np.int32), k, [k])
        ldd = pitch_for(n)
        Dp = np.zeros((n, ldd), np.uint8)
        Dp[:, :n] = D
        D_d = _ffi.DeviceBuffer.from_numpy(Dp)
        r0, nr = (n // 3, n - n // 3) if n > 100 else (0, n)
        nb_d = V.knn_select_dev(D_d.ptr, ldd, n, n_nb, row0=r0, nrows=nr)
        _ffi.sync()
        got = np.sort(nb_d.to_numpy(np.int32, (nr, n_nb)), axis=1)
        want = np.sort(O.knn_select_stable(D, n_nb)[r0:r0 + nr], axis=1)
        np.testing.assert_array_equal(got, want)
    # arbitrary byte matrices: values >= 32 (rows leave the lane-private-counter path for the generic histogram one), a pitch
    # that is no multiple of 16 (byte loads), many ties at the threshold, and rows longer than several 1024-entry steps
    for n, n_nb, hi, pad in ((700, 20, 256, 0), (5000, 20, 256, 0), (5000, 20, 40, 0), (3000, 33, 4, 5), (4100, 20, 31, 0)):
        D = rng.integers(0, hi, size=(n, n), dtype=np.uint8)
        D[np.arange(n), np.arange(n)] = 0
        ldd = (pitch_for(n) if pad == 0 else n + pad)
        Dp = rng.integers(0, 256, size=(n, ldd), dtype=np.uint8)       # padding columns hold garbage
        Dp[:, :n] = D
        D_d = _ffi.DeviceBuffer.from_numpy(Dp)
        nb_d = V.knn_select_dev(D_d.ptr, ldd, n, n_nb)
        _ffi.sync()
        got = np.sort(nb_d.to_numpy(np.int32, (n, n_nb)), axis=1)
        np.testing.assert_array_equal(got, np.sort(O.knn_select_stable(D, n_nb), axis=1))


def test_lut_matches_golden_hd_prob(V, golden):
    e = golden("embed_ops.npz")
    k, n_nb = int(e["kmer_len"]), int(e["n_nb"])
    lut = V.hd_prob_lut(k, n_nb, n_nb * n_nb * k)
    sums = np.rint(e["S"].astype(np.float64) * n_nb * n_nb).astype(np.int64)
    # numpy's f32 exp may differ by an ulp between the CPU that made the fixture and this one
    np.testing.assert_allclose(lut[sums], e["hd_prob"], rtol=2e-6, atol=1e-45)


def test_embed_ops_golden(V, golden):
    e = golden("embed_ops.npz")
    np.testing.assert_array_equal(V.cal_ld_prob_mat_taichi(e["op_ld"]), e["op_q"])          # IEEE ops only
    np.testing.assert_array_equal(V.gradient_loss_taichi(e["op_p"], e["op_q"], e["op_ld"]), e["op_grad"])
    np.testing.assert_allclose(V.cross_entropy_taichi(e["op_p"], e["op_q"]), e["op_loss"], rtol=1e-6)


@pytest.mark.parametrize("tag", ["n96", "n300"])
def test_umap_trace_seq_mode(V, golden, tag):
    """SEQ mode = the reference's arithmetic: per-iteration losses (rtol 2e-6: device log vs numpy log),
    coordinates after the last iteration and the returned best snapshot within 1e-5 abs."""
    u = golden(f"umap_{tag}.npz")
    tr = {}
    final = V.kmap(u["D"].astype(np.int64), int(u["kmer_len"]), n_max_iter=int(u["n_iter"]), random_seed=int(u["seed"]),
                   debug=False, mode=V.EMBED_SEQ, neighbor_inds_mat=u["nb"], trace=tr)
    assert len(tr["losses"]) == len(u["losses"])
    np.testing.assert_allclose(tr["losses"], u["losses"], rtol=2e-6)
    assert tr["state"]["jitter_used"] == int(u["jitter_hits"].sum())
    np.testing.assert_allclose(tr["last_coords"], u["coords"][-1], rtol=0, atol=1e-5)
    np.testing.assert_allclose(final, u["final"], rtol=0, atol=1e-5)


@pytest.mark.parametrize("tag", ["n96", "n300"])
def test_umap_trace_fast_mode(V, golden, tag):
    """FAST mode (wavefront-parallel row sums) computes the same per-pair values but rounds the row sums
    differently; gradient descent amplifies that, so it is pinned step-wise: the first iterations agree with
    the reference trace to f32 round-off and the run ends at a loss as low as the reference's (within 10 %)."""
    u = golden(f"umap_{tag}.npz")
    tr = {}
    V.kmap(u["D"].astype(np.int64), int(u["kmer_len"]), n_max_iter=int(u["n_iter"]), random_seed=int(u["seed"]),
           debug=False, mode=V.EMBED_FAST, neighbor_inds_mat=u["nb"], trace=tr)
    np.testing.assert_allclose(tr["losses"][:3], u["losses"][:3], rtol=2e-6)
    assert tr["losses"].min() <= u["losses"].min() * 1.10          # reaches an equally good optimum
    assert tr["losses"][-1] < 0.5 * tr["losses"][0]


def test_fast_and_seq_forces_agree_per_step(V, golden):
    """One force evaluation from identical coordinates: FAST vs SEQ gradients agree to f32 summation
    round-off (1e-5 relative to the gradient scale) and the losses to 1e-6 relative."""
    from kmap_amd import _ffi
    u = golden("umap_n300.npz")
    n, k, n_nb = 300, int(u["kmer_len"]), 20
    D = u["D"]
    from kmap_amd.hamdist import pitch_for
    ldd = pitch_for(n)
    Dp = np.zeros((n, ldd), np.uint8)
    Dp[:, :n] = D
    lut = V.hd_prob_lut(k, n_nb, n_nb * n_nb * int(D.max()))
    outs = {}
    for mode in (V.EMBED_SEQ, V.EMBED_FAST):
        D_d = _ffi.DeviceBuffer.from_numpy(Dp)
        sums_d, lds = V.knn_sums_dev(D_d.ptr, ldd, u["nb"], n, n_nb)
        sess = V.EmbedSession(n, 10, 0.01, mode)
        sess.set_prob_lut(sums_d, lds, lut)
        sess.set_coords(u["coords"][10])
        g_d, l_d = _ffi.DeviceBuffer(2 * n * 4), _ffi.DeviceBuffer(8)
        g_d.zero()
        sess.forces(g_d.ptr, l_d.ptr)
        _ffi.sync()
        outs[mode] = (g_d.to_numpy(np.float32, (2, n)), l_d.to_numpy(np.float64, (1,))[0])
        sess.close()
        D_d.free()
    (gs, ls), (gf, lf) = outs[V.EMBED_SEQ], outs[V.EMBED_FAST]
    assert abs(ls - lf) <= 1e-6 * abs(ls)
    np.testing.assert_allclose(gf, gs, rtol=0, atol=1e-5 * np.abs(gs).max())


def test_umap_dropin_float_matrix_path(V, golden):
    """umap(hd_dist_mat) with an f32 matrix equals the LUT path on the same data (same kernels, other source)."""
    u, e = golden("umap_n96.npz"), golden("embed_ops.npz")
    k = int(u["kmer_len"])
    S = V.knn_smooth(u["D"].astype(np.int64), 20, neighbor_inds_mat=u["nb"])
    T = V.sigmoid(S, 16.0, change_point=k / 2, scale_factor=0.2 * k - 0.2)
    a = V.umap(T, n_max_iter=50, random_seed=11, debug=False, mode=V.EMBED_SEQ)
    b = V.kmap(u["D"].astype(np.int64), k, n_max_iter=50, random_seed=11, debug=False, mode=V.EMBED_SEQ,
               neighbor_inds_mat=u["nb"])
    np.testing.assert_array_equal(a, b)


def test_rng_stream_left_like_reference(V, golden):
    """After kmap() numpy's global RNG is where the reference leaves it: seed, init + n_best draws, then one
    normal per jitter hit."""
    u = golden("umap_n96.npz")
    V.kmap(u["D"].astype(np.int64), int(u["kmer_len"]), n_max_iter=int(u["n_iter"]), random_seed=int(u["seed"]),
           debug=False, mode=V.EMBED_SEQ, neighbor_inds_mat=u["nb"])
    got = np.random.random()
    np.random.seed(int(u["seed"]))
    np.random.randn(2, 96)
    for _ in range(10):
        np.random.randn(2, 96)
    hits = int(u["jitter_hits"].sum())
    if hits:
        np.random.normal(0, 0.01, hits)
    assert got == np.random.random()


def test_early_stop_and_zero_iters(V):
    rng = np.random.default_rng(2)
    D = rng.integers(0, 9, size=(40, 40))
    D = np.triu(D, 1)
    D = D + D.T
    out0 = V.kmap(D, 8, n_neighbour=5, n_max_iter=0, random_seed=5, debug=False)
    np.random.seed(5)
    np.random.randn(2, 40)
    np.testing.assert_array_equal(out0, np.random.randn(2, 40).astype("float32"))   # first placeholder (reference :293,325)
    tr = {}
    V.kmap(D, 8, n_neighbour=5, n_max_iter=3, random_seed=5, debug=False, trace=tr)
    assert len(tr["losses"]) == 3 and tr["state"]["iters"] == 3


@pytest.mark.parametrize("tag", ["a", "b"])
def test_early_stop_golden(V, golden, tag):
    """Runs of the reference that end by its early-stop rule |loss - prev| < 1e-7 |loss| (visualization.py:310-311) before
    n_max_iter (tests/golden/umap_earlystop.npz: N = 32 stopping after 145 of 500 iterations, N = 48 after 5 of 50; large
    learning rate, every pair ends at the q clip so the loss becomes exactly constant).  The HIP loop (SEQ mode, the run's
    neighbour table injected) must raise `stopped` at the same iteration, log the same losses and return the same snapshot
    (the first iterate at the minimum loss).  Coordinates are O(1e2..1e3) here (learning rate 16), so the 1e-5 bound is taken
    relative to their extent."""
    u = golden("umap_earlystop.npz")
    D, want_losses, want_final = u[f"{tag}_D"].astype(np.int64), u[f"{tag}_losses"], u[f"{tag}_final"]
    tr = {}
    final = V.kmap(D, int(u["kmer_len"]), n_neighbour=int(u["n_nb"]), n_max_iter=int(u[f"{tag}_n_max_iter"]),
                   learning_rate=float(u[f"{tag}_lr"]), random_seed=int(u[f"{tag}_seed"]), debug=False, mode=V.EMBED_SEQ,
                   neighbor_inds_mat=u[f"{tag}_nb"], trace=tr)
    assert tr["state"]["stopped"] is True
    assert tr["state"]["iters"] == len(want_losses) < int(u[f"{tag}_n_max_iter"])
    # near the floor every term is -(1-p) ln(0.999): the device takes one log2 per eight (1-q) factors, the reference one f32 log
    # per term and an f32 pairwise sum -- a systematic 2.4e-6 relative offset of the whole trace (the loss is not part of the
    # bit-pinned path; it only has to take the same stop / snapshot decisions)
    np.testing.assert_allclose(tr["losses"], want_losses, rtol=5e-6)
    assert tr["losses"][-1] == tr["losses"][-2]                      # the rule fired on an exactly repeated loss
    scale = float(np.abs(want_final).max())
    np.testing.assert_allclose(final, want_final, rtol=0, atol=1e-5 * max(scale, 1.0))
    # FAST mode takes the same decision on this robust case (loss steps before the floor are ~1e-5 relative)
    tr2 = {}
    V.kmap(D, int(u["kmer_len"]), n_neighbour=int(u["n_nb"]), n_max_iter=int(u[f"{tag}_n_max_iter"]),
           learning_rate=float(u[f"{tag}_lr"]), random_seed=int(u[f"{tag}_seed"]), debug=False, mode=V.EMBED_FAST,
           neighbor_inds_mat=u[f"{tag}_nb"], trace=tr2)
    assert tr2["state"]["stopped"] is True and abs(tr2["state"]["iters"] - len(want_losses)) <= 2


def test_fast_vs_seq_at_default_size(V):
    """N = 5000 (the reference's default n_total_sample), k = 8: FAST (opt-in) tracks SEQ's loss curve (see default_mode() for why trajectories are not compared digit by digit); the loss decreases; best snapshot has the lowest logged loss."""
    from oracle import oracle as O
    rng = np.random.default_rng(12)
    n, k = 5000, 8
    kh = rng.integers(0, 4 ** k, size=n, dtype=np.uint64)
    lab = np.sort(rng.integers(0, 3, size=n)).astype(np.int32)
    ta, tb = {}, {}
    a, _ = V.kmap_from_kmers(kh, np.ones(n, int), lab, ["ACGTACGT", "ACGTAC"], k, n_max_iter=30, random_seed=7,
                             mode=V.EMBED_SEQ, trace=ta)
    b, _ = V.kmap_from_kmers(kh, np.ones(n, int), lab, ["ACGTACGT", "ACGTAC"], k, n_max_iter=30, random_seed=7,
                             mode=V.EMBED_FAST, trace=tb)
    np.testing.assert_allclose(tb["losses"][:2], ta["losses"][:2], rtol=2e-6)
    print("N=5000 fast-vs-seq: max rel loss diff over 30 iterations =", np.abs(tb["losses"] / ta["losses"] - 1).max(),
          " max |dcoord| =", np.abs(ta["last_coords"] - tb["last_coords"]).max())
    np.testing.assert_allclose(tb["losses"], ta["losses"], rtol=5e-2)
    assert ta["losses"][-1] < ta["losses"][0] and tb["losses"][-1] < tb["losses"][0]
    assert ta["state"]["best_loss"] == ta["losses"].min()


def test_symmetric_fast_kernel_matches_seq_per_step(V):
    """N = 20 000 (>= the symmetric FAST kernel's threshold): one force evaluation from the same coordinates agrees with
    SEQ to summation round-off, with and without the symmetric formulation."""
    import os
    from kmap_amd import _ffi
    from kmap_amd.hamdist import hamdist_matrix_dev, pitch_for
    rng = np.random.default_rng(3)
    n, k = 20000, 8
    kh = rng.integers(0, 4 ** k, size=n, dtype=np.uint64).astype(np.uint32)
    lab = np.zeros(n, np.int32)
    ldd = pitch_for(n)
    kh_d, lab_d = _ffi.DeviceBuffer.from_numpy(kh), _ffi.DeviceBuffer.from_numpy(lab)
    D_d = _ffi.DeviceBuffer(n * ldd)
    hamdist_matrix_dev(kh_d.ptr, lab_d.ptr, n, k, [k], D_d.ptr, ldd)
    nb_d = V.knn_select_dev(D_d.ptr, ldd, n, 20)
    lut = V.hd_prob_lut(k, 20, 400 * k)
    coords = rng.standard_normal((2, n)).astype(np.float32)
    outs = {}
    for tag, mode, sym in (("seq", V.EMBED_SEQ, "1"), ("sym", V.EMBED_FAST, "1"), ("fast", V.EMBED_FAST, "0")):
        os.environ["KMAP_EMBED_SYM"] = sym
        sums_d, lds = V.knn_sums_dev(D_d.ptr, ldd, nb_d, n, 20)
        sess = V.EmbedSession(n, 10, 0.01, mode)
        sess.set_prob_lut(sums_d, lds, lut)
        sess.set_coords(coords)
        g_d, l_d = _ffi.DeviceBuffer(2 * n * 4), _ffi.DeviceBuffer(8)
        g_d.zero()
        sess.forces(g_d.ptr, l_d.ptr)
        _ffi.sync()
        outs[tag] = (g_d.to_numpy(np.float32, (2, n)), l_d.to_numpy(np.float64, (1,))[0])
        sess.close()
    os.environ.pop("KMAP_EMBED_SYM", None)
    gs, ls = outs["seq"]
    for tag in ("sym", "fast"):
        g, l = outs[tag]
        assert abs(l - ls) <= 2e-6 * abs(ls), (tag, l, ls)
        np.testing.assert_allclose(g, gs, rtol=0, atol=2e-5 * np.abs(gs).max())


@pytest.mark.parametrize("scale", [1.0, 30.0, 1e3, 1e16])
def test_seq_forces_bit_exact_vs_oracle(V, scale):
    """SEQ forces == the oracle's restatement of the reference arithmetic (IEEE f32 division, j-ascending sums), bit for bit,
    on random coordinates at several scales: coincident points, far points, and squared distances beyond 1e30 (the generic
    division path; the wrapper-free division used elsewhere must not change a single bit)."""
    from kmap_amd import _ffi
    from oracle import oracle as O
    rng = np.random.default_rng(int(scale) % 1000 + 3)
    n = 701
    P = rng.random((n, n), dtype=np.float32)
    P = np.triu(P, 1)
    P = (P + P.T).astype(np.float32)
    P[rng.random((n, n)) < 0.05] = 1.0
    P = np.maximum(P, P.T)
    np.fill_diagonal(P, 0.0)
    ld = (rng.standard_normal((2, n)) * scale).astype(np.float32)
    ld[:, 5] = ld[:, 4]                                                # coincident points
    ld[:, 17] = ld[:, 16] + np.float32(1e-4 * scale)
    q = O.cal_ld_prob_mat(ld)
    want = O.gradient_loss(P, q, ld) / np.float32(4.0)
    p_d = _ffi.DeviceBuffer.from_numpy(P)
    sess = V.EmbedSession(n, 10, 0.01, V.EMBED_SEQ)
    try:
        sess.set_prob_f32(p_d, n)
        sess.set_coords(ld)
        g_d, l_d = _ffi.DeviceBuffer(2 * n * 4), _ffi.DeviceBuffer(8)
        g_d.zero()
        sess.forces(g_d.ptr, l_d.ptr)
        _ffi.sync()
        got = g_d.to_numpy(np.float32, (2, n))
        np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))
        loss = l_d.to_numpy(np.float64, (1,))[0]
        ref = float(O.cross_entropy(P, q))
        assert abs(loss * 2 - ref) <= 3e-6 * abs(ref) or abs(loss - ref) <= 3e-6 * abs(ref)
    finally:
        sess.close()


@pytest.mark.parametrize("n,k,lens", [(700, 8, [8, 8]), (1003, 8, [8, 6, 5]), (900, 12, [12, 7]), (1200, 16, [16, 9, 16, 4]),
                                      (513, 5, [5, 3]), (600, 11, [11]), (640, 16, [16, 3]), (700, 8, [8, 1, 7])])
def test_knn_sums_from_kmers_equals_matrix_sums(V, n, k, lens):
    """neighbour sums from base-count profiles (no matrix read) == the sums gathered from the Hamming matrix, incl. the
    short-consensus prefix rule, uint32 / uint64 hashes, row blocks; unsupported requests fall back (None).  The matrix-core
    kernel (v_mfma_i32_32x32x32_i8) takes every shape whose short consensuses leave it `tail` free byte slots (tail <= 4 clen);
    (16, [16, 3]) and (8, [8, 1, 7]) do not and run the v_dot4 kernel."""
    from kmap_amd import _ffi
    from kmap_amd.hamdist import hamdist_matrix_dev, pitch_for
    from kmap_amd.kmer_count import get_hash_dtype
    rng = np.random.default_rng(n + k)
    n_nb = 20
    kh = rng.integers(0, 4 ** k, size=n, dtype=np.uint64).astype(get_hash_dtype(k))
    kh[::9] = kh[0]                                                     # duplicates, as in expanded samples
    lab = np.sort(rng.integers(0, len(lens) + 1, size=n)).astype(np.int32)
    nb = np.stack([rng.choice(n, size=n_nb, replace=False) for _ in range(n)]).astype(np.int32)
    nb[3] = 7                                                           # a row whose neighbours are all the same k-mer
    kh_d, lab_d = _ffi.DeviceBuffer.from_numpy(kh), _ffi.DeviceBuffer.from_numpy(lab)
    ldd = pitch_for(n)
    D_d = _ffi.DeviceBuffer(n * ldd)
    hamdist_matrix_dev(kh_d.ptr, lab_d.ptr, n, k, lens, D_d.ptr, ldd)
    want_d, lds = V.knn_sums_dev(D_d.ptr, ldd, nb, n, n_nb)
    want = want_d.to_numpy(np.uint16, (n, lds))[:, :n]
    got_d, lds2 = V.knn_sums_kmers_dev(kh_d.ptr, lab_d.ptr, n, k, lens, nb, n_nb)
    np.testing.assert_array_equal(got_d.to_numpy(np.uint16, (n, lds2))[:, :n], want)
    r0, nr = n // 3, n // 2                                             # a row block, as the multi-GPU path asks for
    blk_d, lds3 = V.knn_sums_kmers_dev(kh_d.ptr, lab_d.ptr, n, k, lens, nb, n_nb, row0=r0, nrows=nr)
    np.testing.assert_array_equal(blk_d.to_numpy(np.uint16, (nr, lds3))[:, :n], want[r0:r0 + nr])
    assert V.knn_sums_kmers_dev(kh_d.ptr, lab_d.ptr, n, k, [max(1, k - 1)] * 5, nb, n_nb) is None   # 5 short consensuses


def test_graph_replay_switch_gives_the_same_run(V, golden):
    """KMAP_EMBED_GRAPH=1 replays a captured pair of iterations (opt-in: measured slower than direct launches on this stack);
    the switch is read once per process, so the graph run is a child process.  Same losses, same best snapshot, bit for bit."""
    import os
    import subprocess
    import sys
    import tempfile
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    code = (
        "import sys, numpy as np; sys.path.insert(0, sys.argv[1]); import kmap_amd.visualization as V\n"
        "u = np.load(sys.argv[1] + '/tests/golden/umap_n96.npz')\n"
        "tr = {}\n"
        "best = V.kmap(u['D'].astype(np.int64), int(u['kmer_len']), n_max_iter=41, random_seed=int(u['seed']), debug=False,\n"
        "              mode=int(sys.argv[3]), neighbor_inds_mat=u['nb'], trace=tr)\n"
        "np.savez(sys.argv[2], best=best, losses=tr['losses'], last=tr['last_coords'])\n")
    with tempfile.TemporaryDirectory() as td:
        for mode in (V.EMBED_SEQ, V.EMBED_FAST):
            outs = []
            for graph in ("0", "1"):
                out = os.path.join(td, f"g{graph}_{mode}.npz")
                env = dict(os.environ, KMAP_EMBED_GRAPH=graph)
                r = subprocess.run([sys.executable, "-c", code, str(root), out, str(mode)], env=env, capture_output=True, text=True, timeout=600)
                assert r.returncode == 0, r.stderr[-2000:]
                outs.append(np.load(out))
            for key in ("best", "losses", "last"):
                np.testing.assert_array_equal(outs[0][key], outs[1][key])
            assert len(outs[0]["losses"]) == 41


@pytest.mark.parametrize("mode,row0,nrows", [(1, 0, 701), (1, 200, 301), (0, 350, 351)])
def test_message_protocol_equals_two_buffer_protocol(V, mode, row0, nrows):
    """The one-message form of forces / apply (multi-GPU loop: ONE float32 all-reduce per iteration): the gradient entries are
    the two-buffer protocol's, the loss limbs decode (host mirror of the device code) to the float64 partial, a row-sharded
    session's apply zeroes the message, and message-driven iterations reproduce kmap_embed_step bit for bit."""
    from kmap_amd import _ffi
    from kmap_amd.distributed import MSG_EXTRA, loss_from_limbs
    rng = np.random.default_rng(11)
    n = 701
    P = rng.random((n, n), dtype=np.float32)
    P = np.triu(P, 1)
    P = (P + P.T).astype(np.float32)
    ld = (rng.standard_normal((2, n)) * 3).astype(np.float32)
    assert int(_ffi.lib().kmap_embed_msg_floats(n)) == 2 * n + MSG_EXTRA
    p_d = _ffi.DeviceBuffer.from_numpy(P[row0:row0 + nrows])
    sess = V.EmbedSession(n, 10, 0.01, mode, row0=row0, nrows=nrows)
    try:
        sess.set_prob_f32(p_d, n)
        sess.set_coords(ld)
        g_d, l_d, m_d = _ffi.DeviceBuffer(2 * n * 4), _ffi.DeviceBuffer(8), _ffi.DeviceBuffer((2 * n + MSG_EXTRA) * 4)
        g_d.zero()
        m_d.zero()
        sess.forces(g_d.ptr, l_d.ptr)
        sess.forces_msg(m_d.ptr)
        _ffi.sync()
        g, loss = g_d.to_numpy(np.float32, (2, n)), l_d.to_numpy(np.float64, (1,))[0]
        msg = m_d.to_numpy(np.float32, (2 * n + MSG_EXTRA,))
        np.testing.assert_array_equal(msg[:2 * n].reshape(2, n).view(np.uint32), g.view(np.uint32))
        assert np.all(msg[2 * n:2 * n + 6] == np.floor(msg[2 * n:2 * n + 6])) and msg[2 * n:2 * n + 6].max() < 65536 and msg[2 * n + 6] == 0
        got = loss_from_limbs(msg[2 * n:])
        assert abs(got - loss) <= 2.0 ** -48 + 1e-15 * abs(loss)
        if nrows == n:
            # full-row session driven through the message for 5 iterations == the resident loop
            ref = V.EmbedSession(n, 10, 0.01, mode)
            ref.set_prob_f32(_ffi.DeviceView(p_d.ptr, p_d.nbytes), n)
            ref.set_coords(ld)
            jit = np.random.default_rng(1).normal(0, 0.01, 64)
            for s in (sess, ref):
                s.set_jitter(jit)
            ref.step(5)
            for _ in range(5):
                sess.forces_msg(m_d.ptr)
                sess.apply_msg(m_d.ptr)
            np.testing.assert_array_equal(sess.coords().view(np.uint32), ref.coords().view(np.uint32))
            np.testing.assert_allclose(sess.losses(), ref.losses(), rtol=1e-7)
            ref._keep = []
            ref.close()
        else:
            sess.apply_msg(m_d.ptr)
            _ffi.sync()
            after = m_d.to_numpy(np.float32, (2 * n + MSG_EXTRA,))
            assert not after[:2 * n].any()                                 # read, then zeroed: the next sum is x + 0 + ... + 0
    finally:
        sess.close()


def test_seq_divisions_exhaustive():
    """The SEQ force kernel replaces the two IEEE f32 divisions of the reference's gradient (1 / (1 + d2), q / (1 - q)) by
    v_rcp_f32 + one Newton step / + one residual correction (csrc/seq_div.h: 7 instructions instead of 16).  That is only
    admissible if it changes no bit: checked here against the compiler's IEEE division for EVERY float operand the kernel can
    meet -- all 8.3e8 values of 1 + d2 in [1, 2^100) (as far as the clipped q is concerned) and all 8.2e7 values of the clipped
    q in [1e-3, 0.999] (1 - q is a function of q).  The shorter sequences DO differ: the test has teeth."""
    import ctypes as C
    import struct
    from kmap_amd import _ffi
    bits = lambda x: struct.unpack("<I", struct.pack("<f", x))[0]
    L = _ffi.lib()
    nb, fb = C.c_uint64(0), C.c_uint32(0)
    # the configuration compiled into the kernel (seq_div.h: KMAP_SEQ_RCP_STEPS = 1, KMAP_SEQ_QUO_RSTEPS = 0, KMAP_SEQ_QUO_STEPS = 1)
    _ffi.check(L.kmap_selftest_seq_div(0, 1, 0, bits(1.0), bits(2.0 ** 100), C.byref(nb), C.byref(fb)))
    assert nb.value == 0, f"1/(1+d2): {nb.value} operands differ from IEEE division, first bits {fb.value:#x}"
    _ffi.check(L.kmap_selftest_seq_div(1, 0, 1, bits(0.001), bits(0.999), C.byref(nb), C.byref(fb)))
    assert nb.value == 0, f"q/(1-q): {nb.value} operands differ from IEEE division, first bits {fb.value:#x}"
    # one step less is NOT exact
    _ffi.check(L.kmap_selftest_seq_div(0, 0, 0, bits(1.0), bits(2.0 ** 100), C.byref(nb), C.byref(fb)))
    assert nb.value > 1_000_000
    _ffi.check(L.kmap_selftest_seq_div(1, 0, 0, bits(0.001), bits(0.999), C.byref(nb), C.byref(fb)))
    assert nb.value > 1_000_000
    assert L.kmap_selftest_seq_div(2, 0, 0, 0, 1, C.byref(nb), None) == -1


@pytest.mark.parametrize("n,row0,nrows", [(16384 + 1029, 0, None), (16384 + 1029, 300, 16384 + 77), (5000, 0, None), (1000, 100, 650),
                                          (5003, 0, None), (4136, 7, 31), (300, 0, None), (9001, 8000, 1001)])
def test_seq_wide_tail_kernel_is_bit_identical(V, n, row0, nrows, tmp_path):
    """Rows that do not fill a round of quad waves go to the wide SEQ kernel (8 .. 64 lanes per row, terms exchanged through
    LDS, same j-ascending f32 sum): the gradient must equal the all-quad kernel's (KMAP_SEQ_TAIL=0) bit for bit, whatever the
    split -- a full round + 1029 left-over rows (one-row waves), a row-sharded session, N = 5000 (all rows wide), a small N.
    The producer / adder form (KMAP_SEQ_FORM=adder; blocks of <= 32 or 64 rows, one adder wave each) must give the same bits too:
    odd N (columns past n - 1 in the last chunk), a 31-row session, N below one chunk, a shard at the end of the rows."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    from kmap_amd import _ffi
    nrows = n - row0 if nrows is None else nrows
    rng = np.random.default_rng(n + row0)
    lds = (n + 127) & ~127
    sums = rng.integers(0, 3201, size=(nrows, lds), dtype=np.uint16)
    lut = V.hd_prob_lut(8, 20, 3200)
    ld = (rng.standard_normal((2, n)) * 5).astype(np.float32)
    np.savez(tmp_path / "in.npz", sums=sums, ld=ld, lut=lut, meta=np.array([n, row0, nrows, lds]))
    # one force evaluation per process: the A/B switch KMAP_SEQ_TAIL is read once per process
    code = ("import numpy as np, sys; sys.path.insert(0, sys.argv[1]); from kmap_amd import _ffi, visualization as V; d = np.load(sys.argv[2]);"
            "n, row0, nrows, lds = [int(v) for v in d['meta']]; sd = _ffi.DeviceBuffer.from_numpy(d['sums']);"
            "s = V.EmbedSession(n, 10, 0.01, V.EMBED_SEQ, row0=row0, nrows=nrows);"
            "_ffi.check(_ffi.lib().kmap_embed_set_prob_lut(s._h, sd.ptr, lds, _ffi.ptr(d['lut']), len(d['lut']))); s.set_coords(d['ld']);"
            "g = _ffi.DeviceBuffer(2 * n * 4); l = _ffi.DeviceBuffer(8); g.zero(); s.forces(g.ptr, l.ptr); _ffi.sync();"
            "np.savez(sys.argv[3], g=g.to_numpy(np.float32, (2, n)), l=l.to_numpy(np.float64, (1,)))")
    root = str(Path(__file__).resolve().parent.parent)
    outs = {}
    # "adder": the producer / adder form (the default of some of these shapes, forced for all of them here)
    for tag, env in (("split", {"KMAP_SEQ_FORM": "classic"}), ("quad", {"KMAP_SEQ_FORM": "classic", "KMAP_SEQ_TAIL": "0"}),
                     ("adder", {"KMAP_SEQ_FORM": "adder"})):
        r = subprocess.run([sys.executable, "-c", code, root, str(tmp_path / "in.npz"), str(tmp_path / f"{tag}.npz")],
                           env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[tag] = np.load(tmp_path / f"{tag}.npz")
    np.testing.assert_array_equal(outs["split"]["g"].view(np.uint32), outs["quad"]["g"].view(np.uint32))
    np.testing.assert_array_equal(outs["adder"]["g"].view(np.uint32), outs["quad"]["g"].view(np.uint32))
    assert abs(float(outs["adder"]["l"][0]) - float(outs["quad"]["l"][0])) <= 1e-7 * abs(float(outs["quad"]["l"][0]))
    assert outs["split"]["g"][:, row0:row0 + nrows].any() and not outs["split"]["g"][:, :row0].any()
    ls, lq = float(outs["split"]["l"][0]), float(outs["quad"]["l"][0])
    assert abs(ls - lq) <= 1e-8 * abs(lq)        # the loss is not bit-pinned: f32 partial sums over batches of different width


def test_seq_adder_form_fuzz():
    """the producer / adder form of the SEQ force kernel against the classic forms on 60 random shapes (rows: all / a few / a tail shard
    / any range; N 40 .. 20 000 incl. values that are no multiple of 4 or 8; LUT lengths of k = 3 .. 20; coordinate scales from
    coincident points to squared distances beyond 1e30): the same gradient bits, the same loss (tools/seqa_check.py)"""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools"))
    import seqa_check
    assert seqa_check.fuzz(60, seed=7) == 0


@pytest.mark.parametrize("n,row0,nrows", [(16384 + 1029, 0, None), (5003, 0, None), (1000, 100, 650), (4136, 7, 31), (9001, 8000, 1001),
                                          (33000, 0, None)])
def test_seq_row_map_is_bit_identical(V, n, row0, nrows, tmp_path):
    """SEQ sessions read the sums of repeated rows through a row map (kmap_embed_set_row_map; dedupe_sums_rows stores a run of equal
    rows once): the gradient and the loss must be those of the expanded matrix bit for bit, in the classic forms (pair / quad / wide,
    scalar and generic loads: n % 4 != 0 takes the generic ones) and in the producer / adder form; runs of 1 .. 40 rows, a run across
    every wave / block border, a session whose rows are all one run at the end."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    nrows = n - row0 if nrows is None else nrows
    rng = np.random.default_rng(n + row0 + 1)
    lds = (n + 127) & ~127
    runs = rng.choice([1, 1, 1, 2, 3, 5, 17, 40], size=nrows)
    runs = runs[:np.searchsorted(np.cumsum(runs), nrows) + 1]
    stored = len(runs)
    rowmap = np.repeat(np.arange(stored), runs)[:nrows].astype(np.int32)
    rowmap[-min(nrows, 70):] = rowmap[-min(nrows, 70)]          # the last rows: one run
    rowmap = (np.cumsum(np.concatenate([[0], np.diff(rowmap) != 0])) ).astype(np.int32)
    stored = int(rowmap[-1]) + 1
    comp = rng.integers(0, 3201, size=(stored, lds), dtype=np.uint16)
    lut = V.hd_prob_lut(8, 20, 3200)
    ld = (rng.standard_normal((2, n)) * rng.choice([5.0, 60.0])).astype(np.float32)
    np.savez(tmp_path / "in.npz", comp=comp, rowmap=rowmap, ld=ld, lut=lut, meta=np.array([n, row0, nrows, lds, stored]))
    code = ("import numpy as np, sys; sys.path.insert(0, sys.argv[1]); from kmap_amd import _ffi, visualization as V; d = np.load(sys.argv[2]);"
            "n, row0, nrows, lds, stored = [int(v) for v in d['meta']]; use_map = sys.argv[4] == 'map';"
            "sd = _ffi.DeviceBuffer.from_numpy(d['comp'] if use_map else d['comp'][d['rowmap']]);"
            "md = _ffi.DeviceBuffer.from_numpy(d['rowmap']) if use_map else None;"
            "s = V.EmbedSession(n, 10, 0.01, V.EMBED_SEQ, row0=row0, nrows=nrows);"
            "s.set_prob_lut(sd, lds, d['lut'], md, stored); s.set_coords(d['ld']);"
            "g = _ffi.DeviceBuffer(2 * n * 4); l = _ffi.DeviceBuffer(8); g.zero(); s.forces(g.ptr, l.ptr); _ffi.sync();"
            "np.savez(sys.argv[3], g=g.to_numpy(np.float32, (2, n)), l=l.to_numpy(np.float64, (1,)))")
    root = str(Path(__file__).resolve().parent.parent)
    outs = {}
    for form in ("classic", "adder"):
        for how in ("full", "map"):
            r = subprocess.run([sys.executable, "-c", code, root, str(tmp_path / "in.npz"), str(tmp_path / f"{form}_{how}.npz"), how],
                               env=dict(os.environ, KMAP_SEQ_FORM=form), capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            outs[form, how] = np.load(tmp_path / f"{form}_{how}.npz")
    ref = outs["classic", "full"]
    assert ref["g"][:, row0:row0 + nrows].any()
    for key, o in outs.items():
        np.testing.assert_array_equal(o["g"].view(np.uint32), ref["g"].view(np.uint32), err_msg=str(key))
        assert float(o["l"][0]) == float(outs[key[0], "full"]["l"][0]), key      # same form, same partial sums: the same loss bits


def test_seq_row_map_is_validated(V):
    """kmap_embed_set_row_map: SEQ sessions only; the map starts at 0, ends at src_rows - 1, steps by 0 or 1; NULL removes it"""
    from kmap_amd import _ffi
    n = 300
    lds = (n + 127) & ~127
    lut = V.hd_prob_lut(8, 20, 3200)
    sums_d = _ffi.DeviceBuffer.from_numpy(np.zeros((n, lds), np.uint16))
    good = np.repeat(np.arange(100), 3).astype(np.int32)
    lib = _ffi.lib()
    s = V.EmbedSession(n, 2, 0.01, V.EMBED_SEQ)
    try:
        s.set_prob_lut(sums_d, lds, lut)
        for bad, rows in ((good[::-1].copy(), 100), (good + 1, 101), (good * 2, 199), (good, 101), (good, 0)):
            md = _ffi.DeviceBuffer.from_numpy(np.ascontiguousarray(bad))
            assert lib.kmap_embed_set_row_map(s._h, md.ptr, rows) != 0
            md.free()
        md = _ffi.DeviceBuffer.from_numpy(good)
        _ffi.check(lib.kmap_embed_set_row_map(s._h, md.ptr, 100))
        _ffi.check(lib.kmap_embed_set_row_map(s._h, None, 0))
        md.free()
    finally:
        s.close()
    f = V.EmbedSession(n, 2, 0.01, V.EMBED_FAST)
    try:
        sums2_d = _ffi.DeviceBuffer.from_numpy(np.zeros((n, lds), np.uint16))
        f.set_prob_lut(sums2_d, lds, lut)
        md = _ffi.DeviceBuffer.from_numpy(good)
        assert lib.kmap_embed_set_row_map(f._h, md.ptr, 100) != 0
        md.free()
    finally:
        f.close()


@pytest.mark.parametrize("n,k,lens", [(700, 8, [8, 8]), (900, 12, [12, 7]), (640, 16, [16, 3])])
def test_knn_sums_natural_diagonal(V, n, k, lens):
    """KMAP_KNN_NATURAL_DIAG: S[i][i] = sum of D over nb(i) x nb(i) instead of the reference's 0 (visualization.py:36-38 sets the
    diagonal of the smoothed matrix to 0 after the fact); every other entry is unchanged, in the matrix-core and the v_dot4 kernel,
    whole and as a row block.  With it two rows of the same k-mer and the same neighbours are equal, which is what lets
    dedupe_sums_rows store them once."""
    from kmap_amd import _ffi
    from kmap_amd.hamdist import hamdist_matrix_dev, pitch_for
    from kmap_amd.kmer_count import get_hash_dtype
    rng = np.random.default_rng(3 * n + k)
    n_nb = 20
    kh = rng.integers(0, 4 ** k, size=n, dtype=np.uint64).astype(get_hash_dtype(k))
    kh[1::2] = kh[0::2]                                                 # every k-mer twice, next to each other
    lab = np.repeat(np.sort(rng.integers(0, len(lens) + 1, size=n // 2)), 2).astype(np.int32)
    nb = np.repeat(np.stack([rng.choice(n, size=n_nb, replace=False) for _ in range(n // 2)]), 2, axis=0).astype(np.int32)
    kh_d, lab_d = _ffi.DeviceBuffer.from_numpy(kh), _ffi.DeviceBuffer.from_numpy(lab)
    ldd = pitch_for(n)
    D_d = _ffi.DeviceBuffer(n * ldd)
    hamdist_matrix_dev(kh_d.ptr, lab_d.ptr, n, k, lens, D_d.ptr, ldd)
    D = D_d.to_numpy(np.uint8, (n, ldd))[:, :n].astype(np.int64)
    zero_d, lds = V.knn_sums_kmers_dev(kh_d.ptr, lab_d.ptr, n, k, lens, nb, n_nb)
    zero = zero_d.to_numpy(np.uint16, (n, lds))[:, :n]
    nat_d, lds2 = V.knn_sums_kmers_dev(kh_d.ptr, lab_d.ptr, n, k, lens, nb, n_nb, natural_diag=True)
    nat = nat_d.to_numpy(np.uint16, (n, lds2))[:, :n]
    assert not zero.diagonal().any()
    off = ~np.eye(n, dtype=bool)
    np.testing.assert_array_equal(nat[off], zero[off])
    want_diag = np.array([D[np.ix_(nb[i], nb[i])].sum() for i in range(n)])
    np.testing.assert_array_equal(nat.diagonal().astype(np.int64), want_diag)
    np.testing.assert_array_equal(nat[0::2], nat[1::2])                 # the twin rows are equal only with the natural diagonal
    assert (zero[0::2] != zero[1::2]).any(axis=1).all()
    r0, nr = n // 4, n // 2
    blk_d, lds3 = V.knn_sums_kmers_dev(kh_d.ptr, lab_d.ptr, n, k, lens, nb, n_nb, row0=r0, nrows=nr, natural_diag=True)
    np.testing.assert_array_equal(blk_d.to_numpy(np.uint16, (nr, lds3))[:, :n], nat[r0:r0 + nr])
    _, _, stored_nat = V.dedupe_sums_rows(nat_d, n, lds2, n=n)
    assert stored_nat <= n // 2


def test_seq_run_is_the_same_with_either_diagonal(V, monkeypatch):
    """The SEQ embedding never reads S[i][i] (no force or loss term pairs a point with itself -- visualization.py:116-118 skips
    i == j), so the run with the natural diagonal (fewer stored rows) equals the run with the reference's zero diagonal bit for bit."""
    rng = np.random.default_rng(77)
    k, conseqs = 8, ["ACGTACGT", "TTGACA"]
    samp_kh = rng.integers(0, 4 ** k, size=260, dtype=np.uint64)
    samp_cnts = rng.integers(1, 5, size=260)                            # repeated k-mers, as visualize_kmers expands them
    samp_label = np.sort(rng.integers(0, 3, size=260))
    runs = {}
    real = V.knn_sums_kmers_dev
    for tag in ("natural", "zero"):
        if tag == "zero":
            monkeypatch.setattr(V, "knn_sums_kmers_dev", lambda *a, **kw: real(*a, **{**kw, "natural_diag": False}))
        tr = {}
        best, _ = V.kmap_from_kmers(samp_kh, samp_cnts, samp_label, conseqs, k, n_max_iter=60, random_seed=5, debug=False,
                                    mode=V.EMBED_SEQ, trace=tr)
        runs[tag] = (np.asarray(best), np.asarray(tr["losses"]), np.asarray(tr["last_coords"]))
    for a, b in zip(runs["natural"], runs["zero"]):
        np.testing.assert_array_equal(a, b)
