"""BASELINE.json's configurations at their FULL sizes on one MI355X (C3: 10 M x 150 bp reads, N = 50 000 sampled k-mers; C4: N =
200 000; C5: 50 M x 300 bp reads, k = 14, radius 5).  An O(reads) / O(N^2) CPU recount is not feasible inside a test, so every
stage is checked by size-independent properties plus the CPU oracle on sampled reads / rows:
  counting   additivity over disjoint read sets (the histogram is linear, the revcom merge too), closed-form totals, and the
             oracle's counts on sampled sub-slices;
  masking    the masked array on sampled read chunks == the oracle's mask_input (masking never leaves a read);
  scan       per-read hits of >= 10^4 sampled reads == the oracle's ko_scan_read;
  Hamming    sampled rows == the oracle's rows, symmetry, zero diagonal;
  embedding  one force evaluation at N = 50 000: symmetric FAST kernel vs the reference-order SEQ kernel."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
K = 8


def _dense(u, c, k):
    d = np.zeros(4 ** k, np.int64)
    d[u.astype(np.int64)] = c
    return d


@pytest.fixture(scope="module")
def c3_reads():
    from kmap_amd.e2e import synth_config_reads
    return synth_config_reads("C3")            # 10 M x 150 bp, seed 2 (1.51e9 positions)


@pytest.fixture(scope="module")
def c3_dev(c3_reads):
    from kmap_amd.motif_discovery import DeviceSeq
    ds = DeviceSeq(*c3_reads)
    yield ds
    ds.close()


def _slice_reads(seq, borders, r0, r1):
    lo, hi = int(borders[r0, 0]), int(borders[r1 - 1, 1]) + 1
    return np.ascontiguousarray(seq[lo:hi]), borders[r0:r1] - lo


def test_c3_counting_full_size(c3_reads, c3_dev):
    from kmap_amd.kmer_count import DeviceCounts
    from kmap_amd.motif_discovery import DeviceSeq
    from oracle import oracle as O
    seq, borders = c3_reads
    n_reads, read_len = len(borders), int(borders[0, 1] - borders[0, 0])
    dc = DeviceCounts()
    full = {}
    for dedupe in (False, True):
        c3_dev.count(dc, K, dedupe=dedupe, merge_revcom=True)
        u, c = dc.fetch()
        assert len(np.unique(u)) == len(u) and c.min() > 0
        full[dedupe] = _dense(u, c, K)
    # closed form: no N in the synthetic reads -> every window counts once, palindromic 8-mers twice (merge_revcom doubles them)
    c3_dev.count(dc, K, dedupe=False, merge_revcom=False)
    u0, c0 = dc.fetch()
    assert int(c0.sum()) == n_reads * (read_len - K + 1)
    pal = np.array([int(O.revcom_hash(int(h), K)) == int(h) for h in u0])
    assert int(full[False].sum()) == int(c0.sum()) + int(c0[pal].sum())
    assert np.all(full[True] <= full[False]) and full[True].sum() < full[False].sum()
    # additivity over 8 disjoint read slices (uneven cuts), both modes
    cuts = [0, 1_000_003, 2_345_678, 3_999_999, 5_000_000, 6_543_210, 8_000_001, 9_123_456, n_reads]
    acc = {False: np.zeros(4 ** K, np.int64), True: np.zeros(4 ** K, np.int64)}
    for a, b in zip(cuts, cuts[1:]):
        s, bd = _slice_reads(seq, borders, a, b)
        ds = DeviceSeq(s, bd)
        for dedupe in (False, True):
            ds.count(dc, K, dedupe=dedupe, merge_revcom=True)
            acc[dedupe] += _dense(*dc.fetch(), K)
        ds.close()
    for dedupe in (False, True):
        np.testing.assert_array_equal(acc[dedupe], full[dedupe])
    # the oracle on sampled sub-slices (k = 8 and the partitioned k = 14 path)
    for a, k in ((777_777, 8), (9_400_000, 8), (4_200_000, 14)):
        s, bd = _slice_reads(seq, borders, a, a + 40_000)
        ds = DeviceSeq(s, bd)
        for dedupe in (False, True):
            ds.count(dc, k, dedupe=dedupe, merge_revcom=True)
            u, c = dc.fetch()
            ou, oc = O.count_kmers(s, bd, k, rep_mode=not dedupe, revcom_mode=True)
            np.testing.assert_array_equal(u, ou)
            np.testing.assert_array_equal(c, oc)
        ds.close()
    dc.close()


@pytest.mark.parametrize("k,G", [(13, 8), (14, 8), (14, 3), (16, 8)])
def test_c3_key_space_shards_full_size(c3_dev, k, G):
    """Counting by key space at C3's full size (1.51e9 positions; round 6, reference kmer_count.py:476-491,643-685): the G shards of
    kmap_counts_run_packed_range_dev -- each from the windows that decide it alone -- concatenated in rank order are the one-GPU
    table, key for key and count for count, with the per-read dedupe of find_motif's first round and without (k = 16: 16-GiB table,
    64-bit keys, the two-level partition behind the staging pass)."""
    from kmap_amd.kmer_count import DeviceCounts
    dc = DeviceCounts()
    n_bins = 4 ** k
    bounds = [(n_bins * r // G) & ~7 for r in range(G)] + [n_bins]
    try:
        for dedupe in ((True, False) if k == 14 and G == 8 else (False,)):
            c3_dev.count(dc, k, dedupe=dedupe, merge_revcom=True)
            u, c = dc.fetch()
            at = 0
            for r in range(G):
                c3_dev.count_range(dc, k, dedupe, True, bounds[r], bounds[r + 1] - bounds[r])
                su, sc = dc.fetch()
                assert np.array_equal(su, u[at:at + len(su)]) and np.array_equal(sc, c[at:at + len(sc)]), (k, G, dedupe, r)
                at += len(su)
            assert at == len(u)
    finally:
        dc.close()


def test_c3_masking_full_size(c3_reads, c3_dev):
    from kmap_amd.kmer_count import kmer2hash, revcom_hash
    from oracle import oracle as O
    seq, borders = c3_reads
    cons = int(kmer2hash("CCTACGTA"))
    ck = np.array([cons, int(revcom_hash(cons, K))], np.uint64)
    c3_dev.reset()
    c3_dev.mask(K, ck, np.array([2, 2]))
    got = c3_dev.download()
    c3_dev.reset()
    assert got.shape == seq.shape
    changed = got != seq
    assert np.all(got[changed] == 255) and 0.005 < changed.mean() < 0.5           # only ever turns bases into 255
    rng = np.random.default_rng(1)
    for a in rng.integers(0, len(borders) - 5000, size=6):
        lo, hi = int(borders[a, 0]), int(borders[a + 5000 - 1, 1]) + 1
        want = O.mask_input(seq[lo:hi].copy(), K, ck, np.array([2, 2]))
        np.testing.assert_array_equal(got[lo:hi], want)


def _check_scan(seq, borders, hits, pos, k, cons, radius, sample):
    from oracle import oracle as O
    offs = np.concatenate([[0], np.cumsum(hits, dtype=np.int64)])
    buf, md = np.empty(int(borders[0, 1] - borders[0, 0]) + 1, np.int32), C.c_int(0)
    L = O.lib()
    n_hit = 0
    for i in sample:
        st, en = int(borders[i, 0]), int(borders[i, 1])
        m = L.ko_scan_read(np.ascontiguousarray(seq[st:en]), en - st, k, cons, radius, 1, buf, C.byref(md))
        assert hits[i] == m, (i, hits[i], m)
        np.testing.assert_array_equal(pos[offs[i]:offs[i + 1]], buf[:m])
        n_hit += m > 0
    return n_hit


def test_c3_scan_full_size(c3_reads, c3_dev):
    from kmap_amd.kmer_count import kmer2hash
    seq, borders = c3_reads
    rng = np.random.default_rng(2)
    sample = np.concatenate([[0, len(borders) - 1], rng.integers(0, len(borders), size=12_000)])
    for conseq, radius in (("CCTACGTA", 2), ("ATCGATA", 1)):
        hits, pos = c3_dev.scan(len(conseq), kmer2hash(conseq), radius, True)
        assert len(hits) == len(borders) and int(hits.sum()) == len(pos)
        frac = np.count_nonzero(hits) / len(hits)
        assert 0.2 < frac < 0.99                                                   # 40 % of the reads carry each planted motif, chance hits on top
        assert _check_scan(seq, borders, hits, pos, len(conseq), int(kmer2hash(conseq)), radius, sample) > 3000


C5_MOTIF = "AGGACCTACGTACA"


@pytest.fixture(scope="module")
def c5_reads():
    from kmap_amd import synth
    seq, borders = synth.synth_reads(50_000_000, 300, 3, motifs=(C5_MOTIF, "AATCGATAGC"))
    assert len(seq) == 15_050_000_000
    return seq, borders


def test_c5_scan_full_size(c5_reads, monkeypatch):
    """BASELINE config C5: k = 14, max_ham_dist = 5 Hamming-ball scan over 50 M x 300 bp reads (1.505e10 positions)."""
    from kmap_amd.kmer_count import kmer2hash
    from kmap_amd.motif_discovery import DeviceSeq
    motif = C5_MOTIF
    seq, borders = c5_reads
    ds = DeviceSeq(seq, borders)
    hits, pos = ds.scan(14, kmer2hash(motif), 5, True)
    # The hit planes come from a kernel that selects its operands through the VGPR index mode (bitslice.hip): a vector instruction
    # right behind the scalar index change saw the OLD index in ~1 wave of 10^4 -- a different set of waves every run, visible
    # only at this size -- until a wait state was put between them.  The formulation without the index mode is the witness: the
    # whole result must be the same, run after run.
    monkeypatch.setenv("KMAP_SCAN_PLANES", "plain")
    hits_plain, pos_plain = ds.scan(14, kmer2hash(motif), 5, True)
    monkeypatch.delenv("KMAP_SCAN_PLANES")
    for rep in range(3):
        h2, p2 = ds.scan(14, kmer2hash(motif), 5, True)
        assert np.array_equal(h2, hits_plain) and np.array_equal(p2, pos_plain), f"index-mode planes differ from the plain ones (run {rep})"
    assert np.array_equal(hits, hits_plain) and np.array_equal(pos, pos_plain)
    ds.close()
    assert len(hits) == 50_000_000 and int(hits.sum(dtype=np.int64)) == len(pos)
    assert np.count_nonzero(hits) > 0.4 * len(hits)                                # radius 5 of 14: most reads have a nearest hit
    rng = np.random.default_rng(3)
    sample = np.concatenate([[0, len(borders) - 1], rng.integers(0, len(borders), size=12_000)])
    assert _check_scan(seq, borders, hits, pos, 14, int(kmer2hash(motif)), 5, sample) > 4000


def test_c5_counting_beyond_2_32_positions(c5_reads):
    """Counting with per-read dedupe on the C5 reads: 1.505e10 positions, i.e. position, group and skip-word indices beyond 2^32 /
    2^28 in the dedupe kernel's batches, the histogram passes and the partitioned k = 14 count (64-bit stream cursors).  The count
    tables are linear in the reads (dedupe works per read, the revcom merge is linear): whole == sum over three uneven read ranges,
    each small enough for 32-bit positions; closed-form total without dedupe."""
    from kmap_amd.kmer_count import DeviceCounts
    from kmap_amd.motif_discovery import DeviceSeq
    seq, borders = c5_reads
    n_reads = len(borders)
    dc = DeviceCounts()
    ds = DeviceSeq(seq, borders)
    full = {}
    for k in (8, 14):
        ds.count(dc, k, dedupe=True, merge_revcom=True)
        full[k] = _dense(*dc.fetch(), k)
    ds.count(dc, 8, dedupe=False, merge_revcom=False)
    assert int(dc.fetch()[1].sum(dtype=np.int64)) == n_reads * (300 - 8 + 1)        # no N in the synthetic reads
    ds.close()
    cuts = [0, 13_000_001, 31_234_567, n_reads]
    acc = {k: np.zeros(4 ** k, np.int64) for k in (8, 14)}
    for a, b in zip(cuts, cuts[1:]):
        s, bd = _slice_reads(seq, borders, a, b)
        part = DeviceSeq(s, bd)
        for k in (8, 14):
            part.count(dc, k, dedupe=True, merge_revcom=True)
            acc[k] += _dense(*dc.fetch(), k)
        part.close()
    for k in (8, 14):
        np.testing.assert_array_equal(acc[k], full[k])
    # the same table from eight key-space shards (positions and group indices beyond 2^32 in the staging pass; 459 000 output chunks)
    ds = DeviceSeq(seq, borders)
    bounds = [(4 ** 14 * r // 8) & ~7 for r in range(8)] + [4 ** 14]
    acc14 = np.zeros(4 ** 14, np.int64)
    for r in range(8):
        ds.count_range(dc, 14, True, True, bounds[r], bounds[r + 1] - bounds[r])
        su, sc = dc.fetch()
        assert len(su) == 0 or (len(np.unique(su)) == len(su))
        acc14 += _dense(su, sc, 14)
    ds.close()
    dc.close()
    np.testing.assert_array_equal(acc14, full[14])


def _pipeline_like_sample(n, seed):
    """n expanded 8-mers with the C3 hand-over's structure: label 0 around CCTACGTA (full length), label 1 around ATCGATA (a
    7-mer: its pairs are compared on 7 bases), noise label 2; sorted by label, with duplicates."""
    from kmap_amd.kmer_count import kmer2hash
    rng = np.random.default_rng(seed)
    parts, labs = [], []
    for lab, (core, m) in enumerate((("CCTACGTA", n // 3), ("ATCGATAC", n // 6))):
        base = int(kmer2hash(core))
        kh = np.full(m, base, np.uint64)
        for _ in range(2):                                                          # up to two substitutions
            pos, val = rng.integers(0, K, size=m), rng.integers(0, 4, size=m).astype(np.uint64)
            sh = (2 * pos).astype(np.uint64)
            kh = (kh & ~(np.uint64(3) << sh)) | (val << sh)
        parts.append(kh)
        labs.append(np.full(m, lab))
    rest = n - sum(len(p) for p in parts)
    parts.append(rng.integers(0, 4 ** K, size=rest, dtype=np.uint64))
    labs.append(np.full(rest, 2))
    return np.concatenate(parts).astype(np.uint32), np.concatenate(labs).astype(np.int32), [8, 7]


def test_c4_hamming_full_size():
    """BASELINE config C4's matrix on ONE GPU: N = 200 000 (4e10 pairs, 40 GB of uint8): sampled rows == oracle, symmetric."""
    from kmap_amd import _ffi
    from kmap_amd.hamdist import hamdist_matrix_dev, pitch_for
    from oracle import oracle as O
    n = 200_000
    kh, lab, lens = _pipeline_like_sample(n, 4)
    ld = pitch_for(n)
    kh_d, lab_d, D_d = _ffi.DeviceBuffer.from_numpy(kh), _ffi.DeviceBuffer.from_numpy(lab), _ffi.DeviceBuffer(n * ld)
    hamdist_matrix_dev(kh_d.ptr, lab_d.ptr, n, K, lens, D_d.ptr, ld)
    _ffi.sync()
    rng = np.random.default_rng(5)
    rows = np.unique(np.concatenate([[0, 1, n // 3 - 1, n // 3, n // 2 - 1, n // 2, n - 1], rng.integers(0, n, size=120)]))
    got = np.stack([D_d.to_numpy(np.uint8, (n,), offset=int(r) * ld) for r in rows])
    for b in (D_d, kh_d, lab_d):
        b.free()
    want = np.empty((1, n), np.uint8)
    kh64, cl = np.ascontiguousarray(kh, np.uint64), np.ascontiguousarray(lens, np.int32)
    for g, r in zip(got, rows):
        O.lib().ko_hamdist_rows(kh64, lab, n, K, cl, len(cl), int(r), 1, want)
        np.testing.assert_array_equal(g, want[0])
    sub = got[:, rows]                                                              # D[rows][:, rows]
    np.testing.assert_array_equal(sub, sub.T)
    assert np.all(np.diag(sub) == 0) and got.max() <= K


def _seq_form_rows(n, round_rows=16384):
    """>= 128 rows covering the three forms of the SEQ split (embed.hip seq_split: R whole rounds of 16 384 rows -> (R - 1) / 2 pair
    rounds of 2 x 16 384 rows first, the other whole rounds in the quad form, the remainder in the wide form), with the first / last
    row of each region, wave and block edges, and a random fill"""
    rounds = n // round_rows
    pair = ((rounds - 1) // 2) * 2 * round_rows if rounds >= 3 else 0
    main = rounds * round_rows
    rng = np.random.default_rng(11)
    rows = set()
    for lo, hi in ((0, pair), (pair, main), (main, n)):
        if hi > lo:
            edge = [lo, lo + 1, lo + 15, lo + 16, lo + 31, lo + 32, lo + 63, lo + 64, lo + 127, lo + 128, hi - 129, hi - 65, hi - 33, hi - 2, hi - 1]
            rows |= {r for r in edge if lo <= r < hi}
            rows |= set(rng.integers(lo, hi, 40).tolist())
    rows = np.array(sorted(rows), np.int64)
    assert len(rows) >= 128 and (pair == 0 or (rows < pair).sum() >= 40) and ((rows >= pair) & (rows < main)).sum() >= 40 and (rows >= main).sum() >= 30
    return rows


def _assert_seq_rows_equal_oracle(g, sums_d, lds, lut, coords, rows):
    """device SEQ gradient rows == the CPU restatement of the reference's row sum (taichi_core.py:305-326 with T from
    visualization.py:131-145: IEEE f32, j ascending, j != i, no FMA), BIT for bit.  The rows' probabilities are rebuilt on the
    host from the rows' u16 neighbour sums through the same LUT (len(rows) x N floats -- no N x N matrix)."""
    from oracle import baseline as B
    n = coords.shape[1]
    P = np.empty((len(rows), n), np.float32)
    for t, r in enumerate(rows):
        P[t] = lut[sums_d.to_numpy(np.uint16, (lds,), offset=int(r) * lds * 2)[:n]]
    want = B.embed_forces_rows(P, rows, coords, threads=8)
    got = np.ascontiguousarray(g[:, rows])
    bad = np.nonzero((got.view(np.uint32) != want.view(np.uint32)).any(axis=0))[0]
    assert len(bad) == 0, f"SEQ rows differ from the oracle: rows {rows[bad][:10].tolist()} got {got[:, bad[:3]]} want {want[:, bad[:3]]}"
    assert np.abs(want).max() > 0


def test_c3_embedding_force_evaluation_full_size():
    """N = 50 000 with the hand-over's label structure: the symmetric FAST kernel (the opt-in mode) against the SEQ kernel (the default: the
    reference's summation order) on one force evaluation -- loss to 2e-6, gradient to 2e-5 of its scale -- and 20 FAST
    iterations that keep lowering the loss."""
    from kmap_amd import _ffi, visualization as V
    from kmap_amd.hamdist import hamdist_matrix_dev, pitch_for
    n = 50_000
    kh, lab, lens = _pipeline_like_sample(n, 6)
    ldd = pitch_for(n)
    kh_d, lab_d, D_d = _ffi.DeviceBuffer.from_numpy(kh), _ffi.DeviceBuffer.from_numpy(lab), _ffi.DeviceBuffer(n * ldd)
    hamdist_matrix_dev(kh_d.ptr, lab_d.ptr, n, K, lens, D_d.ptr, ldd)
    nb_d = V.knn_select_dev(D_d.ptr, ldd, n, 20)
    sums_d, lds = V.knn_sums_kmers_dev(kh_d.ptr, lab_d.ptr, n, K, lens, nb_d, 20)
    # the profile-based sums == the matrix-based sums on sampled rows
    chk_d, _ = V.knn_sums_dev(D_d.ptr, ldd, nb_d, n, 20, row0=12_345, nrows=64)
    np.testing.assert_array_equal(sums_d.to_numpy(np.uint16, (64, lds), offset=12_345 * lds * 2)[:, :n], chk_d.to_numpy(np.uint16, (64, lds))[:, :n])
    for b in (chk_d, D_d, nb_d, kh_d, lab_d):
        b.free()
    lut = V.hd_prob_lut(K, 20, 400 * K)
    coords = np.random.default_rng(7).standard_normal((2, n)).astype(np.float32)
    outs = {}
    for tag, mode in (("seq", V.EMBED_SEQ), ("fast", V.EMBED_FAST)):
        sess = V.EmbedSession(n, 10, 0.01, mode)
        _ffi.check(_ffi.lib().kmap_embed_set_prob_lut(sess._h, sums_d.ptr, lds, _ffi.ptr(lut), len(lut)))
        sess.set_coords(coords)
        g_d, l_d = _ffi.DeviceBuffer(2 * n * 4), _ffi.DeviceBuffer(8)
        g_d.zero()
        sess.forces(g_d.ptr, l_d.ptr)
        _ffi.sync()
        outs[tag] = (g_d.to_numpy(np.float32, (2, n)), float(l_d.to_numpy(np.float64, (1,))[0]))
        if tag == "seq":
            # the SEQ kernel's large-N forms against the ORACLE's arithmetic, bit for bit, on sampled rows of every form: at
            # N = 50 000 rows [0, 32 768) run in the pair form, [32 768, 49 152) in the quad form, the last 848 in the wide form
            _assert_seq_rows_equal_oracle(outs[tag][0], sums_d, lds, lut, coords, _seq_form_rows(n))
            # the producer / adder form (embed_seq.hip: the form of a rank's share of a row-sharded run, where the forms above cannot fill
            # the machine): rank 5 of 8 (rows 31 250 .. 37 499: blocks of 25 rows, the last one short) and, forced, all 50 000 rows (the
            # 64-row-slot geometry) -- bit for bit the other forms' gradient, hence the oracle's arithmetic
            import os
            for row0, nrows, force in ((31_250, 6_250, None), (0, n, "adder")):
                if force:
                    os.environ["KMAP_SEQ_FORM"] = force
                try:
                    sh = V.EmbedSession(n, 1, 0.01, V.EMBED_SEQ, row0=row0, nrows=nrows)
                finally:
                    os.environ.pop("KMAP_SEQ_FORM", None)
                _ffi.check(_ffi.lib().kmap_embed_set_prob_lut(sh._h, sums_d.ptr + row0 * lds * 2, lds, _ffi.ptr(lut), len(lut)))
                sh.set_coords(coords)
                g2, l2 = _ffi.DeviceBuffer(2 * n * 4), _ffi.DeviceBuffer(8)
                g2.zero()
                sh.forces(g2.ptr, l2.ptr)
                _ffi.sync()
                part = g2.to_numpy(np.float32, (2, n))
                np.testing.assert_array_equal(part[:, row0:row0 + nrows].view(np.uint32), outs[tag][0][:, row0:row0 + nrows].view(np.uint32))
                assert not part[:, :row0].any() and not part[:, row0 + nrows:].any()
                if nrows == n:
                    assert abs(float(l2.to_numpy(np.float64, (1,))[0]) - outs[tag][1]) <= 1e-7 * abs(outs[tag][1])
                sh.close()
                g2.free()
                l2.free()
        if tag == "fast":
            sess.set_jitter(np.random.default_rng(8).normal(0, 0.01, 4096))
            sess.step(20)
            losses = sess.losses()
            assert len(losses) == 20 and np.all(np.isfinite(losses)) and losses[-1] < losses[0]
        sess.close()
    sums_d.free()
    (gs, ls), (gf, lf) = outs["seq"], outs["fast"]
    assert abs(lf - ls) <= 2e-6 * abs(ls)
    np.testing.assert_allclose(gf, gs, rtol=0, atol=2e-5 * np.abs(gs).max())


def test_c2_whole_run():
    """BASELINE config C2 -- the reference's DEFAULT size (default_config.toml:24-32): 100 k x 150 bp reads, k = 6..9, N = 5000
    sampled k-mers, 2500 iterations -- as one run of both verbs on a clean res_dir, in the package's default (SEQ) embedding
    mode: the planted motifs come out as the finals, low_dim_data.tsv has the contract's shape, the loss falls; the
    hand-over's first 10 iterations equal the oracle's restatement of the reference loop at this size (1e-5 abs)."""
    import pickle
    import shutil
    from pathlib import Path
    from kmap_amd import _ffi, synth, visualization as V
    from kmap_amd.e2e import run_e2e
    from kmap_amd.hamdist import hamdist_matrix_dev, pitch_for
    from oracle import oracle as O
    r = run_e2e("C2", "default", keep=True)
    res = Path(r["res_dir"])
    try:
        finals = r["final_conseq"]
        planted = [synth.MOTIF_A, synth.MOTIF_B]
        both = planted + [O.reverse_complement(m) for m in planted]
        # the two strongest finals are the planted motifs' cores (one each); the reference's greedy merge may add a weaker
        # flank-overlapping final (seed 1: CCAGGAC, the start of motif B with two flank bases)
        assert 2 <= len(finals) <= 3 and all(any(f in m for m in both) for f in finals[:2]), finals
        assert {next(i for i, m in enumerate(both) if f in m) % 2 for f in finals[:2]} == {0, 1}
        rows = (res / "low_dim_data.tsv").read_text().splitlines()
        assert rows[0] == "x\ty\tlabel" and len(rows) == 5001
        tab = np.array([ln.split("\t") for ln in rows[1:]], dtype=np.float64)
        assert tab.shape == (5000, 3) and np.isfinite(tab).all() and set(np.unique(tab[:, 2])) <= set(map(float, range(len(finals) + 1)))     # one label per final + the noise label
        with open(res / "sample_kmers.pkl", "rb") as fh:
            samp_kh, samp_cnts, samp_label, conseqs = pickle.load(fh)
        assert int(np.sum(samp_cnts)) == 5000 and list(conseqs) == finals
        with open(res / "sample_kmer_hamdist_mat.pkl", "rb") as fh:
            klen, D, labels = pickle.load(fh)
        assert klen == max(len(f) for f in finals) and D.shape == (5000, 5000) and D.dtype == np.int64     # dense contract at this N
        np.testing.assert_array_equal(labels, tab[:, 2].astype(np.int64))
        # the embedding the verb wrote == a traced re-run from the hand-over (same seed); the loss trace falls and stays finite
        nb = np.argpartition(D, 20, axis=1)[:, :20]
        tr = {}
        best = V.kmap(D, klen, n_max_iter=2500, random_seed=7, debug=False, neighbor_inds_mat=nb, trace=tr)
        np.testing.assert_allclose(np.round(best.astype(np.float64), 3).T, tab[:, :2], atol=1.1e-3)
        losses = tr["losses"]
        assert len(losses) == 2500 and np.isfinite(losses).all() and losses.min() < 0.5 * losses[0]
        run_best = np.minimum.accumulate(losses)
        assert run_best[-1] <= run_best[len(run_best) // 2] <= run_best[10] < losses[0]
        # first 10 iterations at the full C2 size against the oracle loop (IEEE f32, j-ascending sums), from the same probabilities
        n = 5000
        ldd = pitch_for(n)
        Dp = np.zeros((n, ldd), np.uint8)
        Dp[:, :n] = D
        D_d = _ffi.DeviceBuffer.from_numpy(Dp)
        sums_d, lds = V.knn_sums_dev(D_d.ptr, ldd, nb, n, 20)
        sums = sums_d.to_numpy(np.uint16, (n, lds))[:, :n]
        D_d.free()
        sums_d.free()
        p = V.hd_prob_lut(klen, 20, 400 * klen)[sums]
        np.fill_diagonal(p, 0.0)
        tr10 = {}
        V.kmap(D, klen, n_max_iter=10, random_seed=7, debug=False, neighbor_inds_mat=nb, mode=V.EMBED_SEQ, trace=tr10)
        np.random.seed(7)
        ld = np.random.randn(2, n).astype("float32")
        for _ in range(10):
            np.random.randn(2, n)
        want_losses = []
        for _ in range(10):
            q = O.cal_ld_prob_mat(ld)
            want_losses.append(O.cross_entropy(p, q))
            ld += (-O.gradient_loss(p, q, ld) * 0.01)
            ld = O.add_jitter(ld, eps=0.1)
        np.testing.assert_allclose(tr10["losses"], np.array(want_losses, np.float32), rtol=3e-6)
        np.testing.assert_allclose(tr10["last_coords"], ld, rtol=0, atol=1e-5)
    finally:
        shutil.rmtree(res, ignore_errors=True)


def test_c4_whole_run():
    """BASELINE config C4 as ONE run of both verbs on one GPU, clean res_dir, package default (SEQ): 10 M x 150 bp reads, k = 6..9,
    N = 200 000 sampled k-mers, 2500 iterations (reference motif_discovery.py:759-808 -> visualization.py:259-326; ~47 s).
      * the planted motifs' cores come out as the finals; the hand-over is the compact one (no 320 GB int64 matrix), 200 000 points;
      * low_dim_data.tsv has the contract's shape and is the lowest-loss snapshot printed with %3.3f;
      * the loss trace is finite, 2500 long (or ends by the reference's stop rule) and falls;
      * the FIRST FIVE ITERATIONS from the same hand-over, one at a time: on >= 128 sampled rows (every form of the SEQ split) the
        device's gradient equals the oracle's row sums (kb_embed_forces_rows: IEEE f32, j ascending, no FMA) BIT for bit, and the
        coordinates after the step equal y + (-(4 g) lr) bit for bit -- each iteration checked from the device's own previous
        coordinates, so the chain covers five iterations without an O(N^2) CPU pass."""
    import pickle
    import shutil
    from kmap_amd import _ffi, synth, visualization as V
    from kmap_amd.e2e import run_e2e
    from kmap_amd.hamdist import hamdist_matrix_dev, pitch_for
    from kmap_amd.kmer_count import get_hash_dtype
    from oracle import baseline as B, oracle as O
    n = 200_000
    V.TRACE_SINK = tr = {}
    try:
        r = run_e2e("C4", "default", keep=True)
    finally:
        V.TRACE_SINK = None
    res = Path(r["res_dir"])
    try:
        finals = r["final_conseq"]
        planted = [synth.MOTIF_A, synth.MOTIF_B]
        both = planted + [O.reverse_complement(m) for m in planted]
        assert 2 <= len(finals) <= 3 and all(any(f in m for m in both) for f in finals[:2]), finals
        assert {next(i for i, m in enumerate(both) if f in m) % 2 for f in finals[:2]} == {0, 1}
        with open(res / "sample_kmer_hamdist_mat.pkl", "rb") as fh:
            klen, D, labels = pickle.load(fh)
        assert D is None and klen == max(len(f) for f in finals) and len(labels) == n          # compact hand-over above DENSE_PKL_MAX_N
        with open(res / "sample_kmers.pkl", "rb") as fh:
            samp_kh, samp_cnts, samp_label, conseqs = pickle.load(fh)
        assert int(np.sum(samp_cnts)) == n and list(conseqs) == finals
        rows_txt = (res / "low_dim_data.tsv").read_text().splitlines()
        assert rows_txt[0] == "x\ty\tlabel" and len(rows_txt) == n + 1
        tab = np.array([ln.split("\t") for ln in rows_txt[1:]], dtype=np.float64)
        assert tab.shape == (n, 3) and np.isfinite(tab).all() and set(np.unique(tab[:, 2])) <= set(map(float, range(len(finals) + 1)))
        np.testing.assert_array_equal(tab[:, 2].astype(np.int64), np.asarray(labels, np.int64))
        np.testing.assert_allclose(np.round(tr["best"].astype(np.float64), 3).T, tab[:, :2], atol=1.1e-3)
        losses = np.asarray(tr["losses"])
        assert (len(losses) == 2500 or tr["state"]["stopped"]) and np.isfinite(losses).all() and losses.min() < 0.5 * losses[0]
        run_best = np.minimum.accumulate(losses)
        assert run_best[-1] <= run_best[len(run_best) // 2] <= run_best[10] < losses[0]
        assert r["times"]["e2e_s"] < 120.0                                   # a whole run, not a stage: ~47 s on one MI355X
        # ---- the first five iterations, from the hand-over, against the oracle's arithmetic on sampled rows
        kh = np.repeat(np.asarray(samp_kh), samp_cnts).astype(get_hash_dtype(klen))
        lab = np.repeat(np.asarray(samp_label), samp_cnts).astype(np.int32)
        lens = [len(c) for c in conseqs]
        ldd = pitch_for(n)
        kh_d, lab_d, D_d = _ffi.DeviceBuffer.from_numpy(kh), _ffi.DeviceBuffer.from_numpy(lab), _ffi.DeviceBuffer(n * ldd)
        hamdist_matrix_dev(kh_d.ptr, lab_d.ptr, n, klen, lens, D_d.ptr, ldd)
        assert V.knn_mode(n) == "device"
        nb_d = V.knn_select_dev(D_d.ptr, ldd, n, 20)
        D_d.free()
        sums_d, lds = V.knn_sums_kmers_dev(kh_d.ptr, lab_d.ptr, n, klen, lens, nb_d, 20, natural_diag=True)     # as the verb's SEQ path calls it
        for b in (nb_d, kh_d, lab_d):
            b.free()
        lut = V.hd_prob_lut(klen, 20, 400 * klen)
        rows = _seq_form_rows(n)
        rows = rows[rows >= 2]                                               # add_jitter (as written) only ever touches points 0 / 1
        P = np.empty((len(rows), n), np.float32)
        for t, rr in enumerate(rows):
            P[t] = lut[sums_d.to_numpy(np.uint16, (lds,), offset=int(rr) * lds * 2)[:n]]
        comp_d, rowmap_d, stored = V.dedupe_sums_rows(sums_d, n, lds, n=n)   # as the verb does: repeated rows stored once
        assert rowmap_d is not None and stored <= len(samp_kh)              # at most one stored row per distinct sampled k-mer (31 249 of 31 298; with the reference's zero diagonal: 98 177)
        ld0, ph = V._init_draws(n, 10, 7)
        sess = V.EmbedSession(n, 10, 0.01, V.EMBED_SEQ)
        try:
            sess.set_prob_lut(comp_d, lds, lut, rowmap_d, stored)
            sess.set_coords(ld0, ph)
            sess.set_jitter(np.random.normal(0, 0.01, 4096))                # the global stream right behind the initial draws, as _run_loop draws it
            g_d, l_d = _ffi.DeviceBuffer(2 * n * 4), _ffi.DeviceBuffer(8)
            lr = np.float32(0.01)
            for it in range(5):
                y0 = sess.coords()
                g_d.zero()
                sess.forces(g_d.ptr, l_d.ptr)
                _ffi.sync()
                g = g_d.to_numpy(np.float32, (2, n))
                want = B.embed_forces_rows(P, rows, y0, threads=8)
                got = np.ascontiguousarray(g[:, rows])
                assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"iteration {it}: SEQ gradient rows differ from the oracle"
                sess.step(1)
                y1 = sess.coords()
                upd = y0[:, rows] + (-(np.float32(4.0) * want) * lr)
                assert np.array_equal(y1[:, rows].view(np.uint32), upd.astype(np.float32).view(np.uint32)), f"iteration {it}: coordinates after the step"
            first5 = sess.losses()
            np.testing.assert_allclose(first5, losses[:5], rtol=1e-6)       # the verb's run started the same way (same seed, same hand-over)
            g_d.free()
            l_d.free()
        finally:
            sess.close()
    finally:
        shutil.rmtree(res, ignore_errors=True)


def test_c4_embedding_force_evaluation_full_size():
    """BASELINE config C4's embedding stage at N = 200 000 (80 GB of neighbour sums on one GPU): one force evaluation of the
    symmetric FAST kernel (4x the tiles of C3; 782 row blocks) against the SEQ kernel (the reference's summation order) -- loss
    to 2e-6, gradient to 4e-5 of its scale -- and the same evaluation sharded cyclically over 3 ranks (each rank's session run
    here in turn, their messages summed as the all-reduce sums them): the sharded gradient and loss equal the one-GPU FAST
    result to summation round-off, no block is lost or counted twice."""
    from kmap_amd import _ffi, visualization as V
    from kmap_amd.distributed import MSG_EXTRA, loss_from_limbs
    from kmap_amd.hamdist import hamdist_matrix_dev, pitch_for
    n = 200_000
    kh, lab, lens = _pipeline_like_sample(n, 8)
    ldd = pitch_for(n)
    kh_d, lab_d, D_d = _ffi.DeviceBuffer.from_numpy(kh), _ffi.DeviceBuffer.from_numpy(lab), _ffi.DeviceBuffer(n * ldd)
    hamdist_matrix_dev(kh_d.ptr, lab_d.ptr, n, K, lens, D_d.ptr, ldd)
    nb_d = V.knn_select_dev(D_d.ptr, ldd, n, 20)
    D_d.free()
    sums_d, lds = V.knn_sums_kmers_dev(kh_d.ptr, lab_d.ptr, n, K, lens, nb_d, 20)
    lut = V.hd_prob_lut(K, 20, 400 * K)
    coords = np.random.default_rng(9).standard_normal((2, n)).astype(np.float32)
    outs = {}
    for tag, mode in (("seq", V.EMBED_SEQ), ("fast", V.EMBED_FAST)):
        sess = V.EmbedSession(n, 10, 0.01, mode)
        _ffi.check(_ffi.lib().kmap_embed_set_prob_lut(sess._h, sums_d.ptr, lds, _ffi.ptr(lut), len(lut)))
        sess.set_coords(coords)
        g_d, l_d = _ffi.DeviceBuffer(2 * n * 4), _ffi.DeviceBuffer(8)
        g_d.zero()
        sess.forces(g_d.ptr, l_d.ptr)
        _ffi.sync()
        outs[tag] = (g_d.to_numpy(np.float32, (2, n)), float(l_d.to_numpy(np.float64, (1,))[0]))
        sess.close()
        g_d.free()
    (gs, ls), (gf, lf) = outs["seq"], outs["fast"]
    # SEQ at N = 200 000 (5 pair rounds, 2 quad rounds, 3392 wide rows) against the oracle's row sums, bit for bit, on sampled rows
    _assert_seq_rows_equal_oracle(gs, sums_d, lds, lut, coords, _seq_form_rows(n))
    assert abs(lf - ls) <= 2e-6 * abs(ls)
    # f32 row sums of 200 000 terms in two different orders: 15 of the 400 000 entries differ by 2.0 .. 2.4e-5 of the gradient's
    # scale (N = 50 000: all within 2e-5; the round-off of a sum grows with its length)
    np.testing.assert_allclose(gf, gs, rtol=0, atol=4e-5 * np.abs(gs).max())
    # three cyclic shards: rank r owns the 256-row blocks r, r + 3, ...; its probability rows are passed block after block
    world, msum = 3, np.zeros(2 * n + MSG_EXTRA, np.float32)
    n_blocks = 0
    for rank in range(world):
        blocks = V.cyclic_blocks(n, world, rank)
        n_blocks += len(blocks)
        blk_bytes = V.CYCLIC_BLOCK_ROWS * lds * 2
        part_d = _ffi.DeviceBuffer(len(blocks) * blk_bytes)
        for b, (r0, nr) in enumerate(blocks):
            _ffi.check(_ffi.lib().kmap_memcpy_d2d(part_d.ptr + b * blk_bytes, sums_d.ptr + r0 * lds * 2, nr * lds * 2, None))
        sess = V.EmbedSession(n, 10, 0.01, V.EMBED_FAST, cyclic=(world, rank))
        _ffi.check(_ffi.lib().kmap_embed_set_prob_lut(sess._h, part_d.ptr, lds, _ffi.ptr(lut), len(lut)))
        sess.set_coords(coords)
        m_d = _ffi.DeviceBuffer((2 * n + MSG_EXTRA) * 4)
        m_d.zero()
        sess.forces_msg(m_d.ptr)
        _ffi.sync()
        msum = msum + m_d.to_numpy(np.float32, (2 * n + MSG_EXTRA,))        # float32 sum, as the all-reduce computes it
        sess.close()
        for buf in (m_d, part_d):
            buf.free()
    assert n_blocks == (n + 255) // 256
    l3 = loss_from_limbs(msum[2 * n:])
    assert abs(l3 - lf) <= 1e-9 * abs(lf)                                       # the same pairs, f64 partials in another grouping
    np.testing.assert_allclose(msum[:2 * n].reshape(2, n), gf, rtol=0, atol=3e-6 * np.abs(gf).max())
    for b in (sums_d, nb_d, kh_d, lab_d):
        b.free()


def test_seq_pair_form_is_bit_identical(tmp_path):
    """N = 3 x 16 384 + 640: the SEQ split turns two of the three quad rounds into one round of pair-form waves (two sub-lanes per
    row) next to the remaining quad round and the wide left-over rows; the gradient must equal the run without the pair form
    (KMAP_SEQ_PAIR=0) bit for bit.  One force evaluation per process (the switch is read once); the 5-GB sums matrix is
    generated in the child from a seed."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    n = 3 * 16384 + 640
    code = ("import numpy as np, sys; sys.path.insert(0, sys.argv[1]); from kmap_amd import _ffi, visualization as V;"
            "n = int(sys.argv[2]); lds = (n + 127) & ~127; rng = np.random.default_rng(5);"
            "sd = _ffi.DeviceBuffer(n * lds * 2);\n"
            "for r0 in range(0, n, 4096):\n"
            "    m = min(4096, n - r0); blk = rng.integers(0, 3201, size=(m, lds), dtype=np.uint16);"
            " _ffi.check(_ffi.lib().kmap_memcpy_h2d(sd.ptr + r0 * lds * 2, _ffi.ptr(blk), m * lds * 2, None)); _ffi.sync()\n"
            "lut = V.hd_prob_lut(8, 20, 3200); ld = (rng.standard_normal((2, n)) * 5).astype(np.float32);"
            "s = V.EmbedSession(n, 10, 0.01, V.EMBED_SEQ);"
            "_ffi.check(_ffi.lib().kmap_embed_set_prob_lut(s._h, sd.ptr, lds, _ffi.ptr(lut), len(lut))); s.set_coords(ld);"
            "g = _ffi.DeviceBuffer(2 * n * 4); l = _ffi.DeviceBuffer(8); g.zero(); s.forces(g.ptr, l.ptr); _ffi.sync();"
            "rows = np.array([int(t) for t in sys.argv[4].split(',')]);"
            "sr = np.stack([sd.to_numpy(np.uint16, (lds,), offset=int(r) * lds * 2)[:n] for r in rows]);"
            "np.savez(sys.argv[3], g=g.to_numpy(np.float32, (2, n)), l=l.to_numpy(np.float64, (1,)), sr=sr, lut=lut, ld=ld)")
    root = str(Path(__file__).resolve().parent.parent)
    outs = {}
    rows = _seq_form_rows(n)
    for tag, env in (("pair", {}), ("quad", {"KMAP_SEQ_PAIR": "0"})):
        r = subprocess.run([sys.executable, "-c", code, root, str(n), str(tmp_path / f"{tag}.npz"), ",".join(str(t) for t in rows)],
                           env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[tag] = np.load(tmp_path / f"{tag}.npz")
    np.testing.assert_array_equal(outs["pair"]["g"].view(np.uint32), outs["quad"]["g"].view(np.uint32))
    assert outs["pair"]["g"].any()
    # ... and both equal the oracle's row sums on sampled rows of the pair, quad and wide regions (not only each other)
    from oracle import baseline as B
    o = outs["pair"]
    want = B.embed_forces_rows(o["lut"][o["sr"]], rows, o["ld"], threads=8)
    np.testing.assert_array_equal(np.ascontiguousarray(o["g"][:, rows]).view(np.uint32), want.view(np.uint32))
    lp, lq = float(outs["pair"]["l"][0]), float(outs["quad"]["l"][0])
    assert abs(lp - lq) <= 1e-8 * abs(lq)        # the loss is not bit-pinned: f32 partial sums over batches of different width
