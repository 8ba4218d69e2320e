"""GPU parity of the 2-bit-packed read path (packed.hip) against the CPU oracle: pack/unpack round trip, window
hashes for every k, histogram counting, Hamming-ball masking (incl. the reference's invalid-hash quirk) and the
occurrence scan.  Bit-exact."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def synth(rng, n_reads, lo, hi, p_n=0.02):
    lens = rng.integers(lo, hi + 1, size=n_reads)
    parts, st, borders = [], 0, []
    for L in lens:
        r = rng.integers(0, 4, size=L).astype(np.uint8)
        r[rng.random(L) < p_n] = 255
        parts += [r, np.array([255], np.uint8)]
        borders.append((st, st + L))
        st += L + 1
    return np.concatenate(parts), np.array(borders, dtype=np.int64)


@pytest.fixture(scope="module")
def env():
    from kmap_amd import _ffi
    from kmap_amd.kmer_count import DeviceCounts
    from kmap_amd.motif_discovery import DeviceSeq
    from oracle import oracle as O
    return _ffi, DeviceCounts, DeviceSeq, O


@pytest.mark.parametrize("n", [0, 1, 15, 16, 17, 1000, 4099])
def test_pack_unpack_roundtrip(env, n):
    _ffi, _, DeviceSeq, _ = env
    rng = np.random.default_rng(n)
    seq = rng.integers(0, 4, size=n).astype(np.uint8)
    seq[rng.random(n) < 0.1] = 255
    ds = DeviceSeq(seq, np.zeros((0, 2), np.int64))
    np.testing.assert_array_equal(ds.download(), seq)
    ds.close()


@pytest.mark.parametrize("k", [1, 2, 7, 8, 15, 16, 17, 24, 31])
def test_packed_hash_every_position(env, k):
    _ffi, _, DeviceSeq, O = env
    rng = np.random.default_rng(100 + k)
    seq, borders = synth(rng, 200, 1, 120)
    ds = DeviceSeq(seq, borders)
    dt = np.uint32 if k < 16 else np.uint64
    out_d = _ffi.DeviceBuffer(len(seq) * np.dtype(dt).itemsize)
    _ffi.check(_ffi.lib().kmap_hash_kmers_packed_dev(ds.codes.ptr, ds.inval_orig.ptr, len(seq), k, out_d.ptr, None))
    np.testing.assert_array_equal(out_d.to_numpy(dt, (len(seq),)), O.comp_kmer_hash(seq, k))
    ds.close()


def test_packed_counts_all_paths(env):
    _ffi, DeviceCounts, DeviceSeq, O = env
    rng = np.random.default_rng(7)
    seq, borders = synth(rng, 12000, 20, 160)          # ~1.1 M positions: LDS-pass histogram (k <= 9) and atomics (k >= 10)
    ds, dc = DeviceSeq(seq, borders), DeviceCounts()
    for k in (4, 8, 9, 10, 13, 16, 19):
        for dedupe in (False, True):
            ds.count(dc, k, dedupe=dedupe, merge_revcom=True)
            u, c = dc.fetch()
            ou, oc = O.count_kmers(seq, borders, k, rep_mode=not dedupe, revcom_mode=True)
            np.testing.assert_array_equal(u, ou)
            np.testing.assert_array_equal(c, oc)
    dc.close()
    ds.close()


def test_packed_mask_vs_oracle_incl_quirks(env):
    _ffi, _, DeviceSeq, O = env
    rng = np.random.default_rng(11)
    seq, borders = synth(rng, 1500, 25, 150)
    # k <= 16: the bit-sliced window test (bitslice.hip, one template instance per k); above: the per-window kernels
    for k, r in ((1, 0), (2, 0), (3, 1), (4, 0), (5, 1), (6, 1), (7, 2), (8, 2), (9, 2), (10, 3), (11, 3), (12, 4), (13, 4), (14, 5), (15, 5),
                 (16, 6), (17, 7), (31, 12)):
        cons = rng.integers(0, 4 ** k, size=3, dtype=np.uint64)
        cons[0] = int(O.kmer2hash("T" * k))            # poly-T matches invalid (all-ones) hashes: masks across separators
        rad = np.array([r, r, 0])
        ds = DeviceSeq(seq, borders)
        ds.mask(k, cons, rad)
        want = O.mask_input(seq.copy(), k, cons, rad)
        np.testing.assert_array_equal(ds.download(), want)
        # a second round masks the already masked array (hashes of the current state), like successive find_motif trials
        cons2 = rng.integers(0, 4 ** k, size=2, dtype=np.uint64)
        ds.mask(k, cons2, np.array([r, r]))
        np.testing.assert_array_equal(ds.download(), O.mask_input(want.copy(), k, cons2, np.array([r, r])))
        ds.reset()
        np.testing.assert_array_equal(ds.download(), seq)
        ds.close()
    # golden quirk case
    a = np.concatenate([O.dna2arr("ACGTACGTAC"), O.dna2arr("GGGGGGGGGG")])
    ds = DeviceSeq(a, np.array([[0, 10], [11, 21]]))
    ds.mask(4, np.array([O.kmer2hash("TTTT")]), np.array([0]))
    assert O.arr2dna(ds.download()) == "ACGTACGNNNNNNNGGGGNNNN"
    ds.close()


def test_packed_mask_negative_radius_matches_nothing(env):
    """max_ham_dist < 0: `ham_dist <= r` (kmer_count.py:594-603) holds for no window, so nothing is masked -- on the bit-sliced path
    (k <= 16, where "count > r" is built for r >= 0) and on the per-window path (k > 16) alike; mixed with a real entry only that
    entry masks."""
    _ffi, _, DeviceSeq, O = env
    rng = np.random.default_rng(17)
    seq, borders = synth(rng, 400, 25, 120)
    for k in (6, 8, 14, 18):
        cons = rng.integers(0, 4 ** k, size=3, dtype=np.uint64)
        cons[1] = int(O.kmer2hash("T" * k))            # would match every invalid window at any radius >= 0
        ds = DeviceSeq(seq, borders)
        ds.mask(k, cons, np.array([-1, -1, -3]))
        np.testing.assert_array_equal(ds.download(), seq)
        rad = np.array([-1, 0, 1 if k < 16 else 4])
        ds.mask(k, cons, rad)
        np.testing.assert_array_equal(ds.download(), O.mask_input(seq.copy(), k, cons, rad))
        ds.close()


def test_packed_scan_tiny_inputs(env):
    """Inputs shorter than one row of code words of the one-pass scan kernel (25 groups = 400 positions): the two-pass form runs;
    and inputs just above that size, whose reads all sit in the zone where the kernel moves its code-word row back from the end of
    the arrays.  Hits, positions == oracle."""
    _ffi, _, DeviceSeq, O = env
    rng = np.random.default_rng(77)
    for total, k, r in ((40, 4, 1), (200, 6, 2), (330, 8, 3), (420, 8, 3), (700, 12, 5), (1500, 14, 6)):
        parts, borders, st = [], [], 0
        while st < total:
            L = int(rng.integers(1, 90))
            parts += [rng.integers(0, 4, size=L).astype(np.uint8), np.array([255], np.uint8)]
            borders.append((st, st + L))
            st += L + 1
        seq, borders = np.concatenate(parts), np.array(borders, np.int64)
        ds = DeviceSeq(seq, borders)
        a, b = borders[len(borders) // 2]
        src = seq[a:b]
        cons = int(O.comp_kmer_hash(src, k)[0]) if b - a >= k else 5          # the k-mer a middle read starts with: at least one exact hit
        for cc in (cons, 0):
            for revcom in (True, False):
                hits, pos = ds.scan(k, cc, r, revcom)
                buf, md, off = np.empty(4096, np.int32), C.c_int(0), 0
                for i, (x, y) in enumerate(borders):
                    m = O.lib().ko_scan_read(np.ascontiguousarray(seq[x:y]), y - x, k, cc, r, int(revcom), buf, C.byref(md))
                    assert hits[i] == m, (total, k, i, hits[i], m)
                    np.testing.assert_array_equal(pos[off:off + m], buf[:m])
                    off += m
                assert off == len(pos)
        ds.close()


def test_packed_mask_many_consensuses(env):
    _ffi, _, DeviceSeq, O = env
    rng = np.random.default_rng(13)
    seq, borders = synth(rng, 300, 30, 100)
    cons = rng.integers(0, 4 ** 7, size=70, dtype=np.uint64)      # > 32: batched flag passes on the entry state
    rad = rng.integers(0, 2, size=70)
    ds = DeviceSeq(seq, borders)
    ds.mask(7, cons, rad)
    np.testing.assert_array_equal(ds.download(), O.mask_input(seq.copy(), 7, cons, rad))
    ds.close()


@pytest.mark.parametrize("k,r", [(3, 0), (5, 1), (6, 1), (7, 7), (8, 2), (9, 3), (10, 0), (11, 4), (12, 3), (13, 5), (14, 5), (15, 6), (16, 14),
                                 (16, 20), (17, 3), (20, 8), (31, 10), (31, 15)])
def test_packed_scan_vs_oracle(env, k, r):
    """k <= 16: bit-sliced hit bits + exact evaluation of the hits per read (reads above 1024 positions by a whole wave); above:
    the flat nibble path (radius <= 14) and the wave-per-read kernel (radius 15); incl. the negative-slice quirk for reads
    shorter than k-1, radii up to and beyond k, and borders off the separators"""
    _ffi, _, DeviceSeq, O = env
    rng = np.random.default_rng(k)
    lens = [0, 1, k - 2, k - 1, k, k + 1, 63, 64, 65, 256, 257, 300, 700, 1023 + k, 1024 + k, 1025 + k, 3000, 5003] \
        + list(rng.integers(5, 200, size=300))
    parts, borders, st = [], [], 0
    for L in lens:
        rd = rng.integers(0, 4 if L % 3 else 1, size=L).astype(np.uint8)   # some poly-A reads: many ties at the minimum
        if L > 20 and L % 5 == 0:
            rd[L // 2] = 255
        parts += [rd, np.array([255], np.uint8)]
        borders.append((st, st + L))
        st += L + 1
    seq, borders = np.concatenate(parts), np.array(borders, np.int64)
    sub = borders.copy()                                      # a second set of "reads": windows strictly inside the first
    wide = (borders[:, 1] - borders[:, 0]) > 12
    sub[wide, 0] += 3
    sub[wide, 1] -= 2
    borders = np.concatenate([borders, sub[wide]])
    ds = DeviceSeq(seq, borders)
    for cons in (0, int(O.kmer2hash("T" * k)), int(rng.integers(0, 4 ** k, dtype=np.uint64))):
        for revcom in (True, False):
            hits, pos = ds.scan(k, cons, r, revcom)
            buf, md, off = np.empty(8192, np.int32), C.c_int(0), 0
            for i, (a, b) in enumerate(borders):
                m = O.lib().ko_scan_read(np.ascontiguousarray(seq[a:b]), b - a, k, cons, r, int(revcom), buf, C.byref(md))
                assert hits[i] == m, (i, b - a, hits[i], m)
                np.testing.assert_array_equal(pos[off:off + m], buf[:m])
                off += m
            assert off == len(pos)
            lazy = ds.scan_lazy(k, cons, r, revcom)            # device-resident variant: summary without a fetch, same lists
            assert (lazy.n_reads_hit, lazy.total, lazy.max_hits) == (int(np.count_nonzero(hits)), len(pos), int(hits.max(initial=0)))
            h2, p2 = lazy
            np.testing.assert_array_equal(h2, hits)
            np.testing.assert_array_equal(p2, pos)
    ds.close()


def test_packed_scan_more_hits_than_the_one_pass_buffer(env):
    """The one-pass per-read form writes the positions into a temporary buffer of max(2 n_seq, 2^20) entries cut into 64 regions; a
    scan with far more hits (a 3-mer at radius 1: most windows of most reads) must notice the overflow and fall back to the
    count / scan / write form -- same lists as the oracle's, read by read."""
    _ffi, _, DeviceSeq, O = env
    rng = np.random.default_rng(99)
    n_reads, L = 30_000, 150
    seq = rng.integers(0, 4, size=n_reads * (L + 1)).astype(np.uint8)
    seq = seq.reshape(n_reads, L + 1)
    seq[::2, :] = 0                                           # every other read poly-A: all its windows tie at the minimum (only the
    seq[1::4, :] = 3                                          #  hits at a read's minimum are kept), a quarter poly-T
    seq[:, L] = 255
    seq = np.ascontiguousarray(seq.reshape(-1))
    borders = np.stack([np.arange(n_reads) * (L + 1), np.arange(n_reads) * (L + 1) + L], axis=1).astype(np.int64)
    ds = DeviceSeq(seq, borders)
    for k, cons, r in ((3, int(O.kmer2hash("AAA")), 1), (4, int(O.kmer2hash("TTTT")), 2)):
        hits, pos = ds.scan(k, cons, r, True)
        assert len(pos) > max(2 * n_reads, 1 << 20)           # beyond the buffer: the fallback ran
        offs = np.concatenate([[0], np.cumsum(hits, dtype=np.int64)])
        assert offs[-1] == len(pos)
        buf, md = np.empty(L, np.int32), C.c_int(0)
        for i in list(range(0, n_reads, 997)) + [n_reads - 1]:
            a, b = borders[i]
            m = O.lib().ko_scan_read(np.ascontiguousarray(seq[a:b]), b - a, k, cons, r, 1, buf, C.byref(md))
            assert hits[i] == m
            np.testing.assert_array_equal(pos[offs[i]:offs[i + 1]], buf[:m])
    ds.close()


def test_counting_full_size_properties(env):
    """3 M x 150 bp reads (4.5e8 positions; the C3 shape at 30 %): size-independent properties instead of a CPU recount:
    sum of counts == number of valid windows, the revcom merge conserves the total up to palindrome doubling, per-read
    dedupe never increases a count, and masking with radius 0 removes exactly the k-mers it targets."""
    _ffi, DeviceCounts, DeviceSeq, O = env
    from kmap_amd import synth
    seq, borders = synth.synth_reads(3_000_000, 150, 5)
    seq[::1009] = 255
    ds, dc = DeviceSeq(seq, borders), DeviceCounts()
    inv = (seq == 255).astype(np.int64)
    csum = np.concatenate([[0], np.cumsum(inv)])
    for k in (8, 10, 12):
        n_valid = int(np.count_nonzero(csum[k:] - csum[:-k] == 0))                   # windows without a 255
        ds.count(dc, k, dedupe=False, merge_revcom=False)
        u, c = dc.fetch()
        assert int(c.sum()) == n_valid == dc.total()
        assert np.all(np.diff(u.astype(np.int64)) > 0) and u.max() < 4 ** k          # ascending unique keys
        ds.count(dc, k, dedupe=False, merge_revcom=True)
        mu, mc = dc.fetch()
        rc = O.get_revcom_hash_arr(u, k)
        pal = int(c[rc == u].sum())                                                  # palindromes are doubled by the reference
        assert int(mc.sum()) == n_valid + pal
        assert len(mu) == len(np.unique(np.minimum(u, rc.astype(u.dtype))))
        ds.count(dc, k, dedupe=True, merge_revcom=False)
        du, dcnt = dc.fetch()
        np.testing.assert_array_equal(du, u)                                         # dedupe removes occurrences, not k-mers
        assert np.all(dcnt <= c) and np.all(dcnt >= 1)
        # mask the most frequent k-mer with radius 0: its count drops to 0, nothing new appears
        top = u[np.argmax(c)]
        ds.mask(k, np.array([top]), np.array([0]))
        ds.count(dc, k, dedupe=False, merge_revcom=False)
        u2, c2 = dc.fetch()
        assert top not in set(u2[:0]) and not np.any(u2 == top)
        assert np.all(np.isin(u2, u))
        ds.reset()
    dc.close()
    ds.close()


@pytest.mark.parametrize("k", [15, 16])
def test_two_level_partitioned_counts_vs_oracle(env, k):
    """k = 15 / 16 above 2^20 positions: bucket partition + second level (32 / 128 sub-buckets of 32768 bins per bucket, keys
    re-sorted tile by tile inside their bucket) == the oracle, with and without per-read dedupe and revcom merge.  The reads
    hold long poly-T / poly-A stretches: at k = 16 the all-T 16-mer's hash 0xFFFFFFFF is the uint32 invalid marker and is
    counted aside; repeats and N bases exercise dedupe and invalid windows."""
    _ffi, DeviceCounts, DeviceSeq, O = env
    rng = np.random.default_rng(1000 + k)
    seq, borders = synth(rng, 9000, 100, 220, p_n=0.002)
    assert len(seq) > (1 << 20)
    for r in range(0, 9000, 37):                       # poly-T / poly-A / tandem-repeat reads
        st, en = borders[r]
        seq[st:en] = (3, 0, 3)[r % 3] if r % 3 != 2 else seq[st:en]
        if r % 3 == 2:
            seq[st:en] = np.resize(np.array([0, 1, 2, 3, 3, 3], np.uint8), en - st)
    ds, dc = DeviceSeq(seq, borders), DeviceCounts()
    for dedupe in (False, True):
        for merge in (True, False):
            ds.count(dc, k, dedupe=dedupe, merge_revcom=merge)
            u, c = dc.fetch()
            ou, oc = O.count_kmers(seq, borders, k, rep_mode=not dedupe, revcom_mode=merge)
            np.testing.assert_array_equal(u, ou)
            np.testing.assert_array_equal(c, oc)
            assert u.dtype == ou.dtype and c.dtype == oc.dtype
    if k == 16:
        ds.count(dc, k, dedupe=False, merge_revcom=False)
        u, c = dc.fetch()
        assert u[-1] == 0xFFFFFFFF and c[-1] > 1000        # the all-T 16-mer is present and counted
    dc.close()
    ds.close()


@pytest.mark.parametrize("k", [6, 8, 9, 10, 12, 13, 14])
def test_lds_histogram_hot_bins(env, k):
    """LDS-pass histogram under extreme skew: 12 M positions, 85 % of them poly-A (plus poly-T and random reads), so one bin takes
    > 8 M of the counts and every wave's lanes hit the same LDS word; windows of other passes' bin ranges and invalid windows go
    to the lanes' private bins behind the table.  k >= 10 take the partitioned histogram (counts_fine.hip): the poly-A bucket holds
    ~85 % of the keys and is cut into ~130 slices that add their bins to the table with device atomics; at k = 13 / 14 (two 16-bit
    counters per LDS word) the poly-A bin also spills its 16 384-count chunks to the global list.  Counts must equal the oracle's,
    with and without per-read dedupe."""
    _hot_bins_case(k, *env)


def _hot_bins_case(k, _ffi=None, DeviceCounts=None, DeviceSeq=None, O=None):
    if _ffi is None:
        from kmap_amd import _ffi
        from kmap_amd.kmer_count import DeviceCounts
        from kmap_amd.motif_discovery import DeviceSeq
        from oracle import oracle as O
    rng = np.random.default_rng(500 + k)
    seq, borders = synth(rng, 60_000, 190, 210, p_n=0.001)
    for r in range(len(borders)):
        st, en = borders[r]
        if r % 20 < 17:
            seq[st:en] = 0                                      # poly-A
        elif r % 20 == 17:
            seq[st:en] = 3                                      # poly-T
    ds, dc = DeviceSeq(seq, borders), DeviceCounts()
    for dedupe in (False, True):
        ds.count(dc, k, dedupe=dedupe, merge_revcom=False)
        u, c = dc.fetch()
        ou, oc = O.count_kmers(seq, borders, k, rep_mode=not dedupe, revcom_mode=False)
        np.testing.assert_array_equal(u, ou)
        np.testing.assert_array_equal(c, oc)
        if not dedupe:
            assert c.max() > 8_000_000
    dc.close()
    ds.close()


@pytest.mark.parametrize("k", [8, 9])
def test_hist_counter_width_switch(k):
    """KMAP_HIST16=0: the k = 8 / 9 histogram with 32-bit LDS counters (32 768 bins per pass, two / eight passes) instead of the default
    16-bit ones (65 536 bins per pass + the sweep that keeps the halves from carrying): the same counts under the same skew (the
    switch is read once per process: a child runs the case)."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = str(Path(__file__).resolve().parent.parent)
    code = f"import sys; sys.path.insert(0, sys.argv[1]); import tests.test_gpu_packed as T; T._hot_bins_case({k}); print('case ok')"
    r = subprocess.run([sys.executable, "-c", code, root], env=dict(os.environ, KMAP_HIST16="0"), cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "case ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.parametrize("base,k", [(0, 8), (1, 8), (2, 9)])
def test_hist16_single_kmer_no_carry(env, base, k):
    """ONE k-mer takes every window of every block (homopolymer reads, 64 M positions = 16 rounds of the 256 x 1024-thread
    grid, five sweeps): a 16-bit LDS half must never carry into its neighbour (even bins: the low half, carry -> the next k-mer's
    count; odd bins: the high half, carry lost).  The count is known in closed form: sum over the reads of (len - k + 1)."""
    _ffi, DeviceCounts, DeviceSeq, O = env
    rng = np.random.default_rng(77 + base)
    n_reads = 200_000
    lens = rng.integers(250, 391, n_reads)
    starts = np.concatenate([[0], np.cumsum(lens + 1)[:-1]])
    seq = np.full(int((lens + 1).sum()), base, np.uint8)
    seq[starts + lens] = 255
    borders = np.stack([starts, starts + lens], axis=1).astype(np.int64)
    assert len(seq) >= 256 * 1024 * 16 * 15
    ds, dc = DeviceSeq(seq, borders), DeviceCounts()
    ds.count(dc, k, dedupe=False, merge_revcom=False)
    u, c = dc.fetch()
    kmer = int(sum(base << (2 * i) for i in range(k)))
    np.testing.assert_array_equal(u, [kmer])
    np.testing.assert_array_equal(c, [int((lens - k + 1).sum())])
    dc.close()
    ds.close()


def _uniform_scan_case():
    import hashlib
    from kmap_amd.kmer_count import kmer2hash
    from kmap_amd.motif_discovery import DeviceSeq
    rng = np.random.default_rng(5150)
    h = hashlib.sha256()
    for L, nr in ((150, 3000), (31, 2000), (300, 1500)):
        seq, borders = synth(rng, nr, L, L, p_n=0.003)
        ds = DeviceSeq(seq, borders)
        for k, cons, rad in ((14, "AGGACCTACGTACA", 5), (8, "CCTACGTA", 2), (6, "TTTTTT", 1)):
            hits, pos = ds.scan(k, kmer2hash(cons), rad, True)
            h.update(hits.tobytes())
            h.update(pos.tobytes())
        ds.close()
    return h.hexdigest()


def test_scan_uniform_layout_declaration(env):
    """kmap_scan_declare_uniform (round 6): for fixed-length reads the per-read pass of the scan derives the borders from the read index
    instead of loading them.  DeviceSeq declares it for its own handles after a host check; the library verifies the declaration on the
    device.  Same hits and positions as an undeclared handle and as the oracle; a wrong declaration and ragged reads are refused;
    KMAP_SCAN_UNIFORM=0 switches the path off (child process, same digest)."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    _ffi, _, DeviceSeq, O = env
    from kmap_amd.kmer_count import kmer2hash
    lib = _ffi.lib()
    rng = np.random.default_rng(9)
    seq, borders = synth(rng, 2500, 120, 120, p_n=0.004)
    ds = DeviceSeq(seq, borders)
    assert ds._uniform_layout() == (120, 121)
    cons = int(kmer2hash("CCTACGTA"))
    hits, pos = ds.scan(8, cons, 2, True)                            # the DeviceSeq's own handle: declared
    raw = _ffi.vp()
    _ffi.check(lib.kmap_scan_create(C.byref(raw)))
    ok = _ffi.i32(7)
    _ffi.check(lib.kmap_scan_declare_uniform(raw.value, ds.borders.ptr, ds.n_seq, 121, 121, C.byref(ok), None))   # wrong length
    assert ok.value == 0
    _ffi.check(lib.kmap_scan_declare_uniform(raw.value, ds.borders.ptr, ds.n_seq, 120, 122, C.byref(ok), None))   # wrong stride
    assert ok.value == 0
    tot = _ffi.i64(0)
    _ffi.check(lib.kmap_scan_run_packed_dev(raw.value, ds.codes.ptr, ds.inval_orig.ptr, ds.n, ds.borders.ptr, ds.n_seq, 8, cons, 2, 1,
                                            C.byref(tot), ds.planes.ptr, None))                                   # undeclared: borders loaded
    h2, p2 = np.empty(ds.n_seq, np.int32), np.empty(tot.value, np.int32)
    _ffi.check(lib.kmap_scan_fetch(raw.value, _ffi.ptr(h2), None, _ffi.ptr(p2)))
    np.testing.assert_array_equal(h2, hits)
    np.testing.assert_array_equal(p2, pos)
    assert ds.declare_layout(raw.value)                                                                            # now declared: same again
    _ffi.check(lib.kmap_scan_run_packed_dev(raw.value, ds.codes.ptr, ds.inval_orig.ptr, ds.n, ds.borders.ptr, ds.n_seq, 8, cons, 2, 1,
                                            C.byref(tot), ds.planes.ptr, None))
    _ffi.check(lib.kmap_scan_fetch(raw.value, _ffi.ptr(h2), None, _ffi.ptr(p2)))
    np.testing.assert_array_equal(h2, hits)
    np.testing.assert_array_equal(p2, pos)
    buf, md, off = np.empty(4096, np.int32), C.c_int(0), 0
    for i, (a, b) in enumerate(borders):
        m = O.lib().ko_scan_read(np.ascontiguousarray(seq[a:b]), b - a, 8, cons, 2, 1, buf, C.byref(md))
        assert hits[i] == m
        np.testing.assert_array_equal(pos[off:off + m], buf[:m])
        off += m
    lib.kmap_scan_destroy(raw.value)
    ds.close()
    rag_seq, rag_borders = synth(rng, 500, 50, 90)
    rag = DeviceSeq(rag_seq, rag_borders)
    assert rag._uniform_layout() is None
    rag.scan(8, cons, 2, True)
    rag.close()
    root = str(Path(__file__).resolve().parent.parent)
    here = _uniform_scan_case()
    code = "import sys; sys.path.insert(0, sys.argv[1]); import tests.test_gpu_packed as T; print('digest', T._uniform_scan_case())"
    r = subprocess.run([sys.executable, "-c", code, root], env=dict(os.environ, KMAP_SCAN_UNIFORM="0"), cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert f"digest {here}" in r.stdout


def _key_range_reads(rng, n_reads=9000):
    """~1.4 M positions (>= 2^20: the partitioned passes run) with what the range rule must get right: planted motifs (heavy buckets),
    N's, poly-A / poly-T reads (k = 16: the all-T 16-mer's hash is the invalid marker, its partner all-A is position 0), reads made of
    ACGT repeats (even-k palindromes: their own partner) and reads followed by their reverse complement (pairs inside and across ranges)"""
    seq, borders = synth(rng, n_reads, 120, 190, p_n=0.002)
    comp = np.array([3, 2, 1, 0], np.uint8)
    for r in range(len(borders)):
        st, en = borders[r]
        m = r % 50
        if m == 0:
            seq[st:en] = 0
        elif m == 1:
            seq[st:en] = 3
        elif m == 2:
            seq[st:en] = np.resize(np.array([0, 1, 2, 3], np.uint8), en - st)
        elif m == 3 and r > 0:
            pst, pen = borders[r - 1]
            L = min(en - st, pen - pst)
            prev = seq[pst:pst + L]
            ok = prev != 255
            rc = np.where(ok, comp[np.where(ok, prev, 0)], 255)[::-1]
            seq[st:st + L] = rc
        elif m < 24:
            p0 = st + int(rng.integers(0, en - st - 20))
            seq[p0:p0 + 20] = np.array([0, 2, 2, 0, 1, 1, 3, 0, 1, 2, 3, 0, 1, 0, 2, 2, 3, 1, 1, 0], np.uint8)
    return seq, borders


@pytest.mark.parametrize("k", [11, 13, 14, 15, 16])
def test_key_range_counting_equals_slices_of_the_table(env, k):
    """kmap_counts_run_packed_range_dev (key-space-sharded counting; reference kmer_count.py:476-491,643-685): for G = 2, 3,
    8 ranks the shards [4^k r / G, 4^k (r + 1) / G) -- each computed from the windows that decide it alone, in a virtual table of a
    smaller k -- concatenated in rank order ARE the single-GPU table, key for key and count for count, with and without the
    reverse-complement merge and the per-read dedupe; k = 14 also against the oracle.  Masked reads (the working mask) included."""
    _ffi, DeviceCounts, DeviceSeq, O = env
    rng = np.random.default_rng(1400 + k)
    seq, borders = _key_range_reads(rng)
    assert len(seq) >= 1 << 20
    ds, dc, ds_small = DeviceSeq(seq, borders), DeviceCounts(), None
    n_bins = 4 ** k
    try:
        for merge, dedupe in ((True, False), (True, True), (False, False), (False, True)):
            ds.count(dc, k, dedupe=dedupe, merge_revcom=merge)
            u, c = dc.fetch()
            if k == 14 and merge:
                ou, oc = O.count_kmers(seq, borders, k, rep_mode=not dedupe, revcom_mode=True)
                np.testing.assert_array_equal(u, ou)
                np.testing.assert_array_equal(c, oc)
            for G in (2, 3, 8):
                bounds = [(n_bins * r // G) & ~7 for r in range(G)] + [n_bins]
                us, cs = [], []
                for r in range(G):
                    ds.count_range(dc, k, dedupe, merge, bounds[r], bounds[r + 1] - bounds[r])
                    su, sc = dc.fetch()
                    us.append(su)
                    cs.append(sc)
                np.testing.assert_array_equal(np.concatenate(us), u, err_msg=f"k={k} merge={merge} dedupe={dedupe} G={G}: keys")
                np.testing.assert_array_equal(np.concatenate(cs), c, err_msg=f"k={k} merge={merge} dedupe={dedupe} G={G}: counts")
        # after masking (find_motif's later rounds count the working mask), and an odd range that is no multiple of the rank rule
        cons = np.array([int(O.kmer2hash("AGGACCTACGTACAGG"[:k])) if k <= 16 else 0], np.uint64)
        ds.mask(k, cons, np.array([2], np.int32))
        ds.count(dc, k, dedupe=False, merge_revcom=True)
        u, c = dc.fetch()
        cuts = [0, 8 * 1237, (n_bins // 3) & ~7, (n_bins // 3 + 4096) & ~7, n_bins]
        us, cs = [], []
        for a, b in zip(cuts[:-1], cuts[1:]):
            ds.count_range(dc, k, False, True, a, b - a)
            su, sc = dc.fetch()
            us.append(su)
            cs.append(sc)
        np.testing.assert_array_equal(np.concatenate(us), u)
        np.testing.assert_array_equal(np.concatenate(cs), c)
        ds.reset()
        # small inputs (below the partitioned passes' threshold) take the whole-table path and must give the same slices
        sseq, sborders = synth(rng, 300, 60, 100)
        ds_small = DeviceSeq(sseq, sborders)
        ds_small.count(dc, k, dedupe=True, merge_revcom=True)
        u, c = dc.fetch()
        us, cs = [], []
        for r in range(3):
            a, b = (n_bins * r // 3) & ~7, ((n_bins * (r + 1) // 3) & ~7) if r < 2 else n_bins
            ds_small.count_range(dc, k, True, True, a, b - a)
            su, sc = dc.fetch()
            us.append(su)
            cs.append(sc)
        np.testing.assert_array_equal(np.concatenate(us), u)
        np.testing.assert_array_equal(np.concatenate(cs), c)
    finally:
        dc.close()
        ds.close()
        if ds_small is not None:
            ds_small.close()


def test_key_range_counting_random_ranges(env):
    """40 random (k, range, merge, dedupe) cases on the reads of the case above: kmap_counts_run_packed_range_dev == the matching slice
    -- by position -- of the one-GPU table.  Positions are not stored in a merged table (an unpaired higher member sits at its own position
    with its key replaced by rc(x), kmer_count.py:643-685), so the expected slice comes from the UNMERGED table and a numpy restatement of
    the merge that keeps the positions; ranges from 8 bins up to the whole table, ends at the table's end included."""
    _ffi, DeviceCounts, DeviceSeq, O = env
    rng = np.random.default_rng(4711)
    seq, borders = _key_range_reads(rng)
    ds, dc = DeviceSeq(seq, borders), DeviceCounts()
    try:
        for case in range(40):
            k = int(rng.choice([11, 12, 13, 14, 15, 16]))
            merge, dedupe = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
            n_bins = 4 ** k
            span = int(min(n_bins, 8 * 4 ** int(rng.integers(0, k - 1))))
            first = int(rng.integers(0, (n_bins - span) // 8 + 1)) * 8
            if case % 7 == 0:
                first = n_bins - span                                       # a range that ends at the table's end
            if k == 16 and merge and span > 2 ** 31 - 8:
                span = 2 ** 31 - 8                                          # beyond: the whole-table path (covered by the G = 2 case above)
            # the table WITHOUT the merge holds every position's own count; the reference's merge (kmer_count.py:643-685) restated on
            # it with the positions kept: the higher member of a present pair is deleted, a palindrome doubles, an unpaired higher member
            # stays at ITS position with its key replaced by the reverse complement
            ds.count(dc, k, dedupe=dedupe, merge_revcom=False)
            u0, c0 = dc.fetch()
            p0 = u0.astype(np.uint64)
            if merge:
                t = np.uint64(4 ** k - 1) - p0
                r0 = np.zeros_like(p0)
                for _ in range(k):
                    r0 = (r0 << np.uint64(2)) | (t & np.uint64(3))
                    t = t >> np.uint64(2)
                idx = np.minimum(np.searchsorted(p0, r0), len(p0) - 1)
                present = p0[idx] == r0
                cr = np.where(present, c0[idx], 0).astype(c0.dtype)
                keep = ~(present & (p0 > r0))
                pos = p0[keep]
                u = np.minimum(p0, r0)[keep].astype(u0.dtype)
                c = (c0 + cr)[keep]                                        # a palindrome is its own partner: c0 + c0
                ds.count(dc, k, dedupe=dedupe, merge_revcom=True)           # ... which is the device's merged table
                du, dcnt = dc.fetch()
                np.testing.assert_array_equal(du, u)
                np.testing.assert_array_equal(dcnt, c)
            else:
                pos, u, c = p0, u0, c0
            sel = (pos >= np.uint64(first)) & (pos < np.uint64(first + span))
            ds.count_range(dc, k, dedupe, merge, first, span)
            su, sc = dc.fetch()
            np.testing.assert_array_equal(su, u[sel], err_msg=f"case {case}: k={k} merge={merge} dedupe={dedupe} range=[{first}, +{span})")
            np.testing.assert_array_equal(sc, c[sel], err_msg=f"case {case}: counts")
    finally:
        dc.close()
        ds.close()


def test_key_range_counting_argument_errors(env):
    _ffi, DeviceCounts, DeviceSeq, O = env
    rng = np.random.default_rng(5)
    seq, borders = synth(rng, 50, 60, 100)
    ds, dc = DeviceSeq(seq, borders), DeviceCounts()
    try:
        for k, first, nb in ((10, 0, 4 ** 10), (17, 0, 64), (12, 4, 64), (12, 0, 0), (12, 4 ** 12 - 8, 16), (12, 4 ** 12, 8)):
            with pytest.raises(ValueError):
                ds.count_range(dc, k, False, True, first, nb)
    finally:
        dc.close()
        ds.close()


@pytest.mark.parametrize("max_len", [512, 513])
def test_dedupe_long_reads_both_paths(env, max_len):
    """Per-read dedupe at the length limit of the LDS-bitmap kernel: reads up to 512 positions (9 steps of 64 windows, claim masks
    in LDS, whole-bitmap clears) and, with one read of 513, the materialised-hash-array path for the whole input.  Tandem repeats
    and poly-A stretches make most windows duplicates; exact (k <= 8) and hashed (k >= 9) bitmaps."""
    _ffi, DeviceCounts, DeviceSeq, O = env
    rng = np.random.default_rng(900 + max_len)
    seq, borders = synth(rng, 3000, 300, 512, p_n=0.003)
    st, en = borders[7]
    seq[st:en] = np.resize(np.array([0, 1, 2, 3, 0, 0, 1], np.uint8), en - st)      # tandem repeat
    st, en = borders[11]
    seq[st:en] = 0
    if max_len > 512:                                                               # one read beyond the limit
        extra = rng.integers(0, 4, size=max_len).astype(np.uint8)
        seq = np.concatenate([seq, extra, np.array([255], np.uint8)])
        borders = np.concatenate([borders, np.array([[len(seq) - max_len - 1, len(seq) - 1]], np.int64)])
    assert (borders[:, 1] - borders[:, 0]).max() == max_len or max_len == 512
    ds, dc = DeviceSeq(seq, borders), DeviceCounts()
    for k in (6, 8, 9, 13, 16):
        ds.count(dc, k, dedupe=True, merge_revcom=False)
        u, c = dc.fetch()
        ou, oc = O.count_kmers(seq, borders, k, rep_mode=False, revcom_mode=False)
        np.testing.assert_array_equal(u, ou)
        np.testing.assert_array_equal(c, oc)
    dc.close()
    ds.close()


@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 6, 7, 8, 9, 11, 16])
def test_dedupe_fast_and_general_forms(env, k):
    """The dedupe kernel's two forms side by side: ragged reads of 1 .. 230 positions (0 .. 4 steps of 64 windows: up to 3 steps run
    the unclamped fast form, 4 the general one; reads of one window have no step at all), every start offset inside a skip word,
    reads WITHOUT a separator between them (a window may run into the next read: it still belongs to the read it starts in), the
    last reads right at the end of the arrays (their unclamped loads would pass the padding: general form), low-complexity reads
    (most windows duplicates) and N bases.  k = 1, 2 take the hashed bitmap (a k-mer needs 5 bits to pick its bit in the exact
    one), 3 .. 8 the exact one, above that the hashed one with its exact confirmation.  Counts == oracle."""
    _ffi, DeviceCounts, DeviceSeq, O = env
    rng = np.random.default_rng(4100 + k)
    lens = rng.integers(1, 231, size=6000)
    lens[-40:] = rng.integers(1, 60, size=40)                    # short reads at the very end: several frames inside the tail zone
    parts, borders, pos = [], [], 0
    for r, ln in enumerate(lens):
        kind = r % 7
        if kind == 0:
            body = np.full(ln, rng.integers(0, 4), np.uint8)                                     # homopolymer
        elif kind == 1:
            body = np.resize(rng.integers(0, 4, size=rng.integers(2, 6)).astype(np.uint8), ln)   # tandem repeat
        else:
            body = rng.integers(0, 4, size=ln).astype(np.uint8)
            if kind == 2:
                body[rng.random(ln) < 0.03] = 255
        parts.append(body)
        borders.append((pos, pos + ln))
        pos += ln
        if r % 3:                                                # two reads in three end in a separator, the third touches the next read
            parts.append(np.array([255], np.uint8))
            pos += 1
    seq = np.concatenate(parts)
    borders = np.array(borders, np.int64)
    ds, dc = DeviceSeq(seq, borders), DeviceCounts()
    ds.count(dc, k, dedupe=True, merge_revcom=False)
    u, c = dc.fetch()
    ou, oc = O.count_kmers(seq, borders, k, rep_mode=False, revcom_mode=False)
    np.testing.assert_array_equal(u, ou)
    np.testing.assert_array_equal(c, oc)
    dc.close()
    ds.close()


def test_dedupe_tiny_arrays(env):
    """Counting with per-read dedupe on arrays around the size of one read frame of the dedupe kernel's unclamped loads (13 groups =
    208 positions): shorter arrays take the hash-array path, the first longer ones run the kernel with every read in its tail zone."""
    _ffi, DeviceCounts, DeviceSeq, O = env
    rng = np.random.default_rng(8)
    for total in (3, 17, 40, 150, 175, 176, 177, 191, 192, 193, 207, 208, 209, 230, 400):
        parts, borders, pos = [], [], 0
        while pos < total:
            ln = int(min(total - pos, rng.integers(1, 70)))
            parts.append(np.resize(rng.integers(0, 4, size=rng.integers(1, 5)).astype(np.uint8), ln))   # short tandem repeats: duplicates
            borders.append((pos, pos + ln))
            pos += ln
            if pos < total:
                parts.append(np.array([255], np.uint8))
                pos += 1
        seq, bd = np.concatenate(parts), np.array(borders, np.int64)
        ds, dc = DeviceSeq(seq, bd), DeviceCounts()
        for k in (2, 4, 8, 9):
            ds.count(dc, k, dedupe=True, merge_revcom=False)
            u, c = dc.fetch()
            ou, oc = O.count_kmers(seq, bd, k, rep_mode=False, revcom_mode=False)
            np.testing.assert_array_equal(u, ou, err_msg=f"n={len(seq)} k={k}")
            np.testing.assert_array_equal(c, oc, err_msg=f"n={len(seq)} k={k}")
        dc.close()
        ds.close()


def test_bitsliced_scan_and_mask_fuzz_vs_oracle(env):
    """Seeded fuzz of the bit-sliced window test (k <= 16) against the oracle: random k, radius (0 .. beyond k), strand flag,
    consensus (random, poly-A, poly-T = the all-ones hash of invalid windows), N rate, read lengths 0 .. 1500 (short-read path,
    whole-wave path for reads above 1024 positions, negative-slice quirk), array lengths off every 32-window boundary."""
    _ffi, _, DeviceSeq, O = env
    rng = np.random.default_rng(2024)
    buf, md = np.empty(4096, np.int32), C.c_int(0)
    for case in range(36):
        k = int(rng.integers(1, 17))
        r = int(rng.integers(0, k + 3))
        revcom = bool(rng.integers(0, 2))
        kind = case % 4
        cons = 0 if kind == 1 else (4 ** k - 1 if kind == 2 else int(rng.integers(0, 4 ** k, dtype=np.uint64)))
        lens = list(rng.integers(0, 90, size=60)) + list(rng.integers(100, 400, size=25)) + [int(rng.integers(1025, 1500)), k - 1 if k > 1 else 0, k, 33, 64]
        rng.shuffle(lens)
        p_n = float(rng.choice([0.0, 0.01, 0.1]))
        parts, borders, st = [], [], 0
        for L in lens:
            L = int(L)
            rd = rng.integers(0, 4 if rng.random() < 0.8 else 1, size=L).astype(np.uint8)
            rd[rng.random(L) < p_n] = 255
            parts += [rd, np.array([255], np.uint8)]
            borders.append((st, st + L))
            st += L + 1
        seq, borders = np.concatenate(parts), np.array(borders, np.int64)
        ds = DeviceSeq(seq, borders)
        hits, pos = ds.scan(k, cons, r, revcom)
        off = 0
        for i, (a, b) in enumerate(borders):
            m = O.lib().ko_scan_read(np.ascontiguousarray(seq[a:b]), b - a, k, cons, r, int(revcom), buf, C.byref(md))
            assert hits[i] == m, (case, k, r, revcom, cons, i, int(b - a), int(hits[i]), int(m))
            np.testing.assert_array_equal(pos[off:off + m], buf[:m])
            off += m
        assert off == len(pos)
        # masking with the same consensus (+ a random second one) on the same reads
        cons2 = np.array([cons, int(rng.integers(0, 4 ** k, dtype=np.uint64))], np.uint64)
        rad2 = np.array([min(r, k), int(rng.integers(0, max(1, k // 2)))])
        ds.mask(k, cons2, rad2)
        np.testing.assert_array_equal(ds.download(), O.mask_input(seq.copy(), k, cons2, rad2), err_msg=f"case {case}: k={k} r={rad2}")
        ds.close()


@pytest.mark.parametrize("case", ["all_invalid", "one_kmer", "two_hot_buckets", "ragged_tail"])
def test_fine_partition_edge_inputs(env, case):
    """counts_fine.hip (k = 10 .. 14 above 2^20 positions) on inputs that stress its bookkeeping: no valid window at all; one k-mer
    only (every key in bucket 0: the bucket is cut into ~2000 slices, the bin spills its 16-bit counter ~70 times); two planted
    k-mers in otherwise random reads (two heavy buckets among 4094 ordinary ones); a length that ends inside a tile and inside a
    16-position group.  Counts == oracle, with and without dedupe / revcom merge."""
    _ffi, DeviceCounts, DeviceSeq, O = env
    rng = np.random.default_rng(77)
    n_reads, L = 7000, 160
    seq, borders = synth(rng, n_reads, L, L, p_n=0.0)
    if case == "all_invalid":
        seq[:] = 255
    elif case == "one_kmer":
        for st, en in borders:
            seq[st:en] = 0
    elif case == "two_hot_buckets":
        for r in range(0, n_reads, 2):
            st, en = borders[r]
            seq[st:en] = np.resize(np.array([1, 2, 3, 0, 2, 1, 1, 3, 0, 0, 2, 3, 1, 2], np.uint8), en - st) if r % 4 == 0 else \
                np.resize(np.array([3, 3, 0, 1, 2, 0, 3, 1], np.uint8), en - st)
    else:
        seq, borders = seq[:borders[-1][1] - 37], borders.copy()
        borders[-1][1] = len(seq)
    assert len(seq) > (1 << 20)
    ds, dc = DeviceSeq(seq, borders), DeviceCounts()
    for k in (10, 12, 13, 14):
        for dedupe, merge in ((False, True), (True, False)):
            ds.count(dc, k, dedupe=dedupe, merge_revcom=merge)
            u, c = dc.fetch()
            ou, oc = O.count_kmers(seq, borders, k, rep_mode=not dedupe, revcom_mode=merge)
            np.testing.assert_array_equal(u, ou)
            np.testing.assert_array_equal(c, oc)
    dc.close()
    ds.close()


@pytest.mark.parametrize("k", [1, 2, 5, 8, 9, 13, 14, 16])
def test_hit_planes_index_mode_equals_plain_at_block_edges(env, k, monkeypatch):
    """The hit planes come from a kernel whose thread holds two 32-window words 256 words apart (a block = 16 384 positions) and picks
    its operand registers through the VGPR index mode; KMAP_SCAN_PLANES=plain selects the one-word formulation without it.  Array
    lengths around the block size (second word missing / partly there / just there), both output forms (scan: hit words, mask:
    16-bit halves), several table entries, every asm block size (k - 1 = 8 + 4 + 2 + 1 planes): the two agree and equal the oracle."""
    _ffi, _, DeviceSeq, O = env
    rng = np.random.default_rng(900 + k)
    for total in (16384 - 33, 16384 - 1, 16384, 16384 + 1, 16384 + 31, 2 * 16384 + 8192 - 7, 2 * 16384 + 8192 + 5, 3 * 16384 + 37):
        L = 173
        n_reads = total // (L + 1)
        lens = [L] * n_reads
        lens[-1] += total - n_reads * (L + 1)                               # the array (reads + separators) has exactly `total` positions
        parts, borders, st = [], [], 0
        for ln in lens:
            rd = rng.integers(0, 4, size=ln).astype(np.uint8)
            rd[rng.random(ln) < 0.01] = 255
            parts += [rd, np.array([255], np.uint8)]
            borders.append((st, st + ln))
            st += ln + 1
        seq, borders = np.concatenate(parts), np.array(borders, np.int64)
        assert len(seq) == total
        cons = int(rng.integers(0, 4 ** k, dtype=np.uint64))
        r = max(0, k // 3)
        cons3 = np.array([cons, int(rng.integers(0, 4 ** k, dtype=np.uint64)), 4 ** k - 1], np.uint64)
        rad3 = np.array([r, max(0, k // 4), 0])
        out = {}
        for mode in ("idx", "plain"):
            monkeypatch.setenv("KMAP_SCAN_PLANES", mode)
            ds = DeviceSeq(seq, borders)
            hits, pos = ds.scan(k, cons, r, True)
            ds.mask(k, cons3, rad3)
            out[mode] = (hits, pos, ds.download())
            ds.close()
        for a, b in zip(out["idx"], out["plain"]):
            np.testing.assert_array_equal(a, b, err_msg=f"k={k} total={total}")
        np.testing.assert_array_equal(out["idx"][2], O.mask_input(seq.copy(), k, cons3, rad3))
        buf, md, off = np.empty(4096, np.int32), C.c_int(0), 0
        for i, (a, b) in enumerate(borders):
            m = O.lib().ko_scan_read(np.ascontiguousarray(seq[a:b]), b - a, k, cons, r, 1, buf, C.byref(md))
            assert out["idx"][0][i] == m
            np.testing.assert_array_equal(out["idx"][1][off:off + m], buf[:m])
            off += m


@pytest.mark.parametrize("k", [11, 14, 16])
def test_key_range_stage_masked_form_equals_plain_form(env, k, monkeypatch):
    """Ranges that are a block of keys with a common prefix of 1 .. 8 bits (equal splits over 2, 4 ... 256 ranks) take the staging pass's
    masked form: "in range" for the 16 windows of a group at once from the first bits of the window / the complemented last bits of
    its reverse complement, only the kept windows hashed.  KMAP_RANGE_STAGE=plain runs the per-window form on the same range: the two
    shards are the same, for first / middle / last ranks, with and without merge and dedupe, on masked reads too; a range that is a
    power of two long but not aligned, and a prefix of 9 bits, take the per-window form in both runs."""
    _ffi, DeviceCounts, DeviceSeq, O = env
    rng = np.random.default_rng(7700 + k)
    seq, borders = _key_range_reads(rng)
    ds, dc = DeviceSeq(seq, borders), DeviceCounts()
    n_bins = 4 ** k
    try:
        for round_ in range(2):
            for G in (2, 4, 16, 256, 512):
                ln = n_bins // G
                for r in sorted({0, G // 2 - 1, G // 3, G - 1}):
                    for merge, dedupe in ((True, False), (False, False), (True, True)):
                        got = {}
                        for form in ("masked", "plain"):
                            if form == "plain":
                                monkeypatch.setenv("KMAP_RANGE_STAGE", "plain")
                            else:
                                monkeypatch.delenv("KMAP_RANGE_STAGE", raising=False)
                            ds.count_range(dc, k, dedupe, merge, r * ln, ln)
                            got[form] = dc.fetch()
                        np.testing.assert_array_equal(got["masked"][0], got["plain"][0], err_msg=f"k={k} G={G} r={r} merge={merge} dedupe={dedupe}")
                        np.testing.assert_array_equal(got["masked"][1], got["plain"][1], err_msg=f"k={k} G={G} r={r} merge={merge} dedupe={dedupe}")
            ln = n_bins // 8                                                 # a power of two long, not aligned
            monkeypatch.delenv("KMAP_RANGE_STAGE", raising=False)
            ds.count_range(dc, k, False, True, ln // 2, ln)
            a = dc.fetch()
            monkeypatch.setenv("KMAP_RANGE_STAGE", "plain")
            ds.count_range(dc, k, False, True, ln // 2, ln)
            b = dc.fetch()
            np.testing.assert_array_equal(a[0], b[0])
            np.testing.assert_array_equal(a[1], b[1])
            if round_ == 0:                                                  # second round: on the working mask
                ds.mask(k, np.array([int(O.kmer2hash("AGGACCTACGTACAGG"[:k]))], np.uint64), np.array([2], np.int32))
    finally:
        dc.close()
        ds.close()
