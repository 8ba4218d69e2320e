"""The device-side workload generator (csrc/synth.hip, used by bench.py's C5 leg): array contract, determinism, planted-motif
statistics -- and that a DeviceSeq built from it scans like the oracle on the very bytes it generated."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_synth_reads_dev_contract_and_scan():
    from kmap_amd import synth
    from kmap_amd.kmer_count import kmer2hash
    from oracle import oracle as O
    n_reads, L = 20_011, 97                                     # odd sizes: the last 16-byte store is ragged
    motif = "AGGACCTACGTACA"
    ds, raw = synth.synth_reads_dev(n_reads, L, 3, motifs=(motif, "AATCGATAGC"), keep_raw=True)
    n = n_reads * (L + 1)
    arr = raw(0, n)
    assert ds.n == n and ds.n_seq == n_reads
    view = arr.reshape(n_reads, L + 1)
    assert np.all(view[:, L] == 255) and view[:, :L].max() <= 3
    borders = ds.borders.to_numpy(np.int64, (n_reads, 2))
    np.testing.assert_array_equal(borders[:, 0], np.arange(n_reads) * (L + 1))
    np.testing.assert_array_equal(borders[:, 1] - borders[:, 0], L)
    counts = np.bincount(view[int(0.8 * n_reads):, :L].ravel(), minlength=4) / (0.2 * n_reads * L)
    assert np.all(np.abs(counts - 0.25) < 0.01)                 # the unplanted fifth is uniform
    # the planted classes: exact-motif reads at about 0.95^len of their class, none to speak of elsewhere
    codes = O.dna2arr(motif, False)
    win = np.lib.stride_tricks.sliding_window_view(view[:, :L], len(motif), axis=1)
    has = (win == codes[None, None, :]).all(axis=2).any(axis=1)
    cls0 = has[:int(0.4 * n_reads)].mean()
    # a "substituted" base keeps the read's random base, which equals the motif's one time in four (the host generator does the same)
    assert abs(cls0 - (0.95 + 0.05 / 4) ** len(motif)) < 0.03 and has[int(0.8 * n_reads):].mean() < 0.001
    # determinism and seed dependence
    ds2, raw2 = synth.synth_reads_dev(n_reads, L, 3, motifs=(motif, "AATCGATAGC"), keep_raw=True)
    np.testing.assert_array_equal(raw2(0, n), arr)
    ds3, raw3 = synth.synth_reads_dev(n_reads, L, 4, motifs=(motif, "AATCGATAGC"), keep_raw=True)
    assert (raw3(0, n) != arr).mean() > 0.5
    # the packed reads are these bytes: download == generated array; scan == oracle on every 37th read
    np.testing.assert_array_equal(ds.download(), arr)
    hits, pos = ds.scan(14, kmer2hash(motif), 5, True)
    offs = np.concatenate([[0], np.cumsum(hits, dtype=np.int64)])
    import ctypes
    buf, md = np.empty(L, np.int32), ctypes.c_int(0)
    for r in range(0, n_reads, 37):
        m = O.lib().ko_scan_read(np.ascontiguousarray(view[r, :L]), L, 14, int(kmer2hash(motif)), 5, 1, buf, md)
        assert m == hits[r]
        np.testing.assert_array_equal(pos[offs[r]:offs[r + 1]], buf[:m])
    assert hits[:int(0.4 * n_reads)].astype(bool).mean() > 0.95
    for d, r_ in ((ds, raw), (ds2, raw2), (ds3, raw3)):
        d.close()
        r_.free()
