#!/usr/bin/env python3
"""bench.py -- headline benchmark of the kmap hot path on MI355X.

Metric (BASELINE.json): Hamming pairs/s of the sampled-k-mer all-pairs matrix at N = 50 k, k = 8 (config C3), plus the
end-to-end scan_motif + visualize_kmers wall time.  One "step" = one pass of the Hamming-matrix kernel over the whole batch of
sampled k-mers: N x N ordered pairs written as uint8 into HBM; hashes/labels are resident in HBM before the timed region.
value = pairs processed by all ranks / max-over-ranks wall time.

The sample is what the C3 pipeline hands over: bench.py runs `scan_motif` on the C3 synthetic reads first (outside the timed
region) and takes sample_kmers.pkl -- expanded by counts, labels 0 / 1 / noise, finals CCTACGTA (8) and ATCGATA (7), so the
rows of the 7-mer's label take the kernel's prefix-compare branch -- not an idealised all-full-length sample.

Multi-GPU (--gpus G, launched by torch.distributed.run, one rank per GPU): the matrix is sharded by row blocks, every rank
holds all N hashes, no data-path collective.  `value` is the weak-scaling Hamming stage: N_total = 50 000 * sqrt(G)
(rounded to 16) so that every GPU keeps 2.5e9 pairs per step.  The same line carries the STRONG-scaling legs: `embed_dist` (the
C3 embedding, N = 50 000 fixed, sharded loop with its ONE all-reduce per iteration: ms per iteration and, from device events,
forces / collective / apply ms), `c4` (BASELINE config C4, N = 200 000 fixed: Hamming rows + the sharded embedding iteration) and,
for G > 1, `count_dist` (one k = 15 count pass over read shards: all-reduce of the 4-GiB table vs bins owned by key range).
At G = 1 `embed_dist` runs the sharded loop on a one-rank RCCL group next to the resident loop: `overhead_ms_per_iter` is what
the multi-GPU plumbing (message kernels, collective launch, Python) costs before any link is involved.  `embed_dist.seq` is the same
for the SEQ loop (the reference's summation order: the package default).  For G > 1 the line also carries `reads_dist` (the C3 reads
sharded: count k = 8 with dedupe, count k = 14, scan k = 8 r = 2 -- ms per pass incl. the collective, max over ranks, per-GPU
roofline fraction) and `e2e` (BOTH VERBS on a clean C3 res_dir under the process group, SEQ default and FAST: north_star's metric).
`--shard-proxy G` (N = 1): every rank's share of a G-GPU run timed on the one GPU, shard after shard (`shard_proxy`).

Extra objects on the JSON line:
  roofline      HIP-event kernel time of the headline kernel (mean / min / median per launch) vs the 8 TB/s HBM peak, algorithmic
                bytes = rows*N + 5N; `frac_of_achievable` = the same against a plain device fill of the same bytes measured in
                this run on this box; `stages`: the same for every other stage of the C3 path with SURVEY 8(d)'s bytes
  cpu_baseline  the compiled OpenMP port (oracle/kmap_cpu_baseline.c + the oracle's Hamming rows) timed on this box's host
                cores, all cores AND one core, CPU model stated: Hamming rows, find_motif k = 6..9 on >= 1 M reads, >= 10 embedding
                iterations at N = 50 000 (a reported baseline, not the target)
  e2e           scan_motif + visualize_kmers wall time: C3 k = 6..9 in both embedding modes, C3 with the reference's default
                k = 6..16, and C2 (100 k reads, N = 5 k, 2500 iterations, the reference's default size)
  c5            BASELINE config C5 at full size: k = 14, radius 5 Hamming-ball scan over 50 M x 300 bp reads generated in HBM
"""
import argparse
import json
import math
import os
import pickle
import statistics
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ceiling ~6290 GB/s
K = 8
N_BASE = 50_000
N_C4 = 200_000


# ---- the sample the C3 pipeline hands over ------------------------------------------------------------------------------
def pipeline_sample(res_dir, n_total):
    """Expanded (hashes uint32, labels int32, consensus lengths) of `n_total` sampled k-mers drawn by sample_disp_kmer from the
    kept C3 result directory (n_total = the run's own n_total_sample re-uses its sample_kmers.pkl)."""
    from kmap_amd._toml import load_toml
    from kmap_amd.kmer_count import gen_motif_def_dict
    from kmap_amd.motif_discovery import sample_disp_kmer
    res = Path(res_dir)
    cfg = load_toml(res / "config.toml")
    if n_total == cfg["motif_discovery"]["n_total_sample"]:
        with open(res / "sample_kmers.pkl", "rb") as fh:
            kh, cnts, lab, conseqs = pickle.load(fh)
    else:
        finals = (res / "final_conseq.txt").read_text().split()
        klen = max(len(c) for c in finals)
        np.random.seed(123)
        kh, cnts, lab, conseqs = sample_disp_kmer(finals, klen, gen_motif_def_dict(cfg), res / "kmer_count", n_total_sample=n_total,
                                                  n_motif_kmer=n_total // 2, revcom_mode=cfg["kmer_count"]["revcom_mode"])
    klen = max(len(c) for c in conseqs)
    assert klen == K, f"the C3 pipeline's longest final consensus is expected to be an {K}-mer, got {conseqs}"
    return (np.repeat(np.asarray(kh), cnts).astype(np.uint32), np.repeat(np.asarray(lab), cnts).astype(np.int32),
            [len(c) for c in conseqs], list(conseqs))


RAMP_MS = 60.0          # sustained GPU work issued right before a timed region (see ramp())


def ramp(fn, min_ms=RAMP_MS, max_calls=2000):
    """MI355X drops its clocks within milliseconds of going idle and needs ~20-40 ms of sustained work to bring them back
    (tools/hamdist_trend.py: the first ~40 back-to-back launches of the headline kernel after an idle gap take 0.50 ms, the
    following ones 0.395 ms, on the same box, same data).  A timed region of a few launches right after set-up therefore
    measures the ramp, not the kernel.  This queues at least `min_ms` of the SAME work, untimed, immediately in front of the
    timed region (no host synchronisation in between), so that the timed launches run at the clocks a real job -- thousands
    of launches -- runs at.  Returns the number of extra launches."""
    from kmap_amd import _ffi
    e0, e1 = _ffi.Event(), _ffi.Event()
    e0.record()
    fn()
    e1.record()
    _ffi.sync()
    one = max(e0.elapsed_ms(e1), 1e-3)
    calls = int(min(max_calls, max(1, math.ceil(min_ms / one))))
    for _ in range(calls):
        fn()
    return calls + 1


def timed_launches(fn, reps, warmup=2):
    """per-launch HIP-event times (ms) of `fn` on the library's stream, at steady-state clocks (ramp())"""
    from kmap_amd import _ffi
    ramp(fn)
    for _ in range(warmup):
        fn()
    evs = [_ffi.Event() for _ in range(reps + 1)]
    evs[0].record()
    for i in range(reps):
        fn()
        evs[i + 1].record()
    _ffi.sync()
    return [evs[i].elapsed_ms(evs[i + 1]) for i in range(reps)]


def roof(bytes_per_launch, ms_list, what, note=None):
    ms = statistics.median(ms_list)
    ach = bytes_per_launch / (ms * 1e-3) / 1e9
    d = {"what": what, "algorithmic_bytes": bytes_per_launch, "ms_median": ms, "ms_min": min(ms_list), "ms_mean": sum(ms_list) / len(ms_list),
         "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS}
    if note:
        d["note"] = note
    return d


# stage -> (workload of profiles/r*_pmc.json, [(kernel-name prefix, launches per stage pass)]): the kernels whose HBM bytes (rocprofv3
# FETCH_SIZE / WRITE_SIZE passes, corrected as MI355X_MICROARCH.md prescribes) make up `moved_bytes` of a stage
STAGE_KERNELS = {
    "count_pass_k8": ("e2e", [("hist_packed16_kernel", 1), ("compact_count_kernel", 1), ("compact_write_kernel<unsigned int>", 1)]),
    "count_k14_keyspace_rank": ("keyspace14", [("range_stage_kernel", 1), ("fine_count_kernel", 1), ("fine_offsets_kernel", 1), ("fine_scatter_kernel", 1),
                                               ("fine_zero_heavy_kernel", 1), ("fine_hist_kernel", 1), ("compact_count_kernel", 1),
                                               ("compact_write_kernel<unsigned int>", 1)]),
    "count_pass_k8_dedupe": ("e2e", [("dedupe_bitmap_packed_kernel<true>", 1), ("max_read_len_kernel", 1), ("hist_packed16_kernel", 1),
                                     ("compact_count_kernel", 1), ("compact_write_kernel<unsigned int>", 1)]),
    "count_pass_k14": ("count14", [("fine_count_kernel", 1), ("fine_offsets_kernel", 1), ("fine_scatter_kernel", 1), ("fine_zero_heavy_kernel", 1),
                                   ("fine_hist_kernel", 1), ("fine_spill_kernel", 1), ("rc_merge_tiles_kernel", 1), ("compact_write_kernel<unsigned int>", 1)]),
    "mask_k8": ("e2e", [("hits_planes_idx_kernel<8, false>", 1), ("mask_cover_packed_kernel", 1)]),
    "scan_k8_r2": ("e2e", [("hits_planes_idx_kernel<8, true>", 1), ("scan_hits_reads_fused_kernel", 1), ("scan_reorder_kernel", 1)]),
    "knn_select": ("e2e", [("knn_select_kernel", 1)]),
    "knn_sums": ("e2e", [("knn_profile_kernel", 1), ("knn_sums_mfma_kernel", 1)]),
    "embed_iter_fast": ("e2e", [("forces_sym2_kernel", 1), ("sym_apply_kernel", 1), ("reduce_loss_kernel", 1)]),
    "embed_iter_seq": ("seq", [("forces_seq_kernel", 1), ("apply_kernel<true>", 1)]),
}


def add_moved_bytes(stages):
    """`frac` of a stage prices SURVEY 8(d)'s ALGORITHMIC bytes; `frac_moved` prices the bytes the stage's kernels actually moved
    to and from HBM (PMC passes of the latest profiles/r*_pmc.json, collected offline by tools/profile_round.sh) over the same
    measured time -- a stage whose kernels read no matrix, or each pair once, is not credited with bytes it never moved"""
    pmc = sorted((ROOT / "profiles").glob("r*_pmc.json"))
    if not pmc:
        return
    try:
        wl = json.loads(pmc[-1].read_text())["workloads"]
    except Exception:   # noqa: BLE001
        return
    for name, (w, kernels) in STAGE_KERNELS.items():
        if name not in stages or w not in wl:
            continue
        total, missing = 0.0, []
        for prefix, times in kernels:
            hit = [v.get("hbm_bytes") for k, v in wl[w].items() if k.startswith(prefix) and v.get("hbm_bytes") is not None]
            if hit:
                total += times * sum(hit) / len(hit)
            else:
                missing.append(prefix)
        st = stages[name]
        if total > 0 and len(missing) < len(kernels):
            st["moved_bytes"] = total
            st["frac_moved"] = total / (st["ms_median"] * 1e-3) / 1e9 / HBM_PEAK_GBS
            st["moved_source"] = f"profiles/{pmc[-1].name} [{w}]" + (f"; no counters for {missing}" if missing else "")
            # the number to quote: where the kernels do not move SURVEY 8(d)'s bytes (they read no matrix, or each pair once, or move more
            # than the algorithm needs) the fraction of the bytes actually moved, otherwise the algorithmic one
            off = abs(total - st["algorithmic_bytes"]) > 0.1 * st["algorithmic_bytes"]
            st["frac_quote"] = st["frac_moved"] if off else st["frac"]
            st["frac_quote_is"] = "frac_moved" if off else "frac"


def stage_rooflines(reads, kh, lab, conseq_lens):
    """Every stage of the C3 path besides the headline kernel, timed in isolation on resident inputs with HIP events; bytes are
    SURVEY 8(d)'s algorithmic figures (1 B per read position and pass for the read stages -- the uint8 contract; the kernels
    actually read the 0.375 B/position packed stream --, N^2 + 2 N^2 for the smoothing, 2 N^2 per embedding iteration)."""
    import ctypes as C
    from kmap_amd import _ffi, visualization as V
    from kmap_amd.hamdist import hamdist_matrix_dev, pitch_for
    from kmap_amd.kmer_count import DeviceCounts, kmer2hash
    from kmap_amd.motif_discovery import DeviceSeq
    seq, borders = reads
    out = {}
    ds = DeviceSeq(seq, borders)
    dc = DeviceCounts()
    npos = float(len(seq))
    out["count_pass_k8"] = roof(npos + 4 ** 8 * 4, timed_launches(lambda: ds.count(dc, 8, dedupe=False, merge_revcom=True), 6),
                                "histogram count of all 8-mers + revcom merge + compaction (later find_motif rounds)")
    out["count_pass_k8_dedupe"] = roof(npos + 4 ** 8 * 4, timed_launches(lambda: ds.count(dc, 8, dedupe=True, merge_revcom=True), 4),
                                       "first find_motif round: per-read de-duplication fused in front of the histogram")
    out["count_pass_k14"] = roof(npos + 4 ** 14 * 4, timed_launches(lambda: ds.count(dc, 14, dedupe=False, merge_revcom=True), 4),
                                 "partitioned histogram (11 <= k <= 15) + tiled revcom merge")
    b14 = [(4 ** 14 * r // 8) & ~7 for r in range(9)]
    out["count_k14_keyspace_rank"] = roof(npos + 4 ** 14 * 4 / 8, timed_launches(lambda: ds.count_range(dc, 14, False, True, b14[3], b14[4] - b14[3]), 4),
                                         "ONE rank's count pass of a key-space-sharded run at G = 8 (rank 3): all reads in, its eighth of the merged k = 14 table out, no collective")
    cons = int(kmer2hash("CCTACGTA"))
    out["mask_k8"] = roof(npos, timed_launches(lambda: (ds.reset(), ds.mask(8, np.array([cons, cons ^ 0x1B]), np.array([2, 2]))), 6),
                          "mask_input: Hamming-ball flag + cover of two consensuses (incl. the n/8-byte restore)")
    lib = _ffi.lib()
    h = _ffi.vp()
    _ffi.check(lib.kmap_scan_create(C.byref(h)))
    ds.declare_layout(h.value)          # fixed-length reads: the per-read pass derives the borders from the read index (as DeviceSeq.scan does)
    tot = _ffi.i64(0)

    def scan(k, kh_, r):
        _ffi.check(lib.kmap_scan_run_packed_dev(h.value, ds.codes.ptr, ds.inval_orig.ptr, ds.n, ds.borders.ptr, ds.n_seq, k, kh_, r, 1,
                                                C.byref(tot), ds.planes.ptr, None))
    out["scan_k8_r2"] = roof(npos, timed_launches(lambda: scan(8, cons, 2), 6),
                             "occurrence scan of one consensus over all reads, device part (incl. its one host sync for the hit total)")
    out["scan_k14_r5"] = roof(npos, timed_launches(lambda: scan(14, int(kmer2hash("AGGACCTACGTACA")), 5), 4),
                              "C5-style ball scan (k = 14, radius 5) on the C3 reads")
    lib.kmap_scan_destroy(h.value)
    dc.close()
    ds.close()
    # ---- sampled k-mers: smoothing and embedding iterations at N = 50 k
    n = len(kh)
    ldd = pitch_for(n)
    kh_d, lab_d = _ffi.DeviceBuffer.from_numpy(kh), _ffi.DeviceBuffer.from_numpy(lab)
    D_d = _ffi.DeviceBuffer(n * ldd)
    hamdist_matrix_dev(kh_d.ptr, lab_d.ptr, n, K, conseq_lens, D_d.ptr, ldd)
    nb_d = _ffi.DeviceBuffer(n * 20 * 4)            # allocated once: hipMalloc / hipFree inside the timed loop made this stage noisy

    def select():
        _ffi.check(lib.kmap_knn_select_u8_dev(D_d.ptr, ldd, n, 20, 0, n, nb_d.ptr, None))
    out["knn_select"] = roof(float(n) * n, timed_launches(select, 6), "20 nearest rows per row of D (device tie rule), one read of D")
    lds = (n + 127) & ~127
    sums_d = _ffi.DeviceBuffer(n * lds * 2)
    out["knn_sums"] = roof(3.0 * n * n, timed_launches(lambda: V.knn_sums_kmers_dev(kh_d.ptr, lab_d.ptr, n, K, conseq_lens, nb_d, 20, out=sums_d.ptr, natural_diag=True), 4),
                           "neighbour sums from base-count profiles (8d: N^2 read at ideal reuse + 2 N^2 written; this kernel reads no matrix)")
    D_d.free()
    lut = V.hd_prob_lut(K, 20, 400 * K)
    ld0, ph = V._init_draws(n, 10, 7)
    for tag, mode, iters in (("embed_iter_fast", V.EMBED_FAST, 40), ("embed_iter_seq", V.EMBED_SEQ, 10)):
        sess = V.EmbedSession(n, 10, 0.01, mode)
        own = []
        if mode == V.EMBED_SEQ:    # as the product does: the repeated rows of the sample stored once, read through a row map
            comp_d, rowmap_d, stored = V.dedupe_sums_rows(sums_d, n, lds, free_input=False, n=n)
            _ffi.check(lib.kmap_embed_set_prob_lut(sess._h, comp_d.ptr, lds, _ffi.ptr(lut), len(lut)))
            if rowmap_d is not None:
                _ffi.check(lib.kmap_embed_set_row_map(sess._h, rowmap_d.ptr, stored))
                own = [comp_d, rowmap_d]
        else:
            _ffi.check(lib.kmap_embed_set_prob_lut(sess._h, sums_d.ptr, lds, _ffi.ptr(lut), len(lut)))    # sums stay ours (not in _keep)
        sess.set_coords(ld0, ph)
        sess.set_jitter(np.random.normal(0, 0.01, 4096))
        ms = [t / 5 for t in timed_launches(lambda: sess.step(5), iters // 5)]
        out[tag] = roof(2.0 * n * n, ms, "one embedding iteration = forces + loss reduction + apply (" +
                        ("FAST: symmetric kernel, each unordered pair once -> N^2 bytes of sums actually read" if mode == V.EMBED_FAST
                         else "SEQ: the reference's summation order, bit-pinned path") + ")")
        sess.close()
        for b in own:
            b.free()
    for b in (sums_d, nb_d, kh_d, lab_d):
        b.free()
    add_moved_bytes(out)
    return out


# ---- CPU baseline (the compiled port, timed on this box's host cores; the checker's code, measured -- never shipped) ---------
def cpu_baseline(kh, lab, conseq_lens, quick=False):
    """SURVEY 8(d) / BASELINE.md "CPU baseline": the compiled OpenMP port on the GPU box's host cores, with ALL the cores this
    process may use (affinity mask capped by the cgroup CPU quota) and with ONE core; CPU model stated.  Bounded samples:
      * Hamming rows of the same N (oracle ko_hamdist_rows) -- the headline metric on the CPU;
      * find_motif for k = 6..9 on 1 M synthetic reads (a tenth of C3; 100 k reads on one core), every stage linear in the reads;
      * >= 10 embedding iterations at N = 50 000 (one pass over a row sample on one core; an iteration is linear in the rows)."""
    from kmap_amd import synth, visualization as V
    from kmap_amd.kmer_count import gen_motif_def_dict, read_default_config_file
    from oracle import baseline as B, oracle as O
    L, BL = O.lib(), B.lib()
    info = B.host_cpu_info()
    T = info["usable"]
    n = len(kh)
    kh64 = np.ascontiguousarray(kh, np.uint64)
    cl = np.ascontiguousarray(conseq_lens, np.int32)

    def ham(threads, target_s):
        BL.kb_set_threads(threads)                       # the oracle's OpenMP loops run with this many threads from here on
        rows = min(n, 64 * threads)
        out = np.empty((rows, n), np.uint8)
        L.ko_hamdist_rows(kh64, lab, n, K, cl, len(cl), 0, rows, out)          # warm-up (thread pool, page faults)
        done, t0 = 0, time.perf_counter()
        while True:                                                            # row blocks round-robin over the matrix
            L.ko_hamdist_rows(kh64, lab, n, K, cl, len(cl), done % max(n - rows + 1, 1), rows, out)
            done += rows
            dt = time.perf_counter() - t0
            if dt >= target_s:
                return done * n / dt, done, dt

    v_all, rows_all, dt_all = ham(T, 1.0 if quick else 4.0)
    v_one, rows_one, dt_one = ham(1, 1.0 if quick else 3.0)
    res = {"value": v_all, "unit": "pairs/s", "cores": T, "kind": "port", "cpu": info,
           "one_core": {"value": v_one, "unit": "pairs/s", "cores": 1},
           "sample": (f"Hamming rows of the timed N={n} sample: {rows_all} rows x {n} columns on {T} threads ({dt_all:.1f} s), {rows_one} rows on "
                      f"1 thread ({dt_one:.1f} s); oracle ko_hamdist_rows, OpenMP; {info['model']}, {info['logical_cpus']} logical CPUs on the host, "
                      f"{info['affinity_cpus']} in this process's affinity mask, cgroup quota {info['cgroup_cpu_quota']}")}
    # ---- the e2e components
    n_reads_all, n_reads_one = (100_000, 20_000) if quick else (1_000_000, 100_000)
    seq, borders = synth.synth_reads(n_reads_all, 150, 2)
    mdd = gen_motif_def_dict(read_default_config_file())

    def fm(nr, threads):
        sub, bsub = seq[:nr * 151].copy(), borders[:nr]
        t0 = time.perf_counter()
        found = {}
        for k in range(6, 10):
            found[k] = len(B.find_motif(sub.copy() if k < 9 else sub, bsub, k, mdd[k], threads=threads))
        return time.perf_counter() - t0, found
    fm_all, found_all = fm(n_reads_all, T)
    fm_one, _ = fm(n_reads_one, 1)
    del seq, borders
    lut = V.hd_prob_lut(K, 20, 400 * K)
    kh32 = np.ascontiguousarray(kh, np.uint32)
    y = np.random.default_rng(7).standard_normal((2, n)).astype(np.float32)
    t0 = time.perf_counter()
    g, loss0 = B.embed_forces_kmers(kh32, lut, 400, y, threads=T)
    t_first = time.perf_counter() - t0
    iters = 2 if quick else (10 if t_first * 10 <= 45.0 else max(3, int(45.0 / t_first)))
    t0 = time.perf_counter()
    for _ in range(iters):
        g, loss = B.embed_forces_kmers(kh32, lut, 400, y, threads=T)
        BL.kb_embed_update(y, g, n, 0.01)
    it_all = (time.perf_counter() - t0) / iters
    rows1 = int(min(n, max(64, n * (2.0 if quick else 6.0) / max(it_all * T, 1e-9))))
    t0 = time.perf_counter()
    B.embed_forces_kmers(kh32, lut, 400, y, 0, rows1, threads=1)
    it_one = (time.perf_counter() - t0) * n / rows1
    c3_reads, c3_iters = 10_000_000, 2500
    res["e2e"] = {
        "kind": "port", "cores": T,
        "measured": {"find_motif_k6_9_s": {"all_cores": fm_all, "reads": n_reads_all, "one_core": fm_one, "reads_one_core": n_reads_one,
                                          "motifs_found_per_k": found_all},
                     "embed_s_per_iteration": {"all_cores": it_all, "iterations": iters, "n_kmers": n, "one_core": it_one,
                                               "one_core_rows": rows1, "loss_first_iteration_x2": 2.0 * loss0}},
        "extrapolated_to_c3": {"find_motif_k6_9_s": {"all_cores": fm_all * c3_reads / n_reads_all, "one_core": fm_one * c3_reads / n_reads_one},
                               "embedding_2500_iterations_s": {"all_cores": it_all * c3_iters, "one_core": it_one * c3_iters},
                               "total_s": {"all_cores": fm_all * c3_reads / n_reads_all + it_all * c3_iters,
                                           "one_core": fm_one * c3_reads / n_reads_one + it_one * c3_iters},
                               "note": "linear extrapolation (reads x, iterations x); the occurrence scans, neighbour smoothing and file "
                                       "writing of the two verbs are NOT included -- a lower bound of the CPU job"},
        "sample": (f"oracle/kmap_cpu_baseline.c (gcc -O2 -fopenmp): find_motif k=6..9 on {n_reads_all} x 150 bp synthetic reads with {T} threads and on "
                   f"{n_reads_one} reads with 1 thread; {iters} embedding iterations at N={n} with {T} threads (fused single pass, the "
                   f"reference's f32 summation order, p looked up from the k-mers on the fly instead of a 10 GB matrix -- cheaper than the "
                   f"reference's data flow), one pass over {rows1} of {n} rows with 1 thread")}
    return res


# ---- multi-GPU legs ----------------------------------------------------------------------------------------------------
def _barrier(dist, torch):
    """every rank has arrived: a one-element all-reduce on the device + its read-back.  (torch's dist.barrier() on the gloo group of the
    one-GPU rehearsals never returned once the key-range legs had run -- all five ranks stood in it: gpurun_out/r6_reh5d.err -- while
    the all-reduces around it kept completing; on RCCL a barrier IS such an all-reduce.)"""
    t = torch.zeros(1, dtype=torch.int32, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def _all_ok(dist, torch, flag):
    """has any rank failed so far (a rank that failed sets its flag; every rank sees it before anyone enters the next phase)"""
    if dist is None:
        return int(flag.item()) == 0
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    return int(flag.item()) == 0


def embed_dist_leg(dist, torch, world, kh, lab, conseqs, iters=200, mode=None, with_direct=True):
    """STRONG scaling of the C3 embedding: the N = 50 000 hand-over sample FIXED, sharded over all ranks (kmap_amd.distributed):
    ms per iteration of the loop (device-synchronised on both sides, max over ranks) and, from a separate short run with device
    events around the phases, forces / collective / apply ms.  On one rank the all-reduce is issued anyway (one-rank RCCL
    group) and the resident single-GPU loop is timed next to it: their difference is the plumbing's own cost.
    Errors are reported in the line, not raised: the headline above is already measured."""
    from kmap_amd import visualization as V
    from kmap_amd.distributed import kmap_from_kmers_distributed
    mode = V.EMBED_FAST if mode is None else mode
    n = len(kh)
    ones = np.ones(n, np.int64)
    err, loop_s, hbm, phases = "", 0.0, {}, None
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    for it, prof in ((24, 20), (iters, 0)):
        if not _all_ok(dist, torch, flag):
            break
        _barrier(dist, torch)
        tr = {}
        try:
            kmap_from_kmers_distributed(kh, ones, lab, conseqs, K, n_max_iter=it, random_seed=7, trace=tr, mode=mode,
                                        always_collective=True, profile_iters=prof)
            loop_s, hbm = tr["loop_s"], tr["hbm"]
            phases = tr.get("phases", phases)
        except Exception as e:   # noqa: BLE001
            err = f"{type(e).__name__}: {e}"[:300]
            flag.fill_(1)
            import traceback
            print(f"bench.py: embed_dist_leg failed on rank {dist.get_rank()}: {err}\n{traceback.format_exc()}", file=sys.stderr, flush=True)
    if not _all_ok(dist, torch, flag):
        return {"error": err or "another rank failed"}
    t = torch.tensor([loop_s], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    cyc = mode == V.EMBED_FAST and os.environ.get("KMAP_DIST_CYCLIC", "1") != "0" and n >= 16384 and world > 1
    # the same loop with the peer-direct exchange (kmap_amd.distributed.PeerExchange: IPC-mapped receive areas, push + flag,
    # no library call between iterations) next to the all-reduce form; its failure is reported, not raised
    direct = {}
    if not with_direct:
        direct = {"skipped": "the peer-direct exchange is timed in the SEQ leg (the package default mode) only"}
    elif _all_ok(dist, torch, flag):
        d_err, d_loop = "", 0.0
        for it in (24, iters):
            if not _all_ok(dist, torch, flag):
                break
            _barrier(dist, torch)
            tr = {}
            try:
                kmap_from_kmers_distributed(kh, ones, lab, conseqs, K, n_max_iter=it, random_seed=7, trace=tr, mode=mode, exchange="direct")
                d_loop = tr["loop_s"]
            except Exception as e:   # noqa: BLE001
                d_err = f"{type(e).__name__}: {e}"[:300]
                flag.fill_(1)
        if _all_ok(dist, torch, flag):
            td = torch.tensor([d_loop], dtype=torch.float64, device="cuda")
            dist.all_reduce(td, op=dist.ReduceOp.MAX)
            direct = {"loop_s": float(td.item()), "ms_per_iteration": float(td.item()) / iters * 1e3,
                      "what": "forces -> push kernel (peer-to-peer stores into every rank's IPC-mapped receive area + flag) -> apply (bounded wait, "
                              "sum of the slots in rank order); the whole segment issued by one native call"}
        else:
            direct = {"error": d_err or "another rank failed"}
            flag.zero_()                                  # the all-reduce numbers above stand
    res = {"n_kmers": n, "scaling": "strong", "layout": "each unordered pair once, cyclic 256-row blocks per rank" if cyc else "contiguous row blocks",
           "mode": "FAST" if mode == V.EMBED_FAST else "SEQ (the reference's summation order: the parity-grade loop)", "iterations": iters, "loop_s": float(t.item()), "ms_per_iteration": float(t.item()) / iters * 1e3,
           "collectives_per_iteration": 1, "message_bytes": (2 * n + 8) * 4, "phases_ms_rank0": phases,
           "phases_note": "device events around forces (force kernel + partial sums + loss limbs), the all-reduce, apply; 20 iterations of a separate run",
           "d_rows_per_rank": hbm.get("d_rows"), "d_bytes_per_rank": hbm.get("d_bytes"), "exchange_direct": direct}
    if world == 1:
        tr = {}
        V.kmap_from_kmers(kh, ones, lab, conseqs, K, n_max_iter=24, random_seed=7, mode=mode)
        V.kmap_from_kmers(kh, ones, lab, conseqs, K, n_max_iter=iters, random_seed=7, trace=tr, mode=mode)
        res["resident_ms_per_iteration"] = tr["loop_s"] / iters * 1e3
        res["overhead_ms_per_iter"] = res["ms_per_iteration"] - res["resident_ms_per_iteration"]
        res["overhead_note"] = ("sharded loop on a one-rank RCCL group (forces_msg -> all_reduce -> apply_msg, issued from Python) minus the "
                                "resident loop (kmap_embed_step) on the same GPU, same N, same iteration count")
    return res


def count_dist_leg(dist, torch, world, k=15, n_reads=1_000_000, read_len=150, reps=3):
    """Multi-GPU counting at a k whose 4^k table is large (k = 15: 4 GiB), one count pass with per-read dedupe and reverse-complement
    merge in three forms -- (a) reads sharded, all-reduce of the whole table, every rank compacts it; (b) reads sharded, bins owned by key
    range: presence all-reduce (half a byte per bin) + one SUM-reduce per slice + all-gather of the compacted shards
    (make_dist_device_seq(shard_counts=True)); (c) KEY SPACE (the default from k = 13): every rank holds all reads and computes its key
    range alone, no table bytes exchanged (key_space=True).  ms per pass (max over ranks, best of `reps`), and that both
    give the same table.  1 M x 150 bp reads: 1.5e8 windows against 1.07e9 bins -- the sparse regime in which make_dist_device_seq
    picks the key-range form by itself (every rank wants the WHOLE list, so its all-gather, 12 B per distinct k-mer, must stay
    below what the slices save: windows < 4^k / 4).  Errors are reported in the line, not raised."""
    from kmap_amd import _ffi, synth
    from kmap_amd.distributed import make_dist_device_seq
    from kmap_amd.kmer_count import DeviceCounts
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    res, err = {}, ""
    try:
        seq, borders = synth.synth_reads(n_reads, read_len, 11)          # the same array on every rank; a rank uploads its slice
        sums = {}
        forms = (("all_reduce", dict(shard_counts=False, key_space=False)), ("key_range", dict(shard_counts=True, key_space=False)),
                 ("key_space", dict(key_space=True)))
        for name, kw in forms:
            ds = make_dist_device_seq(seq, borders, dist, **kw)
            dc = DeviceCounts()
            best = None
            for rep in range(reps + 1):                                  # the first pass allocates the table and the communicator's buffers
                _barrier(dist, torch)
                _ffi.sync()
                t0 = time.perf_counter()
                ds.count(dc, k, dedupe=True, merge_revcom=True)
                torch.cuda.synchronize()
                _ffi.sync()
                dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
                dist.all_reduce(dt, op=dist.ReduceOp.MAX)
                if rep > 0:
                    best = float(dt.item()) if best is None else min(best, float(dt.item()))
            res[name + "_ms"] = best * 1e3
            sums[name] = (int(dc.n_uniq), int(dc.total()))
            dc.close()
            ds.close()
        res["same_table"] = sums["all_reduce"] == sums["key_range"] == sums["key_space"]
        res["n_uniq"], res["total_count"] = sums["key_range"]
    except Exception as e:   # noqa: BLE001
        err = f"{type(e).__name__}: {e}"[:300]
        flag.fill_(1)
    if not _all_ok(dist, torch, flag):
        return {"error": err or "another rank failed"}
    tb = 4 ** k * 4
    res.update({"k": k, "reads": n_reads, "read_len": read_len, "table_bytes": tb,
                "forms": "all_reduce: read shards + SUM all-reduce of the 4^k table; key_range (round 4): read shards + presence nibbles + one SUM-reduce per "
                         "slice; key_space (round 6, the default from k = 13): all reads on every rank, the rank's key range from the windows that decide it, "
                         "no table bytes exchanged",
                "bytes_received_per_rank": {"key_space": 0, "all_reduce": 2 * tb * (world - 1) // world,
                                            "key_range": (tb + 2 * (tb // 8) + (12 * res.get("n_uniq", 0) if res.get("n_uniq", 0) <= 4_000_000 else 0)) * (world - 1) // world,
                                            "note": "ring estimates; key_range = table slices + presence nibbles (all-reduced); above 4e6 distinct k-mers the "
                                                    "(k-mer, count) shards are NOT exchanged: the table stays sharded and find_motif works on local partials "
                                                    "(kmap_amd.distributed.CountShard); below, + the all-gathered shards (12 B per distinct k-mer)"}})
    return res


def reads_dist_leg(dist, torch, world, rank, res_dir, reps=3):
    """STRONG scaling of the read stages on the C3 reads (10 M x 150 bp, FIXED): contiguous read ranges per rank
    (kmap_amd.distributed.make_dist_device_seq; a rank maps the res_dir's pickles and uploads only its slice), one pass each of
    count k = 8 with per-read dedupe, count k = 14, occurrence scan k = 8 r = 2 -- the collective of the pass included (all-reduce
    of the 4^k table; device all-gather of the hit lists).  ms per pass = best of `reps`, max over ranks; `frac_per_gpu` prices a
    rank's own positions (1 B each) over that time against ITS 8 TB/s.  Errors are reported in the line, not raised."""
    from kmap_amd import _ffi
    from kmap_amd.distributed import make_dist_device_seq
    from kmap_amd.kmer_count import DeviceCounts, kmer2hash, load_array_pickle
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    res, err = {}, ""
    try:
        seq = load_array_pickle(Path(res_dir) / "input.bin.pkl", populate=False)
        borders = load_array_pickle(Path(res_dir) / "input.seqboarder.bin.pkl", populate=False)
        ds = make_dist_device_seq(seq, borders, dist)
        dc = DeviceCounts()
        cons = int(kmer2hash("CCTACGTA"))
        scan = ds.scan_lazy if getattr(ds, "scan_lazy", None) is not None else ds.scan
        passes = {"count_k8_dedupe": lambda: ds.count(dc, 8, dedupe=True, merge_revcom=True),
                  "count_k14": lambda: ds.count(dc, 14, dedupe=False, merge_revcom=True),
                  "scan_k8_r2": lambda: scan(8, cons, 2, True)}
        local = float(ds.n)
        for name, fn in passes.items():
            best = None
            for rep in range(reps + 1):                   # the first pass allocates tables / communicator buffers
                _barrier(dist, torch)
                torch.cuda.synchronize()
                _ffi.sync()
                t0 = time.perf_counter()
                fn()
                torch.cuda.synchronize()
                _ffi.sync()
                dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
                dist.all_reduce(dt, op=dist.ReduceOp.MAX)
                if rep > 0:
                    best = float(dt.item()) if best is None else min(best, float(dt.item()))
            res[name] = {"ms": best * 1e3, "positions_this_rank": local, "frac_per_gpu": local / best / 1e9 / HBM_PEAK_GBS}
        res["n_uniq_k14"] = int(dc.n_uniq)
        dc.close()
        ds.close()
    except Exception as e:   # noqa: BLE001
        err = f"{type(e).__name__}: {e}"[:300]
        flag.fill_(1)
    if not _all_ok(dist, torch, flag):
        return {"error": err or "another rank failed"}
    res.update({"scaling": "strong", "reads": int(len(borders)), "what": "C3 reads sharded by contiguous read ranges; wall time of one pass incl. "
                "its collective (host-synchronised on both sides), best of %d, max over ranks" % reps})
    return res


def e2e_dist_leg(dist, rank, reads, modes=("default", "fast")):
    """north_star's own metric under the process group: `scan_motif` (reads sharded) + `visualize_kmers` (rows sharded) on a clean C3
    res_dir, wall time of the slowest rank (kmap_amd.e2e.run_e2e_dist); default = SEQ, the reference's arithmetic."""
    from kmap_amd.e2e import run_e2e_dist
    out = {}
    for mode in modes:
        try:
            r = run_e2e_dist(dist, "C3", mode, reads=reads if rank == 0 else None)
            if r is not None:
                out[mode] = {"scan_motif_s": r["times"]["scan_motif_s"], "visualize_kmers_s": r["times"]["visualize_kmers_s"],
                             "e2e_s": r["times"]["e2e_s"], "final_conseq": r["final_conseq"], "stages": r["stages"]}
        except Exception as e:   # noqa: BLE001 -- a verb that raises on one rank ends the job at the launcher; what can be caught is reported
            out[mode] = {"error": f"{type(e).__name__}: {e}"[:300]}
            break
    out["workload"] = ("C3 (10 M x 150 bp synthetic reads, k = 6..9, N = 50 000, 2500 iterations), both verbs under the process group, clean res_dir, "
                       "wall time = max over ranks; default = SEQ embedding (the reference's summation order), fast = visualization.embed_mode = \"fast\"")
    return out


def shard_proxy(G, reads, res_dir, c3s, overhead_ms, first=None):
    """ONE GPU, N = 1 only: every rank's share of a G-GPU run, one after the other on this GPU, so that the shapes a G-GPU node will
    run are measured before such a node exists.  Per stage: `shard_ms` (HIP-event median per shard), `max_shard_ms`,
    `one_gpu_ms` (the unsharded stage, same code, same run), `work_inflation` = sum of the shards / one_gpu_ms, and
    `predicted_ms` = max_shard_ms (+ the per-iteration exchange overhead measured on the one-rank group for the embedding
    stages; the collectives of the read stages are NOT in it -- their bytes are stated instead).  Stages: Hamming rows, neighbour
    selection, neighbour sums, SEQ forces of contiguous rows, FAST forces of cyclic blocks at N = 50 000 (C3) and N = 200 000 (C4);
    count k = 8 with dedupe / count k = 14 / scan k = 8 r = 2 over reads / G."""
    import ctypes as C
    from kmap_amd import _ffi, visualization as V
    from kmap_amd.distributed import row_partition
    from kmap_amd.hamdist import hamdist_matrix_dev, pitch_for
    from kmap_amd.kmer_count import DeviceCounts, kmer2hash
    from kmap_amd.motif_discovery import DeviceSeq
    lib = _ffi.lib()
    out = {"G": G, "stages": {}}

    def entry(shard, one, extra_ms=0.0, note=None):
        d = {"shard_ms": shard, "max_shard_ms": max(shard), "one_gpu_ms": one, "work_inflation": sum(shard) / one,
             "predicted_ms": max(shard) + extra_ms, "predicted_speedup": one / (max(shard) + extra_ms)}
        if note:
            d["note"] = note
        return d

    def med(fn, reps):
        return statistics.median(timed_launches(fn, reps, warmup=1))

    for tag, n in (("c3", N_BASE), ("c4", N_C4)):
        if tag == "c3":
            kh, lab, conseqs = c3s
            lens = [len(c) for c in conseqs]
        else:
            kh, lab, lens, conseqs = pipeline_sample(res_dir, n)
        n = len(kh)
        ldd, lds = pitch_for(n), (n + 127) & ~127
        kh_d, lab_d = _ffi.DeviceBuffer.from_numpy(kh), _ffi.DeviceBuffer.from_numpy(lab)
        parts = [(0, n)] + [row_partition(n, G, r) for r in range(G)]            # the unsharded stage first, then the G shards
        nb_all = _ffi.DeviceBuffer(n * 20 * 4)
        ham, sel, sums_ms, seq_ms = [], [], [], []
        lut = V.hd_prob_lut(K, 20, 400 * K)
        ld0, ph = V._init_draws(n, 0, 7)
        reps = 5 if n <= N_BASE else 3
        # pass 1: Hamming rows + neighbour selection of every share (the shares' tables together = the all-gathered table)
        for row0, nrows in parts:
            D_d = _ffi.DeviceBuffer(nrows * ldd)
            ham.append(med(lambda: hamdist_matrix_dev(kh_d.ptr, lab_d.ptr, n, K, lens, D_d.ptr, ldd, row0=row0, nrows=nrows), reps))
            sel.append(med(lambda: _ffi.check(lib.kmap_knn_select_u8_dev(D_d.ptr, ldd, n, 20, 0, nrows, nb_all.ptr + row0 * 80, None)), reps))
            D_d.free()
        # pass 2: neighbour sums + SEQ forces of the share's rows
        for row0, nrows in parts:
            sums_d = _ffi.DeviceBuffer(nrows * lds * 2)
            sums_ms.append(med(lambda: V.knn_sums_kmers_dev(kh_d.ptr, lab_d.ptr, n, K, lens, nb_all, 20, row0=row0, nrows=nrows, out=sums_d.ptr, natural_diag=True), reps))
            sess = V.EmbedSession(n, 1, 0.01, V.EMBED_SEQ, row0=row0, nrows=nrows)
            sums_d, rowmap_d, stored = V.dedupe_sums_rows(sums_d, nrows, lds, n=n)       # as the product does (repeated rows stored once)
            _ffi.check(lib.kmap_embed_set_prob_lut(sess._h, sums_d.ptr, lds, _ffi.ptr(lut), len(lut)))
            if rowmap_d is not None:
                _ffi.check(lib.kmap_embed_set_row_map(sess._h, rowmap_d.ptr, stored))
            sess.set_coords(ld0, None)
            seq_ms.append(med(lambda: sess.forces(), reps))
            sess.close()
            sums_d.free()
            if rowmap_d is not None:
                rowmap_d.free()
        # pass 3: FAST forces, cyclic 256-row blocks of the symmetric kernel (rank r owns blocks r, r + G, ...)
        fast_ms = []
        for world, ranks in ((1, [0]), (G, list(range(G)))):
            for r in ranks:
                blocks = V.cyclic_blocks(n, world, r) if world > 1 else [(0, n)]
                blk_bytes = (V.CYCLIC_BLOCK_ROWS if world > 1 else n) * lds * 2
                sums_d = _ffi.DeviceBuffer(max(len(blocks), 1) * blk_bytes)
                for b, (r0, nr) in enumerate(blocks):
                    V.knn_sums_kmers_dev(kh_d.ptr, lab_d.ptr, n, K, lens, nb_all, 20, row0=r0, nrows=nr, out=sums_d.ptr + b * blk_bytes)
                sess = V.EmbedSession(n, 1, 0.01, V.EMBED_FAST, cyclic=(world, r)) if world > 1 else V.EmbedSession(n, 1, 0.01, V.EMBED_FAST)
                _ffi.check(lib.kmap_embed_set_prob_lut(sess._h, sums_d.ptr, lds, _ffi.ptr(lut), len(lut)))
                sess.set_coords(ld0, None)
                fast_ms.append(med(lambda: sess.forces(), reps))
                sess.close()
                sums_d.free()
        for b in (nb_all, kh_d, lab_d):
            b.free()
        st = out["stages"]
        st[f"{tag}_hamming_rows"] = entry(ham[1:], ham[0])
        st[f"{tag}_knn_select"] = entry(sel[1:], sel[0])
        st[f"{tag}_knn_sums"] = entry(sums_ms[1:], sums_ms[0])
        st[f"{tag}_embed_forces_seq"] = entry(seq_ms[1:], seq_ms[0], overhead_ms, f"rows [r N / {G}, (r + 1) N / {G}) x all {n} columns, the reference's summation order; "
                                              "+ exchange overhead of the one-rank group per iteration")
        st[f"{tag}_embed_forces_fast"] = entry(fast_ms[1:], fast_ms[0], overhead_ms, "cyclic 256-row blocks, each unordered pair once")
    # ---- read stages on reads / G
    seq, borders = reads
    nreads = len(borders)
    cons = int(kmer2hash("CCTACGTA"))
    t = {"count_k8_dedupe": [], "count_k14": [], "scan_k8_r2": []}
    h = _ffi.vp()
    _ffi.check(lib.kmap_scan_create(C.byref(h)))
    tot = _ffi.i64(0)
    for r0, nr in [(0, nreads)] + [row_partition(nreads, G, r) for r in range(G)]:
        lo, hi = int(borders[r0, 0]), int(borders[r0 + nr - 1, 1]) + 1
        ds = DeviceSeq(np.ascontiguousarray(seq[lo:hi]), borders[r0:r0 + nr] - lo)
        ds.declare_layout(h.value)
        dc = DeviceCounts()
        t["count_k8_dedupe"].append(med(lambda: ds.count(dc, 8, dedupe=True, merge_revcom=True), 4))
        t["count_k14"].append(med(lambda: ds.count(dc, 14, dedupe=False, merge_revcom=True), 4))
        t["scan_k8_r2"].append(med(lambda: _ffi.check(lib.kmap_scan_run_packed_dev(h.value, ds.codes.ptr, ds.inval_orig.ptr, ds.n, ds.borders.ptr, ds.n_seq, 8, cons, 2, 1,
                                                                               C.byref(tot), ds.planes.ptr, None)), 4))
        dc.close()
        ds.close()
    lib.kmap_scan_destroy(h.value)
    coll = {"count_k8_dedupe": "all-reduce of 4^8 x 4 B = 256 KiB", "count_k14": "all-reduce of 4^14 x 4 B = 1 GiB (ring: 2 (G-1)/G x 1 GiB per link pair)",
            "scan_k8_r2": "all-gather of the hit lists"}
    for name, v in t.items():
        out["stages"]["reads_" + name] = entry(v[1:], v[0], 0.0, "collective not included: " + coll[name])
    # counting by KEY SPACE (the multi-rank form from k = 13 on): every rank holds all reads and computes its key range of the table --
    # a rank's pass is the whole exchange-free job (only the shard sizes travel afterwards)
    ds = DeviceSeq(seq, borders)
    dc = DeviceCounts()
    for k in (14, 16):
        nb = 4 ** k
        bounds = [(nb * r // G) & ~7 for r in range(G)] + [nb]
        one = med(lambda: ds.count(dc, k, dedupe=False, merge_revcom=True), 3)
        sh = [med(lambda: ds.count_range(dc, k, False, True, bounds[r], bounds[r + 1] - bounds[r]), 3) for r in range(G)]
        e = entry(sh, one, 0.0, "key-space form: all reads on every rank, the rank's key range of the table from the windows that decide it; "
                                "table bytes exchanged: 0 (read-sharded form above: + an all-reduce of 4^k x 4 B)")
        e["table_bytes_exchanged"] = 0
        out["stages"][f"reads_count_k{k}_keyspace"] = e
    dc.close()
    ds.close()
    out["exchange_overhead_ms_per_iteration"] = overhead_ms
    if first is not None:
        out.update(predict_e2e(G, first, out["stages"]))
    out["what"] = (f"every rank's share of a {G}-GPU run timed on ONE GPU, shard after shard (HIP events, median); predicted_ms = slowest shard "
                   f"(+ the measured one-rank exchange overhead for the embedding stages); not a measurement on {G} GPUs")
    return out


def predict_e2e(G, first, st):
    """What the C3 default run (k = 6..9, SEQ) measured at the top of this process would take on G GPUs, stage by stage, from the
    one-GPU stage timers of that run and the proxy's slowest-shard ratios -- the arithmetic is in `e2e_prediction` so that it can be
    checked line by line.  A stage either SHARDS (its one-GPU seconds x predicted_ms / one_gpu_ms of the matching proxy stage; the read
    stages' collectives are 256 KiB .. 1 MiB tables at k <= 9 and are not added) or stays SERIAL on rank 0 (as measured).  What the
    stage timers do not cover (interpreter work between stages, file writing) is the `untimed` remainder and stays serial."""
    S = dict(first["stages"])
    iters = first["iters"]

    def ratio(name):
        e = st.get(name)
        return min(1.0, e["predicted_ms"] / e["one_gpu_ms"]) if e else 1.0
    rows = []

    def add(stage, how, factor, seconds=None):
        t = S.get(stage, 0.0) if seconds is None else seconds
        rows.append({"stage": stage, "one_gpu_s": t, "how": how, "factor": factor, "predicted_s": t * factor})
    # scan_motif: the per-k stages are count + mask + top-k passes over the reads (first round with dedupe, later rounds without)
    r_cnt = ratio("reads_count_k8_dedupe")
    r_scan = ratio("reads_scan_k8_r2")
    add("load_inputs", "sharded: a rank maps the pickles and touches only its read range", 1.0 / G)
    add("upload", "sharded: a rank uploads and packs its read range", 1.0 / G)
    add("find_motif", "sharded reads (count / mask passes; ratio of reads_count_k8_dedupe) ", r_cnt)
    add("occurrence_per_k", "sharded reads (ratio of reads_scan_k8_r2); rank 0 writes the CSV", r_scan)
    add("occurrence_final", "sharded reads (ratio of reads_scan_k8_r2)", r_scan)
    for name in ("sample_kmers", "write_hamdist_pkl", "join_table_writers", "hamdist_matrix_int64"):
        if name in S:
            add(name, "serial on rank 0", 1.0)
    sm_known = sum(S.get(k_, 0.0) for k_ in ("load_inputs", "upload", "find_motif", "occurrence_per_k", "occurrence_final", "sample_kmers",
                                             "write_hamdist_pkl", "join_table_writers", "hamdist_matrix_int64"))
    add("scan_motif untimed", "serial (config, tables, merge of consensuses, file names)", 1.0, max(0.0, first["times"]["scan_motif_s"] - sm_known))
    # visualize_kmers
    add("viz_hamdist_matrix", "row blocks (ratio of c3_hamming_rows)", ratio("c3_hamming_rows"))
    add("viz_knn_select", "row blocks: every rank runs the reference's np.argpartition call on ITS rows of D, one all-gather of the indices", 1.0 / G)
    add("viz_knn_sums", "row blocks (ratio of c3_knn_sums)", ratio("c3_knn_sums"))
    if "viz_dedupe_sums" in S:
        add("viz_dedupe_sums", "row blocks", 1.0 / G)
    e = st.get("c3_embed_forces_seq")
    loop = S.get("viz_embed_loop", 0.0)
    if e:
        per_it_other = max(0.0, loop / iters * 1e3 - e["one_gpu_ms"])           # apply, loss, host logic per iteration: stays
        rows.append({"stage": "viz_embed_loop", "one_gpu_s": loop, "how": f"{iters} iterations x (slowest shard's SEQ forces {e['max_shard_ms']:.3f} ms + exchange "
                     f"{e['predicted_ms'] - e['max_shard_ms']:.3f} ms + per-iteration remainder {per_it_other:.3f} ms that does not shard)",
                     "factor": (e["predicted_ms"] + per_it_other) / max(loop / iters * 1e3, 1e-9), "predicted_s": iters * (e["predicted_ms"] + per_it_other) * 1e-3})
    else:
        add("viz_embed_loop", "no proxy stage: as measured", 1.0)
    vz_known = sum(S.get(k_, 0.0) for k_ in ("viz_hamdist_matrix", "viz_knn_select", "viz_knn_sums", "viz_dedupe_sums", "viz_embed_loop"))
    add("visualize_kmers untimed", "serial (reading the hand-over, writing low_dim_data.tsv)", 1.0, max(0.0, first["times"]["visualize_kmers_s"] - vz_known))
    total = sum(r_["predicted_s"] for r_ in rows)
    return {"e2e_predicted_s": total, "e2e_one_gpu_s": first["times"]["e2e_s"], "e2e_predicted_speedup": first["times"]["e2e_s"] / total,
            "e2e_prediction": rows,
            "e2e_prediction_note": f"C3 k = 6..9 default (SEQ) at G = {G}: sum of the rows; a prediction from one-GPU shard timings, not a measurement on {G} GPUs"}


def c4_leg(dist, torch, res_dir, rank, world, barrier):
    """BASELINE config C4: N = 200 000 sampled k-mers FIXED as the GPU count grows (strong scaling).  Hamming rows of this rank
    (HIP events, no collective) and the embedding iteration in both modes (FAST, and SEQ = the package default; the sharded loop with
    its one all-reduce for world > 1; phases from device events)."""
    from kmap_amd import _ffi, visualization as V
    from kmap_amd.distributed import row_partition
    from kmap_amd.hamdist import hamdist_matrix_dev, pitch_for
    kh, lab, lens, conseqs = pipeline_sample(res_dir, N_C4) if rank == 0 else (None, None, None, None)
    if world > 1:
        box = [kh, lab, lens, conseqs]
        dist.broadcast_object_list(box, 0)      # set-up only (1.6 MB once), outside every timed region
        kh, lab, lens, conseqs = box
    n = len(kh)
    row0, nrows = row_partition(n, world, rank)
    ld = pitch_for(n)
    kh_d, lab_d = _ffi.DeviceBuffer.from_numpy(kh), _ffi.DeviceBuffer.from_numpy(lab)
    out_d = _ffi.DeviceBuffer(max(nrows, 1) * ld)
    barrier()
    ms = timed_launches(lambda: hamdist_matrix_dev(kh_d.ptr, lab_d.ptr, n, K, lens, out_d.ptr, ld, row0=row0, nrows=nrows), 10)
    barrier()
    for b in (out_d, kh_d, lab_d):
        b.free()
    med = statistics.median(ms)
    if world > 1:                                # the slowest rank's median launch decides
        t = torch.tensor([med], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        med = float(t.item())
    res = {"n_kmers": n, "scaling": "strong", "rows_per_gpu": nrows, "hamming_ms_median": med, "hamming_ms_min_rank0": min(ms),
           "hamming_pairs_per_s": float(n) * n / (med * 1e-3),
           "hamming_note": "all N^2 pairs / the median launch time (HIP events, 10 launches at steady clocks) of the slowest rank's rows"}
    its = (5, 25)
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    # FAST (the round-1..4 number of this leg) and SEQ (the package default since round 5: the reference's summation order)
    for key, mode, name in (("embed", V.EMBED_FAST, "FAST"), ("embed_seq", V.EMBED_SEQ, "SEQ (package default)")):
        loops, err, phases = [], "", None
        for it in its:
            if not _all_ok(dist if world > 1 else None, torch, flag):      # a rank that failed in the previous run is seen by all before the next
                break
            tr = {}
            try:
                if world > 1:
                    from kmap_amd.distributed import kmap_from_kmers_distributed
                    kmap_from_kmers_distributed(kh, np.ones(n, np.int64), lab, conseqs, K, n_max_iter=it, random_seed=7, trace=tr, mode=mode,
                                                profile_iters=4 if it == its[0] else 0)
                    phases = tr.get("phases", phases)
                else:
                    V.kmap_from_kmers(kh, np.ones(n, np.int64), lab, conseqs, K, n_max_iter=it, random_seed=7, trace=tr, mode=mode)
                loops.append(tr["loop_s"])
            except Exception as e:   # noqa: BLE001 -- reported in the line; the headline above is already measured
                err = f"{type(e).__name__}: {e}"[:300]
                flag.fill_(1)
        if not _all_ok(dist if world > 1 else None, torch, flag):
            err = err or "another rank failed"
        if err or len(loops) < 2:
            res[key + "_error"] = err or "embedding leg did not run"
            flag.zero_()                                                   # the other mode still runs
            continue
        per_it = (loops[1] - loops[0]) / (its[1] - its[0])
        if world > 1:
            t = torch.tensor([per_it], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            per_it = float(t.item())
            res[key + "_phases_ms_rank0"] = phases
        res[key + "_ms_per_iteration"] = per_it * 1e3
        res[key + "_note"] = (f"{name}, from the difference of a {its[1]}- and a {its[0]}-iteration run (max over ranks)" +
                              ("; one all-reduce of (2 N + 8) floats per iteration" if world > 1 else "; resident single-GPU loop"))
    return res


def c5_leg(reps=5):
    """BASELINE config C5 at FULL size: k = 14, max_ham_dist = 5 Hamming-ball (occurrence) scan over 50 M x 300 bp reads =
    1.505e10 positions, generated in HBM (csrc/synth.hip; numpy needs minutes for the 15 GB array), packed to 0.375 B/position.
    Device time of one consensus (HIP events; incl. the scan's one host sync for the hit total) against SURVEY 8(d)'s
    1 B/position, and a spot check of the timed result against the oracle on reads fetched back from the generated array."""
    import ctypes as C
    from kmap_amd import _ffi, synth
    from kmap_amd.kmer_count import kmer2hash
    from oracle import oracle as O
    n_reads, L, k, radius = 50_000_000, 300, 14, 5
    motif = "AGGACCTACGTACA"
    t0 = time.perf_counter()
    ds, raw = synth.synth_reads_dev(n_reads, L, 3, motifs=(motif, "AATCGATAGC"), keep_raw=True)
    _ffi.sync()
    t_gen = time.perf_counter() - t0
    lib = _ffi.lib()
    h = _ffi.vp()
    _ffi.check(lib.kmap_scan_create(C.byref(h)))
    ds.declare_layout(h.value)          # as DeviceSeq.scan does: fixed-length reads, borders derived from the read index
    tot = _ffi.i64(0)
    cons = int(kmer2hash(motif))

    def scan():
        _ffi.check(lib.kmap_scan_run_packed_dev(h.value, ds.codes.ptr, ds.inval_orig.ptr, ds.n, ds.borders.ptr, ds.n_seq, k, cons, radius, 1,
                                                C.byref(tot), ds.planes.ptr, None))
    ms = timed_launches(scan, reps, warmup=1)
    hits = np.empty(n_reads, np.int32)
    pos = np.empty(tot.value, np.int32)
    _ffi.check(lib.kmap_scan_fetch(h.value, _ffi.ptr(hits), None, _ffi.ptr(pos)))
    offs = np.concatenate([[0], np.cumsum(hits, dtype=np.int64)])
    buf, md = np.empty(L, np.int32), C.c_int(0)
    picks = np.unique(np.concatenate([[0, n_reads - 1], np.random.default_rng(5).integers(0, n_reads, 198)]))
    for r in picks:
        read = raw(int(r) * (L + 1), int(r) * (L + 1) + L)
        m = O.lib().ko_scan_read(read, L, k, cons, radius, 1, buf, md)
        assert m == hits[r] and np.array_equal(pos[offs[r]:offs[r + 1]], buf[:m]), f"C5 scan differs from the oracle on read {r}"
    lib.kmap_scan_destroy(h.value)
    raw.free()
    ds.close()
    d = roof(float(ds.n), ms, "C5: Hamming-ball scan, k = 14, radius 5, one consensus, 50 M x 300 bp reads, one GPU")
    d.update({"reads": n_reads, "read_len": L, "positions": ds.n, "k": k, "radius": radius, "positions_per_s": ds.n / (d["ms_median"] * 1e-3),
              "reads_with_hit": int(np.count_nonzero(hits)), "total_hits": int(tot.value), "generate_and_pack_s": t_gen,
              "spot_check": f"{len(picks)} reads == oracle ko_scan_read on the generated bytes",
              "data": "synthetic, generated in HBM by kmap_synth_reads_dev (seed 3; 40 % of the reads carry the 14-mer, 5 % substitutions)"})
    return d


def fill_rate(nbytes, reps=7):
    """what a plain device fill of the same number of bytes reaches on THIS box in THIS run (hipMemsetAsync on the library's
    stream, HIP events): the store-side ceiling the headline kernel can be held against, separating box-to-box spread from
    kernel quality"""
    from kmap_amd import _ffi
    buf = _ffi.DeviceBuffer(nbytes)
    ms = timed_launches(lambda: buf.zero(), reps, warmup=2)
    buf.free()
    med = statistics.median(ms)
    return {"GBps_median": nbytes / (med * 1e-3) / 1e9, "GBps_best": nbytes / (min(ms) * 1e-3) / 1e9, "ms_median": med, "bytes": nbytes,
            "what": "hipMemsetAsync of the same byte count, HIP events, same stream, same run"}


LINE_LIMIT = 4096       # bytes: the one JSON line on stdout never exceeds this (asserted before it is written)
T_START = time.perf_counter()


def _short(s, n):
    s = str(s)
    return s if len(s) <= n else s[:n - 1] + "~"


def compact_line(line, detail_path):
    """The ONE line of stdout: the contract keys, `config` (short strings), `roofline`, `cpu_baseline`, the rank identity counts and a
    handful of scalars of the other legs.  Everything else (stages, shard_proxy, e2e stage tables, embed_dist, c4, c5, per-rank
    identities, the prose) lives in the detail file named by `detail`.  Strict JSON (no NaN / Infinity), re-parsed and
    length-checked here; a line that would not fit loses its optional scalars first and is asserted to fit after that."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    out = {k: line[k] for k in keep}
    c = line["config"]
    out["config"] = {"workload": _short(c["workload"], 160), "n_kmers": c["n_kmers"], "k": c["k"], "rows_per_gpu": c["rows_per_gpu"],
                     "final_conseq": "/".join(c["final_conseq"])[:64], "parallelism": c["parallelism"]}
    r = line["roofline"]
    out["roofline"] = {"bound": r["bound"], "achieved": r["achieved"], "peak": r["peak"], "unit": r["unit"], "frac": r["frac"],
                       "traffic": r.get("traffic"), "kernel": _short(r["kernel"], 64), "kernel_ms": r["kernel_ms"],
                       "algorithmic_bytes": r["algorithmic_bytes"], "frac_of_achievable": r.get("frac_of_achievable")}
    cb = line.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                               "model": _short(cb.get("cpu", {}).get("model", ""), 64), "sample": _short(cb["sample"], 200)}
    for k in ("ranks_seen", "distinct_gpus", "dist_backend", "rccl_version"):
        out[k] = line.get(k)
    opt = {}

    def put(name, *path):
        d = line
        for p_ in path:
            if not isinstance(d, dict) or p_ not in d:
                return
            d = d[p_]
        if isinstance(d, float) and not math.isfinite(d):
            return
        if isinstance(d, (int, float, str, bool)):
            opt[name] = _short(d, 120) if isinstance(d, str) else d
    put("e2e_c3_s", "e2e", "k6_9", "default", "e2e_s")                 # N = 1: both verbs, C3 k 6..9, SEQ (package default)
    put("e2e_c3_s", "e2e", "default", "e2e_s")                         # N > 1: the same under the process group (max over ranks)
    put("e2e_c3_fast_s", "e2e", "k6_9", "fast", "e2e_s")
    put("e2e_c3_fast_s", "e2e", "fast", "e2e_s")
    put("e2e_c3_k6_16_s", "e2e", "k6_16", "default", "e2e_s")
    put("e2e_c2_s", "e2e", "c2", "default", "e2e_s")
    put("e2e_c4_s", "c4", "e2e_s")
    put("c5_frac", "c5", "frac")
    put("c5_ms", "c5", "ms_median")
    put("embed_seq_ms_per_iter", "embed_dist", "seq", "ms_per_iteration")
    put("embed_fast_ms_per_iter", "embed_dist", "ms_per_iteration")
    put("c4_hamming_pairs_per_s", "c4", "hamming_pairs_per_s")
    put("c4_embed_seq_ms_per_iter", "c4", "embed_seq_ms_per_iteration")
    put("e2e_predicted_s_g8", "shard_proxy", "e2e_predicted_s")
    put("cpu_e2e_c3_s", "cpu_baseline", "e2e", "extrapolated_to_c3", "total_s", "all_cores")
    put("watchdog", "watchdog")
    sk = line.get("skipped_legs")
    if sk:
        opt["skipped_legs"] = _short(",".join(sk), 160)
    er = line.get("leg_errors")
    if er:
        opt["leg_errors"] = _short(",".join(er), 160)
    out["detail"] = detail_path
    out["wall_s"] = round(time.perf_counter() - T_START, 1)
    names = list(opt)
    while True:
        cand = _clean(dict(out, **{k: opt[k] for k in names}))
        text = json.dumps(cand, allow_nan=False, separators=(",", ":"))
        if len(text.encode()) <= LINE_LIMIT or not names:
            break
        names.pop()
    assert len(text.encode()) <= LINE_LIMIT, f"bench line is {len(text.encode())} bytes"
    back = json.loads(text)
    assert back["metric"] == line["metric"] and "roofline" in back and "config" in back
    return text


def _clean(o):
    """detail file: strict JSON too (non-finite floats -> null, numpy scalars -> Python)"""
    if isinstance(o, dict):
        return {str(k): _clean(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_clean(v) for v in o]
    if isinstance(o, (np.integer,)):
        return int(o)
    if isinstance(o, (np.floating, float)):
        return float(o) if math.isfinite(float(o)) else None
    if isinstance(o, (str, int, bool)) or o is None:
        return o
    return str(o)


def write_detail(line, world):
    """the full record (every leg, stage table, per-rank identity, prose) -> gpurun_out/bench_detail_n{N}.json; returns the
    path relative to the repo root (or None when the directory cannot be written)"""
    rel = f"gpurun_out/bench_detail_n{world}.json"
    try:
        (ROOT / "gpurun_out").mkdir(exist_ok=True)
        (ROOT / rel).write_text(json.dumps(_clean(line), allow_nan=False, indent=1) + "\n")
        return rel
    except Exception as e:   # noqa: BLE001
        print(f"bench.py: detail file not written: {e}", file=sys.stderr)
        return None


def emit(fd, line, world):
    """detail file + stderr summary, then the compact line to the saved stdout descriptor"""
    rel = write_detail(line, world)
    text = compact_line(line, rel)
    print(f"bench.py: full record in {rel} ({(ROOT / rel).stat().st_size if rel else 0} bytes); line {len(text)} bytes", file=sys.stderr, flush=True)
    os.write(fd, (text + "\n").encode())


class Budget:
    """ONE deadline for the whole command (--time-budget seconds from process start) instead of per-leg limits that add up.
    A leg states what it is expected to cost; it starts only if that still fits (all ranks agree: MIN over ranks), otherwise its
    name goes to `skipped_legs`.  For world > 1 a watchdog thread fires at the deadline + grace: rank 0 writes the line as it
    stands (+ "watchdog"), every rank leaves with os._exit(WATCHDOG_RC) -- a collective that never completes cannot be
    cancelled from Python, and the headline is already measured."""

    def __init__(self, total_s, grace_s, world, rank, fd, dist=None, torch=None):
        self.total, self.world, self.rank, self.fd, self.dist, self.torch = total_s, world, rank, fd, dist, torch
        self.line, self.name, self.skipped, self.timer = None, "start-up", [], None
        if world > 1:
            import threading
            self.timer = threading.Timer(max(1.0, total_s + grace_s - self.elapsed()), self._fire)
            self.timer.daemon = True
            self.timer.start()

    def elapsed(self):
        return time.perf_counter() - T_START

    def left(self):
        return self.total - self.elapsed()

    def _fire(self):
        try:                                        # where every thread of this rank stands, for the post-mortem (stderr)
            import faulthandler
            faulthandler.dump_traceback(all_threads=True)
        except Exception:   # noqa: BLE001
            pass
        if self.rank == 0 and self.line is not None:
            self.line["watchdog"] = f"leg '{self.name}' still running at the deadline ({self.total:.0f} s + grace); ended by the watchdog"
            self.line["skipped_legs"] = self.skipped
            try:
                emit(self.fd, self.line, self.world)
            except Exception:   # noqa: BLE001
                pass
        os._exit(WATCHDOG_RC)

    def can(self, name, est_s):
        ok = self.left() >= est_s
        if self.dist is not None:
            t = self.torch.tensor([1 if ok else 0], dtype=self.torch.int32, device="cuda")
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
            ok = bool(int(t.item()))
        if ok:
            self.name = name
            print(f"bench.py: [{self.elapsed():6.1f} s] leg {name} (expected <= {est_s:.0f} s, {self.left():.0f} s left)", file=sys.stderr, flush=True)
        else:
            self.skipped.append(name)
            print(f"bench.py: [{self.elapsed():6.1f} s] leg {name} SKIPPED: needs ~{est_s:.0f} s, {self.left():.0f} s left", file=sys.stderr, flush=True)
        return ok

    def done(self):
        if self.timer is not None:
            self.timer.cancel()
            self.timer = None


WATCHDOG_RC = 3         # exit code of every rank when the run hit the global deadline inside a leg (the line is still printed)


def self_launch(n_gpus):
    """`python3 bench.py --gpus N` without a launcher around it: start `python -m torch.distributed.run --nproc-per-node N bench.py
    <same arguments>` as a CHILD process -- nothing in this process has touched HIP or torch.cuda yet, and nothing will: a process
    that has initialised the GPU must never exec another program on this pool -- relay the child's one JSON line to stdout
    (anything else the launcher or a library wrote there goes to stderr) and return its exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n_gpus)))
    print("bench.py: launching " + " ".join(cmd), file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True, start_new_session=True)
    line = None
    try:
        for out in proc.stdout:
            if out.lstrip().startswith('{"metric"'):
                line = out.strip()
            else:
                sys.stderr.write(out)
        rc = proc.wait()
    except BaseException:
        try:
            os.killpg(proc.pid, 15)      # exactly the process group started above
        except ProcessLookupError:
            pass
        raise
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        rc = 4                           # the launcher succeeded but no line came out: not a result
    return rc


def rank_identity(dist, torch, dev, world):
    """what proves that N ranks on N different GPUs produced the line: world size as the process group reports it, every rank's
    device name / PCI bus id / uuid (all-gathered), and the collective library's version"""
    p = torch.cuda.get_device_properties(dev)
    bus = "%04x:%02x:%02x" % tuple(int(getattr(p, a, -1)) & 0xffff for a in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
    me = {"rank": int(os.environ.get("RANK", "0")), "device_index": dev, "device_name": p.name, "pci_bus": bus, "uuid": str(getattr(p, "uuid", "")),
          "arch": getattr(p, "gcnArchName", ""), "pid": os.getpid()}
    ranks = [me]
    seen = 1
    backend = "none"
    if dist is not None:
        ranks = [None] * world
        dist.all_gather_object(ranks, me)
        seen = dist.get_world_size()
        backend = dist.get_backend()
    try:
        ver = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:   # noqa: BLE001
        ver = None
    return {"ranks_seen": seen, "backend": backend, "rccl_version": ver, "ranks": ranks,
            "distinct_gpus": len({(r["pci_bus"], r["uuid"]) for r in ranks})}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-embed-dist", action="store_true", help="skip the sharded-embedding leg")
    ap.add_argument("--no-c4", action="store_true", help="skip the N = 200 000 strong-scaling leg")
    ap.add_argument("--time-budget", type=float, default=270.0,
                    help="seconds the WHOLE command may take (one deadline): an optional leg starts only if its expected cost still fits; skipped legs are named in the line")
    ap.add_argument("--grace", type=float, default=60.0, help="multi-rank: seconds past the budget after which the watchdog prints the line as it stands and ends every rank")
    ap.add_argument("--no-c4-e2e", action="store_true", help="skip the whole C4 run on one GPU (N=1 only; ~50 s)")
    ap.add_argument("--no-count-dist", action="store_true", help="skip the multi-GPU counting leg (k = 15: all-reduce vs key-range shards)")
    ap.add_argument("--no-c5", action="store_true", help="skip the full-size C5 scan leg (N=1 only)")
    ap.add_argument("--quick", action="store_true", help="smaller CPU-baseline samples (rehearsals)")
    ap.add_argument("--shard-proxy", type=int, default=8, metavar="G",
                    help="N=1 only: time every rank's share of a G-GPU run on this one GPU (Hamming / kNN / SEQ + FAST forces at N = 50 000 and 200 000, "
                         "count / scan of reads / G) and print max-shard, work inflation and the predicted G-GPU time per stage; 0 = skip")
    ap.add_argument("--no-reads-dist", action="store_true", help="skip the read-sharded C3 count / scan leg (multi-rank)")
    ap.add_argument("--no-e2e-dist", action="store_true", help="skip the end-to-end run of both verbs under the process group (multi-rank)")
    ap.add_argument("--no-stages", action="store_true", help="skip the per-stage roofline timings")
    ap.add_argument("--e2e", default="full", choices=["none", "k9", "full"],
                    help="end-to-end timings on C3 (rank 0, N=1 only): k9 = k 6..9 in both embedding modes; full = also the default k 6..16")
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus))

    # stdout carries ONE JSON line and nothing else: libraries that print to fd 1 (RCCL's version banner at communicator
    # creation, for one) are sent to stderr for the rest of the run; the line itself goes to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        args.gpus = world

    import torch
    from kmap_amd import _ffi
    from kmap_amd.e2e import run_e2e, synth_config_reads
    from kmap_amd.hamdist import hamdist_matrix_dev, pitch_for
    # KMAP_BENCH_BACKEND=gloo + KMAP_BENCH_SAME_GPU=1 rehearse the multi-rank path on a one-GPU box (timing collectives
    # over gloo, every rank on GPU 0); the real runs use one GPU per rank and RCCL ("nccl").
    backend = os.environ.get("KMAP_BENCH_BACKEND", "nccl")
    dev = 0 if os.environ.get("KMAP_BENCH_SAME_GPU") else local_rank
    if os.environ.get("KMAP_BENCH_SAME_GPU"):
        os.environ["KMAP_DIST_SAME_GPU"] = "1"      # the verbs of the e2e leg pick their device the same way
    torch.cuda.set_device(dev)
    _ffi.check(_ffi.lib().kmap_set_device(dev))
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend)

    def barrier():
        if dist is not None:
            _barrier(dist, torch)
        torch.cuda.synchronize()
        _ffi.sync()

    # ---- the C3 pipeline (rank 0): its hand-over is the workload of the timed kernel; its wall time is e2e["k6_9"]["default"] ----
    n = int(round(N_BASE * math.sqrt(world) / 16)) * 16
    reads, first, res_dir = None, None, None
    if rank == 0:
        os.environ["KMAP_DIST_DISABLE"] = "1"       # rank 0 runs the pipeline on its own GPU; the verbs must not join the process group
        reads = synth_config_reads("C3")
        first = run_e2e("C3", "default", reads=reads, keep=True)
        os.environ.pop("KMAP_DIST_DISABLE")
        res_dir = first["res_dir"]
        kh, lab, lens, conseqs = pipeline_sample(res_dir, n)
        s50 = (kh, lab, lens, conseqs) if n == N_BASE else pipeline_sample(res_dir, N_BASE)
        c3s = (s50[0], s50[1], s50[3])              # the C3 hand-over itself, N = 50 000 (the strong-scaling embedding leg)
    else:
        kh = lab = lens = conseqs = c3s = None
    if dist is not None:
        box = [kh, lab, lens, conseqs, c3s]
        dist.broadcast_object_list(box, 0)          # set-up only, outside the timed region
        kh, lab, lens, conseqs, c3s = box
    n = len(kh)
    rows_per = (n + world - 1) // world
    row0 = rank * rows_per
    nrows = max(0, min(rows_per, n - row0))
    ld = pitch_for(n)
    kh_d, lab_d = _ffi.DeviceBuffer.from_numpy(kh), _ffi.DeviceBuffer.from_numpy(lab)
    out_d = _ffi.DeviceBuffer(max(nrows, 1) * ld)

    def step():
        hamdist_matrix_dev(kh_d.ptr, lab_d.ptr, n, K, lens, out_d.ptr, ld, row0=row0, nrows=nrows)

    evs = [_ffi.Event() for _ in range(args.steps + 1)]
    n_ramp = ramp(step)                 # clocks up (untimed, see ramp()), then the W warm-up steps, then the K timed ones
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(args.steps):
        step()
        evs[i + 1].record()
    barrier()
    wall = time.perf_counter() - t0
    per_launch = [evs[i].elapsed_ms(evs[i + 1]) for i in range(args.steps)]
    kern_ms = evs[0].elapsed_ms(evs[args.steps]) / args.steps
    if dist is not None:
        t = torch.tensor([wall], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())

    # spot-check the timed output against the oracle (a wrong fast kernel is not a result): rows of every label
    from oracle import oracle as O
    if nrows:
        picks = sorted({row0, row0 + nrows // 2, row0 + nrows - 1} | {int(np.searchsorted(lab, l)) for l in np.unique(lab) if row0 <= int(np.searchsorted(lab, l)) < row0 + nrows})
        for r in picks:
            got = out_d.to_numpy(np.uint8, (n,), offset=(r - row0) * ld)
            want = np.empty((1, n), np.uint8)
            O.lib().ko_hamdist_rows(np.ascontiguousarray(kh, np.uint64), lab, n, K, np.ascontiguousarray(lens, np.int32), len(lens), r, 1, want)
            assert np.array_equal(got, want[0]), f"timed kernel output differs from the oracle in row {r}"
    out_d.free()

    fill = fill_rate(max(nrows, 1) * ld) if rank == 0 else None

    ident = rank_identity(dist, torch, dev, world)      # all ranks (an all-gather of small objects, outside the timed region)
    line = None
    if rank == 0:
        pairs_total = float(n) * float(n)
        algo_bytes = nrows * n + 5 * n          # u8 out + u32 hashes + u8 group ids, per launch on this rank
        achieved = algo_bytes / (kern_ms * 1e-3) / 1e9
        short_rows = int(sum(int(np.count_nonzero(lab == l)) for l, c in enumerate(lens) if c < K))
        line = {
            "metric": "hamming_pairs_per_s", "value": pairs_total * args.steps / wall, "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"C3 Hamming stage: all-pairs Hamming matrix of the N={n} 8-mers the C3 pipeline samples "
                                   f"(finals {'/'.join(conseqs)}), uint8 out, row-sharded over {world} GPU(s)",
                       "n_kmers": n, "k": K, "final_conseq": conseqs, "conseq_lens": lens, "rows_of_short_consensus_labels": short_rows,
                       "rows_per_gpu": rows_per, "pairs_per_gpu_per_step": float(nrows) * n,
                       "parallelism": f"row blocks over {world} GPU(s); every rank holds all N hashes; no data-path collective",
                       "sample": "sample_kmers.pkl of scan_motif on 10M x 150 bp synthetic reads (k=6..9), expanded by counts"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "achievable": fill, "frac_of_achievable": achieved / fill["GBps_median"],
                         "kernel": "hamdist_tile_kernel<1 code word> (+ build_gid / build_codes pre-passes, inside the timed region)",
                         "kernel_ms": kern_ms, "kernel_ms_min": min(per_launch), "kernel_ms_median": statistics.median(per_launch),
                         "kernel_ms_max": max(per_launch), "algorithmic_bytes": algo_bytes,
                         "clock_ramp": (f"{n_ramp} untimed launches of the same step (>= {RAMP_MS:.0f} ms of sustained work) are queued right before "
                                        f"the {args.warmup} warm-up steps: after an idle gap the GPU needs ~20-40 ms to restore its clocks, and the first ~40 "
                                        f"launches take 0.50 instead of 0.395 ms (tools/hamdist_trend.py, profiles/r03_hamdist_trend.txt)")},
        }
        line.update({"ranks_seen": ident["ranks_seen"], "distinct_gpus": ident["distinct_gpus"], "dist_backend": ident["backend"],
                     "rccl_version": ident["rccl_version"], "ranks": ident["ranks"]})
        # HBM bytes per launch from the PMC passes of the same command (rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE, separate
        # runs, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes); collected offline, stored under profiles/
        pmc = sorted((ROOT / "profiles").glob("r*_bench_pmc.json"))
        if pmc and world == 1:
            try:
                kern = json.loads(pmc[-1].read_text())["kernels"]
                hk = [v for k_, v in kern.items() if "hamdist_tile_kernel" in k_ or "hamdist_matrix_kernel" in k_][0]
                line["roofline"]["traffic"] = hk["hbm_bytes_per_launch"]
                line["roofline"]["traffic_source"] = f"profiles/{pmc[-1].name}"
            except Exception:
                pass
    # ---- the optional legs (all ranks take part), most important first, under ONE deadline (Budget): a leg starts only if its
    # expected cost still fits.  The headline above is measured and in `line`; for world > 1 a watchdog lets rank 0 print the line
    # with what has finished if a collective never returns, instead of the whole run -- headline included -- dying at the launcher.
    budget = Budget(args.time_budget, args.grace, world, rank, json_fd, dist, torch)
    budget.line = line
    errors = []
    slow = [1.0]

    def leg(name, est_s, fn, key=None):
        """run one optional leg: budget-gated, its exception reported (`leg_errors`), its result under line[key or name]"""
        # what the legs so far took against their estimates stretches the later estimates (a slow box, a gloo rehearsal on one GPU)
        if not budget.can(name, est_s * slow[0]):
            return None
        t0 = time.perf_counter()
        try:
            res = fn()
        except AssertionError:
            raise                                   # a timed result that differs from the oracle is not a result
        except Exception as e:   # noqa: BLE001 -- reported; the headline is already measured
            res = {"error": f"{type(e).__name__}: {e}"[:300]}
            errors.append(name)
        took = time.perf_counter() - t0
        if took > 5.0:
            slow[0] = max(slow[0], min(took / est_s, 8.0))
        if isinstance(res, dict):
            res["leg_wall_s"] = took
        if line is not None and res is not None:
            line[key or name] = res
        return res

    G = max(world, 1)
    from kmap_amd.visualization import EMBED_SEQ
    if line is not None and world == 1 and not args.no_cpu_baseline:
        leg("cpu_baseline", 12 if args.quick else 40, lambda: cpu_baseline(kh, lab, lens, quick=args.quick))

    if dist is not None:
        box = [res_dir]
        dist.broadcast_object_list(box, 0)
        res_dir_all = box[0]
        if not args.no_e2e_dist:          # north_star's own metric first: both verbs under the process group, SEQ default
            leg("e2e", 25 + 60 / G, lambda: e2e_dist_leg(dist, rank, reads, modes=("default",)))
        if not args.no_embed_dist:
            leg("embed_dist.seq", 10 + 30 / G, lambda: embed_dist_leg(dist, torch, world, *c3s, iters=100, mode=EMBED_SEQ), key="embed_dist_seq")
        if not args.no_c4:
            leg("c4", 25 + 60 / G, lambda: c4_leg(dist, torch, res_dir_all, rank, world, barrier))
        if not args.no_reads_dist:
            def _reads():
                r_ = reads_dist_leg(dist, torch, world, rank, res_dir_all)
                _ffi.check(_ffi.lib().kmap_scratch_release(1 << 30))
                return r_
            leg("reads_dist", 30, _reads)
        if not args.no_count_dist:
            def _count():
                _ffi.check(_ffi.lib().kmap_scratch_release(1 << 30))
                r_ = count_dist_leg(dist, torch, world)
                _ffi.check(_ffi.lib().kmap_scratch_release(1 << 30))
                return r_
            leg("count_dist", 40, _count)
        if not args.no_embed_dist:
            leg("embed_dist", 10 + 20 / G, lambda: embed_dist_leg(dist, torch, world, *c3s, with_direct=False))
        if not args.no_e2e_dist:
            leg("e2e.fast", 25 + 40 / G, lambda: e2e_dist_leg(dist, rank, reads, modes=("fast",)), key="e2e_fast")
        if line is not None:              # one shape for the record whatever ran: embed_dist{..., seq}, e2e{default, fast, workload}
            ed = line.pop("embed_dist", None) or {}
            if "embed_dist_seq" in line:
                ed["seq"] = line.pop("embed_dist_seq")
            if ed:
                line["embed_dist"] = ed
            if "e2e_fast" in line:
                line.setdefault("e2e", {}).update({k_: v for k_, v in line.pop("e2e_fast").items() if k_ in ("fast",)})
    else:
        if args.e2e != "none":
            # the other half of BASELINE.json's metric: end-to-end wall time on a clean res_dir (synthetic reads are generated
            # and written outside the timed stages; scan_motif loads them from the pickles like the reference)
            def pack(r):
                return {"scan_motif_s": r["times"]["scan_motif_s"], "visualize_kmers_s": r["times"]["visualize_kmers_s"],
                        "e2e_s": r["times"]["e2e_s"], "final_conseq": r["final_conseq"], "stages": r["stages"]}
            e2e = {"k6_9": {"default": pack(first)}}
            line["e2e"] = e2e
            e2e["workload"] = (f"C3: {first['n_reads']} x {first['read_len']} bp synthetic reads, N={first['n_total']} sampled k-mers, "
                               f"{first['iters']} iterations, 1 GPU, clean res_dir; k6_9: k = 6..9 (longest final = the configs' k = 8), k6_16: the "
                               f"reference's default k range (default_config.toml:7-8); default = the package default = SEQ, the reference's "
                               f"arithmetic and summation order at every N, neighbours by np.argpartition up to N = 65536 (the parity-grade number; "
                               f"top-k / draws by the device rules above 4e6 unique k-mers), fast = config.toml visualization.embed_mode = \"fast\" (opt-in: wavefront-parallel "
                               f"row sums, per-step pinned), exact = config.toml general.exact = true (SEQ + np.argpartition neighbours / top-k + "
                               f"np.random.multinomial at every size: the strict drop-in run), reports = default + the reference's default report flags "
                               f"(motif_pos_density_flag, motif_co_occurence_flag, gen_hamball_flag) on")

            def into(d, key, fn):
                def run():
                    d[key] = pack(fn())
                    return None
                return run
            leg("e2e.fast", 8, into(e2e["k6_9"], "fast", lambda: run_e2e("C3", "fast", reads=reads)))
            if args.e2e == "full":
                leg("e2e.k6_16", 12, into(e2e.setdefault("k6_16", {}), "default", lambda: run_e2e("C3", "default", min_k=6, max_k=16, reads=reads)))
                leg("e2e.exact", 10, into(e2e["k6_9"], "exact", lambda: run_e2e("C3", "exact", reads=reads)))
                # BASELINE.md switches the report flags off on both sides; a user of the reference's config.toml has them ON: the
                # position-density, co-occurrence and Hamming-ball DATA files (no figures) next to the k = 6..9 run
                leg("e2e.reports", 10, into(e2e["k6_9"], "reports", lambda: run_e2e("C3", "default", reads=reads, reports=True)))

            def _c2():
                c2 = run_e2e("C2", "default")
                e2e["c2"] = {"default": pack(c2), "workload": (f"C2: {c2['n_reads']} x {c2['read_len']} bp synthetic reads, N={c2['n_total']} sampled k-mers, "
                                                               f"{c2['iters']} iterations (the reference's default size, default_config.toml:24-32), k = 6..9, "
                                                               f"package default embedding mode = SEQ (the reference's summation order), 1 GPU, clean res_dir")}
            leg("e2e.c2", 6, _c2)
        if not args.no_stages:
            def _stages():
                line["roofline"]["stages"] = stage_rooflines(reads, kh, lab, lens)
            leg("stages", 20, _stages)
        if not args.no_embed_dist:
            def _embed1():
                # one GPU: the sharded loop runs on a one-rank RCCL group, so its overhead is a number in the record
                import socket
                import torch.distributed as dist1
                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0))
                    port = sk.getsockname()[1]
                dist1.init_process_group("nccl" if backend == "nccl" else backend, init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                                         **({"device_id": torch.device("cuda", dev)} if backend == "nccl" else {}))
                try:
                    ed = embed_dist_leg(dist1, torch, 1, *c3s)
                    ed["seq"] = embed_dist_leg(dist1, torch, 1, *c3s, iters=100, mode=EMBED_SEQ)
                    return ed
                finally:
                    try:
                        dist1.destroy_process_group()
                    except Exception:   # noqa: BLE001
                        pass
            leg("embed_dist", 20, _embed1)
        if not args.no_c4:
            leg("c4", 25, lambda: c4_leg(None, torch, res_dir, rank, world, barrier))
        if args.shard_proxy > 1:
            ed = line.get("embed_dist") or {}
            ov = (ed.get("seq") or {}).get("overhead_ms_per_iter") or ed.get("overhead_ms_per_iter") or 0.0
            leg("shard_proxy", 35, lambda: shard_proxy(args.shard_proxy, reads, res_dir, c3s, max(float(ov), 0.0), first))
        reads = None              # 1.5 GB of host memory back before the next legs
        if not args.no_c5:
            leg("c5", 20, c5_leg)
        if not args.no_c4 and not args.no_c4_e2e:
            def _c4e2e():
                r_ = run_e2e("C4", "default")
                line.setdefault("c4", {}).update({"e2e_s": r_["times"]["e2e_s"], "e2e": {
                    "scan_motif_s": r_["times"]["scan_motif_s"], "visualize_kmers_s": r_["times"]["visualize_kmers_s"], "final_conseq": r_["final_conseq"],
                    "n_embedded": r_["n_embedded"], "stages": r_["stages"],
                    "workload": "C4 on ONE GPU: 10 M x 150 bp synthetic reads, k = 6..9, N = 200 000 sampled k-mers, 2500 iterations, SEQ (package default), clean res_dir"}})
            leg("c4.e2e", 75, _c4e2e)
    budget.done()
    kh_d.free()
    lab_d.free()
    if rank == 0:
        line["skipped_legs"] = budget.skipped
        line["leg_errors"] = errors
        line["time_budget_s"] = args.time_budget
        sys.stdout.flush()
        emit(json_fd, line, world)
    if res_dir:
        import shutil
        shutil.rmtree(res_dir, ignore_errors=True)
    if dist is not None:
        _barrier(dist, torch)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
