"""End-to-end `scan_motif` + `visualize_kmers` on seeded synthetic reads (BASELINE configs), with per-stage timers.
Used by tools/e2e.py and by bench.py's `e2e` leg.  Plot-only flags are off (BASELINE.md prescribes that for both
sides); k range 6..9 so that the longest final consensus is an 8-mer (the configs' "k = 8");
np.random.seed(123) before scan_motif; visualization.random_seed = 7."""
import os
import shutil
import tempfile
import time
from pathlib import Path

import numpy as np

CONFIGS = {
    "C1s": dict(n_reads=2000, read_len=60, seed=9, n_total=300, n_motif=150, iters=100),
    "C2": dict(n_reads=100_000, read_len=150, seed=1, n_total=5000, n_motif=2500, iters=2500),
    "C3": dict(n_reads=10_000_000, read_len=150, seed=2, n_total=50_000, n_motif=25_000, iters=2500),
    # BASELINE's 8-GPU configuration; on ONE GPU: 40 GB of distances + 80 GB of neighbour sums resident, device neighbour rule (N > 65 536)
    "C4": dict(n_reads=10_000_000, read_len=150, seed=2, n_total=200_000, n_motif=100_000, iters=2500),
}


def synth_config_reads(config):
    """(seq uint8 array, borders) of a config's seeded synthetic reads"""
    from . import synth
    c = CONFIGS[config]
    return synth.synth_reads(c["n_reads"], c["read_len"], c["seed"])


def run_e2e(config="C2", mode="default", min_k=6, max_k=9, iters=None, keep=False, quiet=True, reports=False, reads=None):
    """reads: (seq, borders) from synth_config_reads(config) to re-use across runs (not consumed); keep=True leaves the result
    directory in place and returns its path as "res_dir"."""
    import contextlib
    import io
    from . import motif_discovery as md, synth, visualization as vz
    c = CONFIGS[config]
    assert mode in ("default", "seq", "fast", "exact"), mode
    prev_mode = os.environ.get("KMAP_EMBED_MODE")
    os.environ.pop("KMAP_EMBED_MODE", None)       # the run is steered through config.toml's optional keys, like a user's
    md.STAGE_TIMES.clear()
    vz.STAGE_TIMES.clear()
    t = {}
    t0 = time.perf_counter()
    seq, borders = reads if reads is not None else synth.synth_reads(c["n_reads"], c["read_len"], c["seed"])
    t["synth_s"] = time.perf_counter() - t0
    res = Path(tempfile.mkdtemp(prefix=f"kmap_{config}_"))
    over = {"kmer_count": {"min_k": min_k, "max_k": max_k},
            "motif_discovery": {"motif_pos_density_flag": reports, "motif_co_occurence_flag": reports, "gen_hamball_flag": reports,
                                "n_total_sample": c["n_total"], "n_motif_sample": c["n_motif"]},
            "visualization": {"gen_fig_flag": False, "random_seed": 7, "n_max_iter": iters or c["iters"]}}
    # "default": the keys are absent (SEQ, the reference's arithmetic, at every N); "fast": opt-in FAST embedding; "exact": SEQ +
    # the reference's numpy calls at every size (np.argpartition neighbours / top-k, np.random.multinomial)
    if mode in ("seq", "fast"):
        over["visualization"]["embed_mode"] = mode
    if mode == "exact":
        over["general"] = {"exact": True}
    t0 = time.perf_counter()
    synth.write_res_dir(res, seq, borders, over)
    t["write_inputs_s"] = time.perf_counter() - t0
    del seq, borders
    sink = io.StringIO() if quiet else None
    try:
        with (contextlib.redirect_stdout(sink) if quiet else contextlib.nullcontext()):
            np.random.seed(123)
            t0 = time.perf_counter()
            md._scan_motif(str(res))
            t["scan_motif_s"] = time.perf_counter() - t0
            t0 = time.perf_counter()
            vz._visualize_kmers(str(res))
            t["visualize_kmers_s"] = time.perf_counter() - t0
        t["e2e_s"] = t["scan_motif_s"] + t["visualize_kmers_s"]
        finals = (res / "final_conseq.txt").read_text().split()
        rows = (res / "low_dim_data.tsv").read_text().splitlines()
        stages = dict(md.STAGE_TIMES)
        stages.update({"viz_" + k: v for k, v in vz.STAGE_TIMES.items()})
        return {"config": config, "mode": mode, **c, "k_range": [min_k, max_k], "final_conseq": finals,
                "n_embedded": len(rows) - 1, "times": t, "stages": stages, "res_dir": str(res) if keep else None}
    finally:
        if prev_mode is None:
            os.environ.pop("KMAP_EMBED_MODE", None)
        else:
            os.environ["KMAP_EMBED_MODE"] = prev_mode
        if not keep:
            shutil.rmtree(res, ignore_errors=True)


def run_e2e_dist(dist, config="C3", mode="default", min_k=6, max_k=9, iters=None, reads=None, shared_dir=None):
    """The two verbs on one clean res_dir under an initialised process group (every rank calls this): rank 0 writes the inputs
    (reads: its (seq, borders), or generated), all ranks then run `_scan_motif` (reads sharded) and `_visualize_kmers` (rows sharded)
    on it; the times are the slowest rank's.  shared_dir: where the res_dir is created (a path every rank sees; default: the system
    temp directory -- one node).  Returns the same dictionary as run_e2e on rank 0, None elsewhere."""
    import contextlib
    import io
    import torch
    from . import motif_discovery as md, synth, visualization as vz
    from .distributed import _coll_device, barrier as _dist_barrier
    assert mode in ("default", "seq", "fast", "exact"), mode
    c = CONFIGS[config]
    rank = dist.get_rank()
    prev_mode = os.environ.pop("KMAP_EMBED_MODE", None)
    box = [None]
    if rank == 0:
        seq, borders = reads if reads is not None else synth.synth_reads(c["n_reads"], c["read_len"], c["seed"])
        res = Path(tempfile.mkdtemp(prefix=f"kmap_{config}_dist_", dir=shared_dir))
        over = {"kmer_count": {"min_k": min_k, "max_k": max_k},
                "motif_discovery": {"motif_pos_density_flag": False, "motif_co_occurence_flag": False, "gen_hamball_flag": False,
                                    "n_total_sample": c["n_total"], "n_motif_sample": c["n_motif"]},
                "visualization": {"gen_fig_flag": False, "random_seed": 7, "n_max_iter": iters or c["iters"]}}
        if mode in ("seq", "fast"):
            over["visualization"]["embed_mode"] = mode
        if mode == "exact":
            over["general"] = {"exact": True}
        synth.write_res_dir(res, seq, borders, over)
        del seq, borders
        box = [str(res)]
    dist.broadcast_object_list(box, 0)
    res = Path(box[0])
    md.STAGE_TIMES.clear()
    vz.STAGE_TIMES.clear()

    def slowest(seconds):
        t = torch.tensor([seconds], dtype=torch.float64, device=_coll_device(dist))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    t = {}
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            np.random.seed(123)
            _dist_barrier(dist)
            t0 = time.perf_counter()
            md._scan_motif(str(res))              # ends with a barrier of its own
            t["scan_motif_s"] = slowest(time.perf_counter() - t0)
            t0 = time.perf_counter()
            vz._visualize_kmers(str(res))
            t["visualize_kmers_s"] = slowest(time.perf_counter() - t0)
        t["e2e_s"] = t["scan_motif_s"] + t["visualize_kmers_s"]
        if rank != 0:
            return None
        finals = (res / "final_conseq.txt").read_text().split()
        rows = (res / "low_dim_data.tsv").read_text().splitlines()
        stages = dict(md.STAGE_TIMES)
        stages.update({"viz_" + k: v for k, v in vz.STAGE_TIMES.items()})
        return {"config": config, "mode": mode, **c, "k_range": [min_k, max_k], "final_conseq": finals, "n_embedded": len(rows) - 1,
                "times": t, "stages": stages, "world": dist.get_world_size(), "res_dir": None}
    finally:
        if prev_mode is not None:
            os.environ["KMAP_EMBED_MODE"] = prev_mode
        _dist_barrier(dist)
        if rank == 0:
            shutil.rmtree(res, ignore_errors=True)
