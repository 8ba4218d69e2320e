#!/usr/bin/env python3
"""the FAST sharded embedding loop under a G-rank gloo group sharing one GPU (where does the bench leg's time go?):
   KMAP_DIST_SAME_GPU=1 python -m torch.distributed.run --nproc-per-node 5 --master-addr 127.0.0.1 tools/probes/dist_fast_probe.py"""
import faulthandler
import os
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
faulthandler.dump_traceback_later(90, exit=True)
import torch
import torch.distributed as dist
from kmap_amd import visualization as V
from kmap_amd.distributed import kmap_from_kmers_distributed
torch.cuda.set_device(0)
dist.init_process_group("gloo")
rank = dist.get_rank()
rng = np.random.default_rng(3)
n = int(os.environ.get("PROBE_N", "50000"))
kh = np.sort(rng.integers(0, 4 ** 8, size=n, dtype=np.uint64))
lab = np.zeros(n, np.int64)
for mode, name in ((V.EMBED_FAST, "fast"), (V.EMBED_SEQ, "seq")):
    for it, prof, ac in ((24, 20, True), (100, 0, True), (24, 0, False)):
        dist.barrier()
        t0 = time.perf_counter()
        tr = {}
        kmap_from_kmers_distributed(kh, np.ones(n, np.int64), lab, ["ACGTACGT"], 8, n_max_iter=it, random_seed=7, trace=tr, mode=mode,
                                    always_collective=ac, profile_iters=prof)
        if rank == 0:
            print(f"{name} iters={it} prof={prof} always={ac}: total {time.perf_counter() - t0:.2f} s, loop {tr['loop_s']:.2f} s, phases {tr.get('phases')}", flush=True)
dist.destroy_process_group()
