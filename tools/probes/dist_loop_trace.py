#!/usr/bin/env python3
"""The sharded embedding loop on a one-rank RCCL group (what bench.py's embed_dist leg runs at G = 1), for kernel traces:
which kernels does an iteration of the multi-GPU path launch, and how long do they take next to the resident loop's?"""
import os
import socket
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))


def main():
    import torch
    import torch.distributed as dist
    from kmap_amd import visualization as V
    from kmap_amd.distributed import kmap_from_kmers_distributed
    n, k = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000, 8
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    rng = np.random.default_rng(2)
    kh = rng.integers(0, 4 ** k, size=n, dtype=np.uint64).astype(np.uint32)
    lab = np.zeros(n, np.int64)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    tr = {}
    kmap_from_kmers_distributed(kh, np.ones(n, np.int64), lab, ["ACGTACGT"], k, n_max_iter=iters, random_seed=7, mode=V.EMBED_FAST, trace=tr,
                                always_collective=True)
    print(f"sharded loop: {tr['loop_s'] / iters * 1e3:.4f} ms / iteration")
    tr = {}
    V.kmap_from_kmers(kh, np.ones(n, np.int64), lab, ["ACGTACGT"], k, n_max_iter=iters, random_seed=7, mode=V.EMBED_FAST, trace=tr)
    print(f"resident loop: {tr['loop_s'] / iters * 1e3:.4f} ms / iteration")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
