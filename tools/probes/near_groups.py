#!/usr/bin/env python3
"""In the pair-form steps (32 rows x 16 columns) of a converged C3 layout that are NOT all-far: how many of the step's four column-pair
groups (the columns {2d, 2d + 1, 8 + 2d, 9 + 2d} of the batch, what one packed instruction of the wave covers) are all-far, and how
many of its 16 columns / 32 rows hold a near pair -- would a far test per group pay?"""
import shutil
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from kmap_amd.e2e import run_e2e   # noqa: E402

r = run_e2e("C3", "fast", keep=True)
tab = np.loadtxt(Path(r["res_dir"]) / "low_dim_data.tsv", skiprows=1)
shutil.rmtree(r["res_dir"], ignore_errors=True)
x, y = tab[:, 0].astype(np.float32), tab[:, 1].astype(np.float32)
n = len(x)
rng = np.random.default_rng(2)
near_steps = 0
groups_far = np.zeros(5, int)
cols_near, rows_near, lanes_near = [], [], []
trials = 40000
for _ in range(trials):
    i0 = int(rng.integers(0, (n - 32) // 32)) * 32
    j0 = int(rng.integers(0, (n - 16) // 16)) * 16
    dx = x[i0:i0 + 32, None] - x[None, j0:j0 + 16]
    dy = y[i0:i0 + 32, None] - y[None, j0:j0 + 16]
    near = (dx * dx + dy * dy) < 1000.0
    if not near.any():
        continue
    near_steps += 1
    g = sum(not near[:, [2 * d, 2 * d + 1, 8 + 2 * d, 9 + 2 * d]].any() for d in range(4))
    groups_far[g] += 1
    cols_near.append(near.any(axis=0).sum())
    rows_near.append(near.any(axis=1).sum())
    lanes_near.append((near[:, :8].any(axis=1).sum() + near[:, 8:].any(axis=1).sum()))
print(f"{near_steps / trials:.3f} of the sampled steps have a near pair")
print("all-far groups per such step (0..4):", (groups_far / max(near_steps, 1)).round(3), "mean", (groups_far * np.arange(5)).sum() / max(near_steps, 1))
print("columns with a near pair: mean", np.mean(cols_near), "median", np.median(cols_near), " rows with a near pair: mean", np.mean(rows_near), "median", np.median(rows_near))
print("lanes (of 64) with a near pair: mean", np.mean(lanes_near), "median", np.median(lanes_near))
