import os, sys, pickle, shutil
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pathlib import Path
import numpy as np
from kmap_amd.e2e import run_e2e
r = run_e2e("C3", "fast", iters=1, keep=True)
with open(Path(r["res_dir"]) / "sample_kmers.pkl", "rb") as fh:
    skh, scnt, slab, sconseq = pickle.load(fh)
shutil.rmtree(r["res_dir"], ignore_errors=True)
scnt = np.asarray(scnt)
print("unique sampled k-mers", len(skh), "rows", int(scnt.sum()), "max count", int(scnt.max()), "rows that repeat their predecessor", int((scnt - 1).sum()))
