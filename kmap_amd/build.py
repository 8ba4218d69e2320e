"""Builds libkmap_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import subprocess
from pathlib import Path

_CSRC = Path(__file__).resolve().parent / "csrc"


def build(jobs=4, verbose=False):
    cmd = ["make", "-C", str(_CSRC), f"-j{jobs}"]
    r = subprocess.run(cmd, capture_output=not verbose, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"building libkmap_hip.so failed:\n{r.stdout}\n{r.stderr}")
    return _CSRC.parent / "libkmap_hip.so"
