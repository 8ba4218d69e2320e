"""ctypes binding of oracle/kmap_cpu_baseline.c -- the compiled OpenMP port of the kmap hot path that bench.py's
`cpu_baseline` leg TIMES on the GPU box's host cores.  Test / bench infrastructure only: nothing under
kmap_amd/ imports this module.  tests/test_oracle_golden.py pins every function here against the oracle."""
import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB = None
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


def lib():
    global _LIB
    if _LIB is None:
        so = _HERE / "libkmap_cpu_baseline.so"
        if not so.exists() or so.stat().st_mtime < (_HERE / "kmap_cpu_baseline.c").stat().st_mtime:
            subprocess.run(["make", "-s", "-C", str(_HERE), "libkmap_cpu_baseline.so"], check=True)
        L = C.CDLL(str(so))
        i64, i32, f64 = C.c_int64, C.c_int, C.c_double
        sig = {
            "kb_count": (None, [_u8p, i64, _i64p, i64, i32, i32, _u32p, i32]),
            "kb_merge_compact": (i64, [_u32p, i32, i32, _u32p, _i64p]),
            "kb_hamball_mass": (None, [_u32p, _i64p, i64, i32, _u32p, i32, i32, i32, _f64p, i32]),
            "kb_mask": (None, [_u8p, i64, i32, _u32p, _i32p, i32, i32]),
            "kb_find_motif": (i32, [_u8p, i64, _i64p, i64, i32, i32, f64, f64, i32, i32, i32, i32, _u32p, _f64p, C.POINTER(i64), i32]),
            "kb_fill_prob": (None, [_u32p, i64, _f32p, i32, _f32p, i32]),
            "kb_embed_forces": (f64, [_f32p, _f32p, i64, i64, i64, _f32p, i32]),
            "kb_embed_forces_src": (f64, [C.c_void_p, _u32p, _f32p, i32, _f32p, i64, i64, i64, _f32p, i32]),
            "kb_embed_forces_rows": (None, [_f32p, _i64p, i64, _f32p, i64, _f32p, i32]),
            "kb_set_threads": (None, [i32]),
            "kb_embed_update": (None, [_f32p, _f32p, i64, C.c_float]),
            "kb_max_threads": (i32, []),
        }
        for name, (res, args) in sig.items():
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        _LIB = L
    return _LIB


def host_cpu_info():
    """What the timed numbers ran on: CPU model (/proc/cpuinfo), logical CPUs of the host, CPUs this process may run on
    (affinity mask) and the cgroup CPU quota, if any; `usable` = the thread count the baseline uses for its all-core runs."""
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as fh:
            for ln in fh:
                if ln.lower().startswith("model name"):
                    model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    logical = os.cpu_count() or 1
    try:
        affinity = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        affinity = logical
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = Path(path).read_text().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
            break
        except (OSError, ValueError, IndexError):
            continue
    usable = affinity if quota is None else max(1, min(affinity, int(round(quota))))
    return {"model": model, "logical_cpus": logical, "affinity_cpus": affinity, "cgroup_cpu_quota": quota, "usable": usable}


def count(seq, borders, k, dedupe, merge_revcom=True, threads=0):
    """(uniq uint32, cnt int64) of the k-mers of the reads, k <= 13 -- a set: compare with the oracle as a dict"""
    hist = np.empty(4 ** k, np.uint32)
    borders = np.ascontiguousarray(borders, np.int64)
    lib().kb_count(np.ascontiguousarray(seq, np.uint8), len(seq), borders, len(borders), k, int(dedupe), hist, threads)
    u, c = np.empty(4 ** k, np.uint32), np.empty(4 ** k, np.int64)
    m = lib().kb_merge_compact(hist, k, int(merge_revcom), u, c)
    return u[:m].copy(), c[:m].copy()


def mask(seq, k, cons, radius, threads=0):
    cons = np.ascontiguousarray(cons, np.uint32)
    lib().kb_mask(seq, len(seq), k, cons, np.ascontiguousarray(radius, np.int32), len(cons), threads)
    return seq


def find_motif(seq, borders, k, mdef, top_k=5, n_trial=10, revcom_mode=True, rep_mode=False, threads=0):
    """{consensus hash: hamball proportion}; seq is masked in place"""
    cons, prop, nu = np.zeros(64, np.uint32), np.zeros(64, np.float64), C.c_int64(0)
    borders = np.ascontiguousarray(borders, np.int64)
    m = lib().kb_find_motif(seq, len(seq), borders, len(borders), k, int(mdef.max_ham_dist), float(mdef.p_uniform),
                            float(mdef.ratio_cutoff), top_k, n_trial, int(revcom_mode), int(rep_mode), cons, prop, C.byref(nu), threads)
    return {int(cons[i]): float(prop[i]) for i in range(m)}


def embed_forces(P, y, r0=0, r1=None, threads=0):
    """(raw gradient float32[2, n] -- rows outside [r0, r1) zero --, CE partial of those rows)"""
    n = P.shape[0]
    r1 = n if r1 is None else r1
    g = np.zeros((2, n), np.float32)
    loss = lib().kb_embed_forces(P, np.ascontiguousarray(y, np.float32), n, r0, r1, g, threads)
    return g, loss


def embed_forces_kmers(kh, lut, per_mismatch, y, r0=0, r1=None, threads=0):
    """embed_forces with p_ij = lut[per_mismatch * ham(kh_i, kh_j)] looked up on the fly (no N x N matrix in host memory)"""
    n = len(kh)
    r1 = n if r1 is None else r1
    g = np.zeros((2, n), np.float32)
    loss = lib().kb_embed_forces_src(None, np.ascontiguousarray(kh, np.uint32), np.ascontiguousarray(lut, np.float32), per_mismatch,
                                     np.ascontiguousarray(y, np.float32), n, r0, r1, g, threads)
    return g, loss


def embed_forces_rows(Pslab, rows, y, threads=0):
    """raw gradient float32[2, len(rows)] of the sampled rows; Pslab[r, :] = the probabilities of row rows[r] (no N x N matrix)"""
    rows = np.ascontiguousarray(rows, np.int64)
    Pslab = np.ascontiguousarray(Pslab, np.float32)
    n = Pslab.shape[1]
    assert Pslab.shape[0] == len(rows) and y.shape == (2, n)
    g = np.zeros((2, len(rows)), np.float32)
    lib().kb_embed_forces_rows(Pslab, rows, len(rows), np.ascontiguousarray(y, np.float32), n, g, threads)
    return g
