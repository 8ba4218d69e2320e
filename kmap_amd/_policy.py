"""Run-wide choices between the reference's own host calls / arithmetic and the device rules that replace them at scale.

The reference has ONE arithmetic for the embedding (visualization.py:296-317, taichi_core.py:305-326) and three numpy calls whose
results depend on numpy's tie / draw order (np.argpartition in knn_smooth :100 and find_motif motif_discovery.py:661,
np.random.multinomial in sample_disp_kmer :912).  The package follows them by default where that is affordable and leaves them
above documented sizes; this module is the one place where a user pins them:

* `visualization.embed_mode = "seq" | "fast"` in config.toml (absent = "seq" at every N), overridden by KMAP_EMBED_MODE.
* `general.exact = true` in config.toml, or KMAP_EXACT=1: np.argpartition top-k, np.random.multinomial and the numpy neighbour
  choice at every size, whatever the thresholds say (the strict drop-in run; the cost at C3 is `e2e.k6_9.exact` of bench.py).
"""
import os

_cfg = {"exact": False, "embed_mode": None}


def apply_config(config_dict):
    """called by `_scan_motif` / `_visualize_kmers` after they parsed config.toml; both keys are optional"""
    _cfg["exact"] = bool(config_dict.get("general", {}).get("exact", False))
    mode = config_dict.get("visualization", {}).get("embed_mode")
    if mode is not None and str(mode).lower() not in ("seq", "fast"):
        raise ValueError(f'config.toml: visualization.embed_mode must be "seq" or "fast", not {mode!r}')
    _cfg["embed_mode"] = None if mode is None else str(mode).lower()


def reset():
    _cfg["exact"], _cfg["embed_mode"] = False, None


def exact():
    """True: take the reference's numpy call at every size (KMAP_EXACT=1|0 wins over config.toml's general.exact)"""
    forced = os.environ.get("KMAP_EXACT", "")
    if forced in ("0", "1"):
        return forced == "1"
    return _cfg["exact"]


def embed_mode():
    """'seq' | 'fast': KMAP_EMBED_MODE, then config.toml's visualization.embed_mode, then 'seq'"""
    forced = os.environ.get("KMAP_EMBED_MODE", "").lower()
    if forced in ("seq", "fast"):
        return forced
    return _cfg["embed_mode"] or "seq"
