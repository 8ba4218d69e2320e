"""bench.py as the driver runs it, on the GPU: the ONE stdout line must parse (round 5's 25-KB line did not)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_bench_prints_one_compact_line_and_a_detail_file(tmp_path):
    """`python3 bench.py --gpus 1 --steps 3 --warmup 1` with the long legs switched off (the C3 pipeline, the headline, `stages`, `c5`
    and a quick `cpu_baseline` still run): stdout is exactly one line of strict JSON <= 4096 bytes with the contract keys, `roofline`
    and `cpu_baseline`; the detail file it names exists, is strict JSON and holds the legs; the headline agrees with its own numbers."""
    env = dict(os.environ, PYTHONPATH=str(ROOT))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--quick", "--no-c4", "--no-embed-dist",
           "--shard-proxy", "0", "--e2e", "none", "--time-budget", "200"]
    r = subprocess.run(cmd, env=env, cwd=str(ROOT), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    assert len(lines[0].encode()) <= 4096

    def no_const(c):
        raise AssertionError(f"non-finite constant {c} in the line")
    d = json.loads(lines[0], parse_constant=no_const)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "cpu_baseline", "ranks_seen", "distinct_gpus", "detail"):
        assert key in d, key
    assert d["metric"] == "hamming_pairs_per_s" and d["unit"] == "pairs/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["dtype"] == "u8" and d["scaling"] == "weak" and d["vs_baseline"] is None and d["higher_is_better"] is True
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and 0.3 < rf["frac"] < 1.0
    assert abs(rf["achieved"] - rf["algorithmic_bytes"] / (rf["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * rf["achieved"]
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    n = d["config"]["n_kmers"]
    assert n == 50000 and abs(d["value"] - n * n / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "pairs/s" and cb["cores"] >= 1 and cb["value"] > 0
    det = json.loads((ROOT / d["detail"]).read_text(), parse_constant=no_const)
    assert det["metric"] == d["metric"] and "stages" in det["roofline"] and "c5" in det and det["leg_errors"] == []
    assert {"count_pass_k8", "count_pass_k14", "count_k14_keyspace_rank", "scan_k8_r2", "embed_iter_seq"} <= set(det["roofline"]["stages"])
    assert 0.3 < det["c5"]["frac"] < 1.0 and "oracle" in det["c5"]["spot_check"]
