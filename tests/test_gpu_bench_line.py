"""bench.py's one JSON line (the driver's contract): a short run on a 400-frame stack, every key the contract names, the
roofline and cpu_baseline objects, and the sanity relations a reader checks first."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], cwd=ROOT, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout[-2000:]                # ONE JSON line on stdout
    return json.loads(lines[0])


def test_default_line_on_a_short_stack():
    d = _bench("--frames", "400", "--steps", "3", "--warmup", "1", "--cpu", "on", "--cpu-frames", "24", "--host-batches", "0")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "frames/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "u8" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    # value = frames of the timed steps / their wall time
    assert abs(d["value"] - 400 / (d["ms_per_step"] * 1e-3)) <= 0.01 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["frac"] < 1
    # achieved = algorithmic bytes of a launch / its average duration (HIP events of the library, read after every call)
    assert abs(r["achieved"] - r["bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) <= 0.01 * r["achieved"]
    assert r["kernel"] == "label_map" and r["bytes_per_launch"] == 4 * 2560 * 2560 * 200
    assert r["traffic"] is None or r["traffic_source"]
    assert 0 < r["frac_alone"] < 1
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "frames/s" and c["value"] > 0 and c["cores"] >= 1 and c["sample"]
    g = d["roofline_groups"]
    assert set(g) >= {"bg_correct", "sparse_labelling", "label_map", "crops_props"}
    p = d["path"]
    assert p["contexts_consistent_after_timed_region"] is True and p["bytes_per_frame"] == 6 * 2560 * 2560
    assert abs(p["frac_of_hbm_peak_per_gpu"] - d["value"] * p["bytes_per_frame"] / 8e12) < 2e-3
    assert p["band_scan_fallbacks"] == 0 and p["fills_deferred"] >= 0 and p["deferred_fills_given_up"] <= p["fills_deferred"]


def test_frame_sharded_line_with_one_rank():
    """the N > 1 code path as far as one GPU goes (--force-dist): strong scaling over frame ranges, plain order of kernels"""
    d = _bench("--force-dist", "--frames", "600", "--steps", "2", "--warmup", "1", "--cpu", "off", "--host-batches", "0")
    assert d["scaling"] == "strong" and d["n_gpus"] == 1 and d["path"]["ranks_seen"] == 1
    assert d["path"]["fills_deferred"] == 0                  # bench.py turns deferred fills off for frame-sharded runs
    assert d["path"]["contexts_consistent_after_timed_region"] is True
