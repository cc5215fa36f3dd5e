"""-m gpu: the two driver entry points as the driver runs them -- `python bench.py` (one JSON line with the contract's
keys) and `__graft_entry__.smoke()` -- each in its own process (bench.py imports torch before the library)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_contract_line(built):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "plate100k", "--steps", "2",
                        "--warmup", "1", "--cpu-sample-iters", "20", "--op-reps", "50"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["unit"] == "elements/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["value"] - d["config"]["elements"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and 0.0 < rf["frac"] < 1.0
    assert "traffic" in rf
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["unit"] == "elements/s" and cb["value"] > 0 and cb["sample"]
    assert d["value"] > 10.0 * cb["value"]          # north_star: >= 10x the CPU solver at 1 GPU


def test_smoke_entry_point(built):
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], cwd=ROOT, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
