"""bench.py's output contract (one JSON line on stdout with the fields the round driver and the judge read), run as the
driver runs it: a child process, default workload (BASELINE config 2), few steps."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_fields():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2", "--cpu-sample-bs", "2"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["unit"] == "images/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 2
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "bf16" and d["data"] == "synthetic"
    assert "ViT-B/16" in d["metric"] and "bs=128" in d["config"]["workload"] and "model" not in d["config"]
    assert d["value"] > 1000 and abs(d["value"] - 128 / (d["ms_per_step"] * 1e-3)) < 0.01 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == 2500.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and 0.05 < rf["frac"] < 1.0
    assert abs(rf["achieved"] - 2 * 25216 * 768 * 3072 / (rf["kernel_ms"] * 1e-3) / 1e12) < 0.01 * rf["achieved"]   # algorithmic FLOP / in-place duration
    assert rf["traffic"] is None or rf["traffic"] > rf["algorithmic_bytes"] * 0.9
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "images/s" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
