"""bench.py's one-line JSON contract, checked on the GPU box with a short run of the default workload (a child process, like the driver's)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_roofline_and_cpu_baseline():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-fewshot", "--cpu-baseline-samples", "8"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    baseline = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert "train samples/sec" in baseline["metric"]                      # the half of BASELINE.json's metric this line carries
    assert d["metric"] == "mapper_train_samples_per_sec" and d["unit"] == "samples/s" and d["higher_is_better"] is True
    assert (d["n_gpus"], d["steps"], d["warmup"]) == (1, 3, 1) and d["scaling"] == "weak" and d["data"] == "synthetic" and d["dtype"] == "bf16"
    assert d["vs_baseline"] is None                                       # BASELINE.md publishes no number for this metric
    assert d["value"] > 0 and abs(d["value"] - d["config"]["global_batch"] / (d["ms_per_step"] * 1e-3)) <= 0.01 * d["value"]
    assert d["config"]["workload"].startswith("cfg2") and "model" not in d["config"]
    roof = d["roofline"]
    assert roof["bound"] == "mfma" and roof["unit"] == "TFLOP/s" and roof["peak"] == 2500.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3 and 0.05 < roof["frac"] < 1.0
    assert roof["launches_per_step"] > 100 and roof["algorithmic_bytes_per_launch"] > 0
    # traffic is either the stamped PMC figure of these very GEMM sources or explicitly dropped - never silently stale
    assert (roof["traffic"] is None) == (roof["traffic_source"] is None or roof["traffic_source"].startswith("dropped"))
    cpu = d["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["unit"] == "samples/s" and cpu["value"] > 0 and cpu["cores"] >= 1 and "sample" in cpu
