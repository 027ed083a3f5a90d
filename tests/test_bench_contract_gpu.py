"""bench.py's one-line JSON contract, checked on the GPU box with a short run of the default workload (a child process, like the driver's)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_roofline_and_cpu_baseline():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--cpu-baseline-samples", "8"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]      # stdout is the JSON line and nothing else
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
    # the dominant kernel cannot take longer than the step: the roofline pass brackets the main stream's GEMMs with the CLIP
    # encode serialised, so no bracket holds another stream's kernels (round 2's cfg3 line did: 43.8 ms of GEMM in a 30.6 ms step)
    assert roof["gemm_ms_per_step"] <= d["ms_per_step"] and roof["gemm_ms_within_step"] is True
    assert roof["vit_tower_gemms"]["launches_per_step"] > 0
    ops_seen = {h["op"] for h in roof["hbm_kernels"]}
    assert {"eavqa_layernorm_fwd", "eavqa_layernorm_bwd", "eavqa_ce_fwd", "eavqa_ce_bwd", "eavqa_adamw"} <= ops_seen
    for h in roof["hbm_kernels"]:
        assert h["unit"] == "GB/s" and h["peak"] == 8000.0 and 0 < h["frac"] < 1.0 and abs(h["frac"] - h["achieved"] / 8000.0) < 1e-3
    # "extra": every other 1-GPU BASELINE config, each with its own self-consistent roofline
    extra = d["extra"]
    assert isinstance(extra, list) and all("error" not in e for e in extra), extra
    legs = {(e["metric"], e["config"]["workload"][:4], e.get("dtype")) for e in extra}
    assert ("fewshot_vqa_questions_per_sec", "few-", None) in legs
    assert ("mapper_train_samples_per_sec", "cfg3", "bf16") in legs and ("mapper_train_samples_per_sec", "cfg5", "fp8") in legs
    for e in extra:
        if e["metric"] != "mapper_train_samples_per_sec":
            continue
        r3 = e["roofline"]
        assert e["value"] > 0 and r3["gemm_ms_per_step"] <= e["ms_per_step"], (e["config"]["workload"], r3["gemm_ms_per_step"], e["ms_per_step"])
        assert abs(r3["frac"] - r3["achieved"] / r3["peak"]) < 1e-3 and r3["peak"] == (5000.0 if e["dtype"] == "fp8" else 2500.0)
        assert r3["vit_tower_gemms"]["gemm_ms_per_step"] > 0


def test_bench_two_ranks_rehearsal_prints_one_line_with_the_exchange_fields():
    """The N > 1 control flow of bench.py end to end - launched as the driver launches it (torch.distributed.run, one process per rank) - on
    this one-GPU box: both ranks on the card, gloo instead of RCCL (RCCL refuses two ranks on one device).  The numbers mean nothing; the
    line must be the only thing on stdout (gloo prints a connection notice there) and carry the distributed fields."""
    env = dict(os.environ, EAVQA_DIST_BACKEND="gloo", EAVQA_FORCE_DEVICE="0")
    for exchange in ("auto", "sharded"):
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                            "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                            "--cpu-baseline-samples", "0", "--no-roofline", "--dp-exchange", exchange],
                           capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.strip()]
        assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]
        d = json.loads(lines[0])
        c = d["config"]
        assert d["n_gpus"] == 2 and c["global_batch"] == 128 and c["parallelism"] == "dp2" and c["world_size"] == 2 and c["backend"] == "gloo"
        assert set(c["dp_exchange_model_ms_unmeasured"]) == {"allreduce", "sharded", "factors"}
        assert c["dp_exchange"].startswith("mapper gradient factors" if exchange == "auto" else "reduce-scatter")
        assert d["extra"] is None and d["cpu_baseline"] is None and d["value"] > 0           # secondary legs belong to the N = 1 run
        assert abs(d["value"] - 128 / (d["ms_per_step"] * 1e-3)) <= 0.01 * d["value"]
