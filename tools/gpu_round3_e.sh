#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_ops_gpu.py tests/test_fp8_gpu.py tests/test_model_gpu.py -q -x > gpurun_out/r3e_tests.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/r3e_tests.log
if [ $rc -ne 0 ]; then exit 1; fi
{
python tools/gemm_bench.py --big 2 --epi quick --only "vitL fc1"
python tools/gemm_bench.py --epi fc1 --only "packed fc1"; python tools/gemm_bench.py --epi bwd --only "packed fc1"; python tools/gemm_bench.py --only "packed fc1"
} > gpurun_out/r3e_bench.log 2>&1; cat gpurun_out/r3e_bench.log
timeout -k 10 600 python bench.py --cpu-baseline-samples 0 > gpurun_out/r3e_bench.json 2> gpurun_out/r3e_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3e_bench.json').read().strip().splitlines()[-1])
print("cfg2", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["gemm_ms_per_step"])
for e in d["extra"]:
    r=e["roofline"]
    if isinstance(r,list): print(e["metric"], e["value"], [(x["phase"], x["ms"], x["frac"]) for x in r])
    else: print(e["config"]["workload"][:5], e["value"], e["ms_per_step"], r["frac"], r["gemm_ms_per_step"], "vit", r["vit_tower_gemms"]["gemm_ms_per_step"], r["vit_tower_gemms"]["frac"])
PY
