#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout -k 10 600 python -m pytest tests/test_model_gpu.py tests/test_bench_contract_gpu.py -x -q > gpurun_out/fs_ahead_tests.log 2>&1 || { tail -20 gpurun_out/fs_ahead_tests.log; exit 1; }
tail -2 gpurun_out/fs_ahead_tests.log
timeout -k 10 900 python bench.py --steps 10 --warmup 5 --cpu-baseline-samples 0 --no-roofline --no-extra-train > gpurun_out/fs_ahead_on.json 2> gpurun_out/fs_ahead_on.err || { tail -5 gpurun_out/fs_ahead_on.err; exit 1; }
timeout -k 10 900 python bench.py --steps 10 --warmup 5 --cpu-baseline-samples 0 --no-roofline --no-extra-train --no-overlap > gpurun_out/fs_ahead_off.json 2> gpurun_out/fs_ahead_off.err || exit 1
python3 - <<'PY'
import json
for n in ("on", "off"):
    d = json.loads(open(f"gpurun_out/fs_ahead_{n}.json").read().strip().splitlines()[-1])
    print(n, "cfg2", d["value"], [(e["config"]["workload"][:14], e["value"], e.get("ms_per_batch") or e.get("ms_per_step")) for e in d["extra"]])
PY
