#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -q -x -k "attention or packed or model or training or forward" > gpurun_out/r3h_tests.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/r3h_tests.log
if [ $rc -ne 0 ]; then exit 1; fi
python tools/attn_bench.py 2>/dev/null | tee gpurun_out/r3h_attn.log
timeout -k 10 300 python bench.py --no-fewshot --no-extra-train --cpu-baseline-samples 0 > gpurun_out/r3h_cfg2.json 2> gpurun_out/r3h_cfg2.err; python -c "
import json; d=json.loads(open('gpurun_out/r3h_cfg2.json').read().strip().splitlines()[-1]); print('cfg2', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['gemm_ms_per_step'])"
