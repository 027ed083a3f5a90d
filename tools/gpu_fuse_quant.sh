#!/bin/bash
# fp8 row quantiser fused into LayerNorm forward / backward: tests, then the cfg5 A / B on one box (EAVQA_FUSE_QUANT=1 default, 0 = separate kernels)
cd ${GRAFT_REPO_ROOT:-.}
timeout -k 10 900 python -m pytest tests/test_fp8_gpu.py tests/test_ops_gpu.py -x -q -k "fp8 or layernorm" > gpurun_out/fuse_quant_tests.log 2>&1 || { tail -20 gpurun_out/fuse_quant_tests.log; exit 1; }
tail -2 gpurun_out/fuse_quant_tests.log
ARGS="--workload cfg5 --dtype fp8 --steps 20 --warmup 6 --cpu-baseline-samples 0 --no-roofline --no-fewshot --no-extra-train --no-t0"
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2; do
  for f in 1 0; do
    EAVQA_FUSE_QUANT=$f timeout -k 10 400 python3 bench.py $ARGS 2>/dev/null | line "fuse=$f run $i" || exit 1
  done
done
