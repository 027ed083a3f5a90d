#!/bin/bash
# A / B of two checkouts on the same box: the tree itself and a second checkout under _ab_old/ (git worktree of the previous commit, built)
cd ${GRAFT_REPO_ROOT:-.}
W=${1:-cfg2}
timeout -k 10 600 python -m pytest tests/test_gemm_ln_gpu.py tests/test_model_gpu.py -x -q > gpurun_out/ln_fold_tests.log 2>&1 || { tail -20 gpurun_out/ln_fold_tests.log; exit 1; }
tail -2 gpurun_out/ln_fold_tests.log
ARGS="--workload $W --steps 40 --warmup 10 --cpu-baseline-samples 0 --no-roofline --no-fewshot --no-extra-train --no-t0"
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2; do
  (cd _ab_old && timeout -k 10 300 python3 bench.py $ARGS 2>/dev/null | line "old run $i") || exit 1
  EAVQA_LN_FOLD=0 timeout -k 10 300 python3 bench.py $ARGS 2>/dev/null | line "new fold=0 run $i" || exit 1
  EAVQA_LN_FOLD=1 timeout -k 10 300 python3 bench.py $ARGS 2>/dev/null | line "new fold=1 run $i" || exit 1
done
