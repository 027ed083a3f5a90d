#!/bin/bash
# pipelined AdamW / armed exchange: tests, the 2-rank bench rehearsal on one card, then cfg2 / cfg5 with and without the pipelined update
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 700 python -m pytest tests/test_model_gpu.py tests/test_bench_contract_gpu.py tests/test_dp_factor_gpu.py tests/test_rccl_gpu.py tests/test_executor_gpu.py -x -q > gpurun_out/r4_upd_test.log 2>&1; tail -4 gpurun_out/r4_upd_test.log
grep -q passed gpurun_out/r4_upd_test.log && ! grep -q failed gpurun_out/r4_upd_test.log || exit 1
for a in "" "--pipelined-update"; do
  timeout -k 10 300 python bench.py --no-fewshot --no-extra-train --no-t0 --cpu-baseline-samples 0 --no-roofline $a > gpurun_out/r4_upd_cfg2.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/r4_upd_cfg2.json').read().strip().splitlines()[-1]); print('cfg2 [$a]', d['value'], d['ms_per_step'])"
  timeout -k 10 300 python bench.py --workload cfg5 --dtype fp8 --steps 10 --no-fewshot --no-extra-train --no-t0 --cpu-baseline-samples 0 --no-roofline $a > gpurun_out/r4_upd_cfg5.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/r4_upd_cfg5.json').read().strip().splitlines()[-1]); print('cfg5 [$a]', d['value'], d['ms_per_step'])"
done
