#!/bin/bash
# round 4: T5 split-K decoder step - parity tests, then the T0_3B few-shot tool (phase times, step time per route) under rocprofv3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_t5_gpu.py -x -q > gpurun_out/r4_t5_test.log 2>&1; tail -3 gpurun_out/r4_t5_test.log
grep -q passed gpurun_out/r4_t5_test.log && ! grep -q failed gpurun_out/r4_t5_test.log || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace -d gpurun_out/prof_t0 -o t0 -- python tools/t0_fewshot_bench.py > gpurun_out/r4_t0_bench.log 2>&1
grep -v "simple_timer\|amdgpu.ids\|rocprofv3\|generateRocpd\|tool.cpp" gpurun_out/r4_t0_bench.log | tail -12
