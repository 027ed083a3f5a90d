#!/bin/bash
# round 4: decode attention in registers - parity tests, phase stamps, few-shot profile (OPT-2.7B) and the T0-3B tool
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py tests/test_t5_gpu.py tests/test_model_gpu.py -x -q -k "decode or generate or t5 or vct0 or attention" > gpurun_out/r4_attn_test.log 2>&1; tail -3 gpurun_out/r4_attn_test.log
grep -q passed gpurun_out/r4_attn_test.log && ! grep -q failed gpurun_out/r4_attn_test.log || exit 1
timeout -k 10 200 python tools/attn_stamps.py > gpurun_out/r4_attn_stamps.log 2>&1; grep -v amdgpu gpurun_out/r4_attn_stamps.log
timeout -k 10 400 rocprofv3 --kernel-trace -d gpurun_out/prof_fs -o fs -- python tools/fewshot_profile.py > gpurun_out/r4_fewshot_profile.log 2>&1; grep -v "simple_timer\|amdgpu.ids\|rocprofv3\|generateRocpd\|tool.cpp" gpurun_out/r4_fewshot_profile.log | tail -8
timeout -k 10 400 python tools/t0_fewshot_bench.py > gpurun_out/r4_t0_bench2.log 2>&1; grep -v "amdgpu.ids\|Warning\|self.lm" gpurun_out/r4_t0_bench2.log | tail -9
