#!/bin/bash
# act-in-registers check: parity first, then timing
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_fp8_gpu.py tests/test_model_gpu.py -q -x > gpurun_out/i_tests.log 2>&1 || { tail -30 gpurun_out/i_tests.log; exit 1; }
tail -3 gpurun_out/i_tests.log
timeout -k 10 200 python tools/gemm_bench.py --big 2 --epi quick --only "vitL fc1" > gpurun_out/i_gemm.log 2>&1 && \
timeout -k 10 200 python tools/gemm_bench.py --epi fc1 --only "packed fc1" >> gpurun_out/i_gemm.log 2>&1 && \
timeout -k 10 300 python tools/fewshot_profile.py > gpurun_out/i_fewshot.log 2>&1
grep -v amdgpu.ids gpurun_out/i_gemm.log | tail -30
grep -v amdgpu.ids gpurun_out/i_fewshot.log | tail -20
