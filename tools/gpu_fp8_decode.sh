#!/bin/bash
# fp8 decode: parity tests, then the OPT-6.7B cached-generation figure (bf16 against e4m3 weights) with the kernel statistics of the fp8 run
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_fp8_gpu.py tests/test_ops_gpu.py -x -q -k "fp8 or greedy or splitk" > gpurun_out/r4_fp8_test.log 2>&1; tail -4 gpurun_out/r4_fp8_test.log
grep -q passed gpurun_out/r4_fp8_test.log && ! grep -q failed gpurun_out/r4_fp8_test.log || exit 1
timeout -k 10 500 python tools/fp8_decode_bench.py > gpurun_out/r4_fp8_decode.log 2>&1; grep -v "amdgpu.ids\|Warning" gpurun_out/r4_fp8_decode.log | tail -4
timeout -k 10 400 rocprofv3 --kernel-trace -d gpurun_out/prof_fp8dec -o d -- python tools/fp8_decode_bench.py --formats fp8 > gpurun_out/r4_fp8_decode_prof.log 2>&1
python tools/rocpd_stats.py gpurun_out/prof_fp8dec/d_results.db | head -14
