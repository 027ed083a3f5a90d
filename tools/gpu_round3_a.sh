#!/bin/bash
# round-3 call A: new fp8 / fullsize bounds (printed statistics), then the restructured default bench
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_fp8_gpu.py tests/test_fullsize_gpu.py -q -s -k "fp8 or real_size" > gpurun_out/r3a_tests.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -5 gpurun_out/r3a_tests.log
if [ $rc -le 1 ]; then
  timeout -k 10 600 python bench.py > gpurun_out/r3a_bench.json 2> gpurun_out/r3a_bench.err
  echo "bench rc=$?"; tail -3 gpurun_out/r3a_bench.err; head -c 3000 gpurun_out/r3a_bench.json
fi
