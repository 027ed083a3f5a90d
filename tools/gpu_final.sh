#!/bin/bash
# round-end check on one box:  tools/gpu_final.sh bench  (smoke + the default bench line)   |   tools/gpu_final.sh tests  (the whole GPU suite)
cd ${GRAFT_REPO_ROOT:-.}
if [ "$1" = "bench" ]; then
  python -c 'import __graft_entry__ as g; g.smoke()' > gpurun_out/final_smoke.log 2>&1 || { tail -5 gpurun_out/final_smoke.log; exit 1; }
  timeout -k 10 800 python bench.py > gpurun_out/final_bench_line.json 2> gpurun_out/final_bench_line.err || { tail -5 gpurun_out/final_bench_line.err; exit 1; }
  echo "bench done"
else
  timeout -k 10 1150 python -m pytest tests -x -q -m gpu > gpurun_out/final_gpu_tests.log 2>&1; rc=$?
  tail -5 gpurun_out/final_gpu_tests.log
  exit $rc
fi
