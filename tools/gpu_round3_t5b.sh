#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_t5_gpu.py -q -x -k "executors" > gpurun_out/r3_t5b.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -15 gpurun_out/r3_t5b.log
timeout -k 10 1100 python -m pytest tests/test_fullsize_gpu.py -q -x -s -k "t0_3b" > gpurun_out/r3_t5c.log 2>&1
rc=$?; echo "pytest t0_3b rc=$rc"; grep -n "T0_3B" gpurun_out/r3_t5c.log; tail -15 gpurun_out/r3_t5c.log
