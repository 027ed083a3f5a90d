#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_t5_gpu.py -q -x -s > gpurun_out/r3_t5.log 2>&1
rc=$?; echo "pytest rc=$rc"; grep -n "cosine\|passed\|failed\|Error" gpurun_out/r3_t5.log | head -20; tail -25 gpurun_out/r3_t5.log
