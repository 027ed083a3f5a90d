#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -q -x -k "low_precision or layernorm" > gpurun_out/r3c_tests.log 2>&1
rc=$?; echo "pytest ops rc=$rc"; tail -3 gpurun_out/r3c_tests.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 900 python -m pytest tests/test_fullsize_gpu.py -q -s -k "clip_vit or cfg3" > gpurun_out/r3c_vit.log 2>&1
rc=$?; echo "pytest vit rc=$rc"; grep -n "image_embeds\|cfg3" gpurun_out/r3c_vit.log | head -20; tail -3 gpurun_out/r3c_vit.log
if [ $rc -gt 1 ]; then exit 1; fi
timeout -k 10 300 python tools/fewshot_profile.py > gpurun_out/r3c_fewshot.log 2>&1; echo "fewshot rc=$?"; cat gpurun_out/r3c_fewshot.log
