#!/bin/bash
# round-3 call B: resident attention + low-precision residual (tests, A/B bench, few-shot phases, cfg3)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -q -x -k "resident or low_precision or attention_forward_backward" > gpurun_out/r3b_tests.log 2>&1
rc=$?; echo "pytest ops rc=$rc"; tail -3 gpurun_out/r3b_tests.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 600 python -m pytest tests/test_fullsize_gpu.py -q -s -k "vit or clip or image" > gpurun_out/r3b_vit.log 2>&1
rc=$?; echo "pytest vit rc=$rc"; tail -3 gpurun_out/r3b_vit.log
if [ $rc -gt 1 ]; then exit 1; fi
timeout -k 10 300 python tools/attn_vit_bench.py > gpurun_out/r3b_attn.log 2>&1; echo "attn bench rc=$?"; cat gpurun_out/r3b_attn.log
timeout -k 10 300 python tools/fewshot_profile.py > gpurun_out/r3b_fewshot.log 2>&1; echo "fewshot rc=$?"; cat gpurun_out/r3b_fewshot.log
timeout -k 10 300 python bench.py --workload cfg3 --steps 10 --warmup 5 --no-fewshot --cpu-baseline-samples 0 > gpurun_out/r3b_cfg3.json 2> gpurun_out/r3b_cfg3.err; echo "cfg3 rc=$?"; python -c "
import json; d=json.loads(open('gpurun_out/r3b_cfg3.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['vit_tower_gemms'])"
