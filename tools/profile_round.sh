#!/bin/bash
# Round profile of the default bench command on the MI355X box (run from the repo root through gpurun):
#   1. rocprofv3 --kernel-trace --stats           -> gpurun_out/prof_<tag>/stats      (per-kernel time)
#   2. rocprofv3 --pmc FETCH_SIZE                 -> gpurun_out/prof_<tag>/fetch      (separate pass, TCC slots)
#   3. rocprofv3 --pmc WRITE_SIZE                 -> gpurun_out/prof_<tag>/write
#   4. rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES
# The program goes directly after `--` (no env / bash -c hop).  Usage: tools/profile_round.sh <tag> [bench args...]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 6 --warmup 2 --cpu-baseline-samples 0 --no-roofline --no-fewshot --no-extra-train $@"
# 0. un-profiled: the byte model of the HBM-bound ops of one step (input of round_profile_report.py's _hbm.md)
python3 $ROOT/bench.py --steps 3 --warmup 2 --cpu-baseline-samples 0 --no-fewshot --no-extra-train --hbm-bytes-out $OUT/hbm_bytes.json "$@" > $OUT/plain.json 2> $OUT/plain.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 $ROOT/bench.py --steps 10 --warmup 3 --cpu-baseline-samples 0 --no-roofline --no-fewshot --no-extra-train "$@" > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o run -- python3 $ROOT/bench.py $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o run -- python3 $ROOT/bench.py $ARGS > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --output-format csv -d $OUT/mfma -o run -- python3 $ROOT/bench.py $ARGS > $OUT/mfma.log 2>&1
ls -R $OUT | head -30
