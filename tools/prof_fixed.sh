#!/bin/bash
# where the fixed cost of a specialised GEMM launch goes: rocprofv3 kernel durations with the timing-only ablations of eavqa_gemm_ex
cd /tmp && export TMPDIR=/tmp
for k in 16 14; do
  for a in 0 1 2 3 4; do
    rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_fixed_${k}_$a -o run -- python3 $GRAFT_REPO_ROOT/tools/gemm_bench.py --k64 $k --ablate $a --iters 20 --only "packed proj,packed qkv,packed fc2" > $GRAFT_REPO_ROOT/gpurun_out/prof_fixed_${k}_$a.log 2>&1
  done
done
