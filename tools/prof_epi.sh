#!/bin/bash
# kernel-only durations (rocprofv3 kernel trace) of single-K-step and real-K GEMMs: what is fixed cost, what is main loop
cd /tmp && export TMPDIR=/tmp
for k in 1 14 18 16 17; do
  rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_epi_$k -o run -- python3 $GRAFT_REPO_ROOT/tools/gemm_bench.py --k64 $k --iters 20 --only "epi,packed proj,packed qkv,packed fc2" > $GRAFT_REPO_ROOT/gpurun_out/prof_epi_$k.log 2>&1
done
