#!/bin/bash
# which tile runs the large-M CLIP-tower shapes fastest?  every specialised tile forced, against the dispatcher's pick (256 x 256 kernel)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
{
echo "=== auto"; python tools/gemm_bench.py --only "vitL qkv,vitL proj,vitL fc1,vitL fc2,vitL64,prefill"
for k in 4 5 6 9 10 11 12; do echo "=== --k64 $k"; python tools/gemm_bench.py --k64 $k --only "vitL qkv,vitL proj,vitL fc1,vitL fc2,vitL64,prefill" 2>/dev/null; done
} > gpurun_out/r3f_tiles.log 2>&1
cat gpurun_out/r3f_tiles.log | grep -v amdgpu.ids
