#!/bin/bash
# Profiling build of the attention kernels with phase timestamps (wall_clock64, 10 ns ticks) in attn_decode_kernel: a SEPARATE library,
# tools/_bin/libeavqa_attn_stamps.so, never shipped or loaded by the package.  tools/attn_stamps.py drives it.
set -e
cd "$(dirname "$0")/.."
C=explicit-alignment-for-vqa-tasks_amd/csrc
mkdir -p tools/_bin
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DEAVQA_ATTN_STAMPS -Iinclude -I$C -shared -o tools/_bin/libeavqa_attn_stamps.so $C/attention.hip $C/api.cpp -x hip
echo built tools/_bin/libeavqa_attn_stamps.so
