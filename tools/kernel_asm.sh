#!/bin/bash
# Instruction skeleton (waits, loads, MFMAs, branches, barriers) of one kernel:  tools/kernel_asm.sh decode_direct.hip ILi2ELi1ELb0ELb1E [regex]
cd "$(dirname "$0")/../explicit-alignment-for-vqa-tasks_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -S --cuda-device-only "$1" -o /tmp/kernel_asm.s 2>/dev/null || exit 1
pat="${3:-s_waitcnt|buffer_load|global_load|v_mfma|s_cbranch|^\.LBB|s_barrier|ds_read|ds_write|buffer_store|global_store}"
awk -v k="$2" '$0 ~ "^_Z.*" k ".*: *;" {on=1} on {print} on && /s_endpgm/ {exit}' /tmp/kernel_asm.s | grep -E "$pat" | awk '{print $1, $2}' | uniq -c
