#!/bin/bash
# Register / spill / LDS summary of every kernel in one csrc file:  tools/kernel_regs.sh decode_direct.hip [filter]
cd "$(dirname "$0")/../explicit-alignment-for-vqa-tasks_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -c "$1" -o /tmp/kernel_regs.o \
    -Rpass-analysis=kernel-resource-usage 2>&1 \
  | grep -E "Function Name|VGPRs:|AGPRs:|ScratchSize|VGPRs Spill|LDS Size" \
  | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' \
  | awk '/Function Name/ {if (line) print line; line=$3; next} {gsub(/^ +/, ""); line=line "  " $0} END {print line}' \
  | while read -r l; do n=$(echo "$l" | cut -d' ' -f1 | c++filt | sed -E 's/\(anonymous namespace\):://; s/\(.*//'); echo "$n $(echo "$l" | cut -d' ' -f2-)"; done \
  | grep -E "${2:-.}"
