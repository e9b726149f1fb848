#!/bin/bash
# Registers, spills and LDS of the decode kernels of the working tree (or of the tree given as $1).
# bash tools/regs.sh [dir with tansu_amd/csrc and include] [name filter]
root=${1:-$(dirname "$0")/..}
filter=${2:-decode_fused\|chunk_kernel\|second_pass\|decode_after\|batch_header}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $NISSHI_CXXFLAGS -Rpass-analysis=kernel-resource-usage \
  -I "$root/include" -I "$root/tansu_amd/csrc" -o /tmp/regs_probe.so "$root/tansu_amd/csrc/nisshi_hip.hip" 2>&1 |
  grep -A12 "Function Name: _ZN6nisshi[0-9]*\($filter\)" |
  grep -E "Function Name|VGPRs:|Spill|ScratchSize|LDS Size" |
  sed -E 's/.*remark: +//; s/ \[-Rpass.*//' |
  awk '/Function Name/ {if (line) print line; line=$3} !/Function Name/ {line=line "  " $0} END {print line}' | cut -c1-250
