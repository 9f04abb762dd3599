#!/bin/bash
# usage: bash profiles/kernel_resources.sh <file.hip> [filter]   -- VGPR / spill / occupancy per kernel, from hipcc remarks
F=$1; PAT=${2:-.}
cd "$(dirname "$0")/../plonky2-lib_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden -DGLP_LDE_NO_HOIST -Rpass-analysis=kernel-resource-usage -c $F -o /tmp/_kr.o 2>&1 \
 | grep -E "Function Name|    VGPRs:|AGPRs:|ScratchSize|Occupancy|LDS Size" \
 | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g; s/^.*remark: *//' \
 | awk '/Function Name/{if(line)print line; line=$0; next}{line=line" | "$0}END{print line}' | sed 's/  */ /g' | grep -E "$PAT"
