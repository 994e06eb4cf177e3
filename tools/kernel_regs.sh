#!/bin/bash
# Compile one .hip file for gfx950 and print VGPRs / scratch / occupancy per kernel (no GPU needed).
#   tools/kernel_regs.sh gated-graph-neural-network-samples_amd/csrc/ggnn_gru_fused.hip [grep filter] [extra hipcc flags...]
f=$1; pat=${2:-.}; shift; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$f" -o /tmp/kernel_regs.o -Rpass-analysis=kernel-resource-usage "$@" 2>&1 \
 | grep -E "error|Function Name|  VGPRs:|ScratchSize|Occupancy" \
 | sed -E 's/.*remark: +//; s/ \[-Rpass.*//; s/Function Name: //' | paste - - - - | c++filt \
 | sed -E 's/\([^\t]*\)//; s/void ggnn:://; s/ \[bytes\/lane\]//; s/ \[waves\/SIMD\]//' | grep -E "$pat"
