#!/bin/bash
# The R = 0 launch of the fused GRU in the f16 x 2 format under the GGNN_GRU_DBG ablation bits (1 MFMAs off, 2 | 4 epilogues off, 8 image
# DMA off, 16 fragment splits off; 30 = only MFMAs + fetches + barriers, 31 = only fetches / stores / barriers): DESIGN.md K3 (2) for this format.
OUT=gpurun_out/${1:-ablate_f16}; mkdir -p $OUT; export TMPDIR=/tmp
for b in 0 1 6 16 8 30 31; do echo "== GGNN_GRU_DBG=$b" >> $OUT/fwd.txt; GGNN_GRU_DBG=$b timeout 100 python tools/fwd_kernels.py 2>&1 | grep -E "^V =|one stream" >> $OUT/fwd.txt; done
cat $OUT/fwd.txt
