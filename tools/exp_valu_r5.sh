#!/bin/bash
# Round 5: vector-instruction count of the fused GRU forward (two-piece f16 format): variant libraries of ggnn_gru_fused_split.hip
#   pk    compiled WITH packed-f32 vector instructions (the tree builds it without: build.py NO_PACKED_F32)
#   nc    without the +-65504 clamp of the activations before the f16 split (GGNN_F16_CLAMP=0: 2 v_med3_f32 per value pair less)
#   pknc  both
OUT=gpurun_out/${1:-valu}; mkdir -p $OUT; export TMPDIR=/tmp
run() { echo "== $*" >> $OUT/fwd.txt; env "$@" timeout 200 python tools/fwd_kernels.py >> $OUT/fwd.txt 2>&1; }
for v in base pk nc pknc base pk nc pknc; do if [ $v = base ]; then run A=1; else run GGNN_LIB_VARIANT=$v; fi; done
grep -E "^==|^V =|one stream" $OUT/fwd.txt
