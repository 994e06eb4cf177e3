#!/bin/bash
# Round-4 experiment: sustained bf16 MFMA rate under the power limit by operand data; PHASED form on all-zero data; product order
OUT=gpurun_out/${1:-exp_power}; mkdir -p $OUT
export TMPDIR=/tmp
( while true; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Socket|sclk" | sed -E 's/^GPU\[[0-9]+\][ \t]*: //' | tr '\n' '|'; echo; sleep 0.5; done ) > $OUT/smi.txt &
SMI=$!
tools/_bin/mfma_power_probe > $OUT/mfma_power_probe.txt 2>&1
kill $SMI
run() { echo "== $*" >> $OUT/fwd.txt; env "$@" python tools/fwd_kernels.py >> $OUT/fwd.txt 2>&1; }
run GGNN_FWD_DATA=zero
run GGNN_FWD_DATA=zero GGNN_GRU_FORM_R0=3
run GGNN_LIB_VARIANT=ord
run GGNN_LIB_VARIANT=ord GGNN_GRU_FORM_R0=3
run A=0
cat $OUT/mfma_power_probe.txt; grep -E "^==|^V =|one stream" $OUT/fwd.txt; awk 'NR%3==0' $OUT/smi.txt | head -30
