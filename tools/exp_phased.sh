#!/bin/bash
# Round-4 experiment: the PHASED form (GGNN_GRU_FORM_R0=3) of the fused GRU -- parity suite on it, then per-kernel timings per variant.
OUT=gpurun_out/${1:-exp_phased}; mkdir -p $OUT
export TMPDIR=/tmp
run() { echo "== $*" >> $OUT/fwd.txt; env "$@" python tools/fwd_kernels.py >> $OUT/fwd.txt 2>&1; }
( GGNN_GRU_FORM_R0=3 timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $OUT/pytest_form3.txt
run A=0
run GGNN_GRU_FORM_R0=3
run GGNN_GRU_FORM_R0=3 GGNN_LIB_VARIANT=ec0
run GGNN_GRU_FORM_R0=3 GGNN_LIB_VARIANT=eu0
run A=0
run GGNN_GRU_FORM_R0=3
cat $OUT/pytest_form3.txt | tail -5; grep -E "^==|^V =|one stream" $OUT/fwd.txt
