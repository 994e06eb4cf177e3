#!/bin/bash
OUT=gpurun_out/${1:-exp_tl}; mkdir -p $OUT
export TMPDIR=/tmp
GGNN_LIB_VARIANT=tl python tools/gru_gather_timeline.py > $OUT/gtl_form0.txt 2>&1
GGNN_LIB_VARIANT=tl GGNN_GRU_FORM_R0=3 python tools/gru_gather_timeline.py > $OUT/gtl_form3.txt 2>&1
cat $OUT/gtl_form0.txt; echo ======; cat $OUT/gtl_form3.txt
