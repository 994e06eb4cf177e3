#!/bin/bash
OUT=gpurun_out/${1:-exp_misc}; mkdir -p $OUT
export TMPDIR=/tmp
run() { echo "== $*" >> $OUT/fwd.txt; env "$@" python tools/fwd_kernels.py >> $OUT/fwd.txt 2>&1; }
run A=0
run GGNN_LIB_VARIANT=r0 GGNN_GRU_FORM=0
run GGNN_GRU_FORM=0
for s in 2 3 4; do echo "== streams $s" >> $OUT/bench.txt; python bench.py --streams $s --no-secondary --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('ms_per_step_one_stream'), d['end_to_end_fresh_batch']['value'])" >> $OUT/bench.txt 2>&1; done
grep -E "^==|^V =|one stream" $OUT/fwd.txt; cat $OUT/bench.txt
