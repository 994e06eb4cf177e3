#!/bin/bash
# Round 5 (2): exact-three-round batches (no tail round) under forms 0 / 1 / 5; the headline's stream count.
OUT=gpurun_out/${1:-forms2}; mkdir -p $OUT; export TMPDIR=/tmp
run() { echo "== $*" >> $OUT/fwd.txt; env "$@" timeout 200 python tools/fwd_kernels.py >> $OUT/fwd.txt 2>&1; }
for f in 0 1 5; do run GGNN_GRU_FORM=$f GGNN_FWD_BATCH_NODES=98304; run GGNN_GRU_FORM=$f; done
grep -E "^==|^V =|one stream" $OUT/fwd.txt
for s in 2 3 4; do for f in 0 1; do
  echo "== streams $s form $f"; GGNN_GRU_FORM=$f GGNN_BENCH_CHILD=1 timeout 300 python bench.py --streams $s --no-secondary --no-cpu-baseline --no-roofline --min-time 1.0 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'] / 1e9, d['ms_per_step'], d.get('ms_per_step_one_stream'))"
done; done
