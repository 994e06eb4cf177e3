#!/bin/bash
# Round 5: transform-out (the GRU launch writes the next timestep's type-0 transformed rows) against the unfused path, one box.
OUT=gpurun_out/${1:-xf}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "transform_out or native_driver or sparse_model_matches or full_size or hip_graph or two_streams" 2>&1 | tail -15
run() { echo "== $*" >> $OUT/fwd.txt; env "$@" timeout 200 python tools/fwd_kernels.py >> $OUT/fwd.txt 2>&1; }
run GGNN_XF=0; run GGNN_XF=1; run GGNN_XF=0; run GGNN_XF=1
grep -E "^==|^V =|one stream" $OUT/fwd.txt
for x in 0 1; do echo "== bench GGNN_XF=$x"; GGNN_XF=$x GGNN_BENCH_CHILD=1 timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-roofline --min-time 1.0 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'] / 1e9, d['ms_per_step'], d.get('ms_per_step_one_stream'))"; done
