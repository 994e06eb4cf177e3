#!/bin/bash
# Round 5: the compacted message transform in the two-piece f16 operand format (selected per launch where proven) vs the exact bf16x3 one.
OUT=gpurun_out/${1:-edgefmt}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -8
run() { echo "== $*" >> $OUT/fwd.txt; env "$@" timeout 200 python tools/fwd_kernels.py >> $OUT/fwd.txt 2>&1; }
run A=1; run GGNN_GRU_FMT=3; run A=1; run GGNN_GRU_FMT=3
grep -E "^==|^V =|one stream" $OUT/fwd.txt
for x in auto 3; do echo "== bench GGNN_GRU_FMT=$x"; GGNN_GRU_FMT=$x GGNN_BENCH_CHILD=1 timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-roofline --min-time 1.0 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'] / 1e9, d['ms_per_step'], d.get('ms_per_step_one_stream'), d['operand_format'])"; done
