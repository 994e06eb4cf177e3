#!/bin/bash
# Round 5: the training step with the 16-byte partial-product reduction (xty_reduce4_kernel) against the scalar one.
OUT=gpurun_out/${1:-train}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_parity.py -m gpu -x -q -k "xty or gradients or native or train_step or variant or reduces_loss or side_stream" 2>&1 | tail -5
for e in 1 0 1 0; do echo "== GGNN_XTY_REDUCE_SCALAR=$e"; GGNN_XTY_REDUCE_SCALAR=$e timeout 300 python tools/bench_extra.py train 2>/dev/null | tail -1 | cut -c1-200; done
cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT -o train -- python $OLDPWD/tools/bench_extra.py train > $OLDPWD/$OUT/train_stats.log 2>&1; cd $OLDPWD
python - <<PY
import csv, re
rows = list(csv.DictReader(open("$OUT/train_kernel_stats.csv")))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:14]:
    n = r['Name']; m = re.search(r'ggnn::(\w+)', n)
    print((m.group(1) if m else n[:40]).ljust(36), r['Calls'].rjust(6), '%8.1f us' % (float(r['AverageNs']) / 1e3), '%5.1f%%' % (100 * float(r['TotalDurationNs']) / tot))
PY
rm -f $OUT/*_kernel_trace.csv
