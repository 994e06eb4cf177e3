#!/bin/bash
# Round 5: the graph-resident dense forward (BASELINE configs[2]) in the two-piece f16 operand format vs the exact bf16x3 one.
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_dense.py -m gpu -x -q 2>&1 | tail -4
for f in auto 3 auto 3; do echo "== GGNN_GRU_FMT=$f"; GGNN_GRU_FMT=$f timeout 200 python tools/bench_extra.py dense 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k = d.get('kernels', {}); print(d.get('ms_per_step'), {n: round(v['avg_us'], 1) for n, v in k.items()})"; done
