#!/bin/bash
# Round 5: config 5 (100k nodes / 1M edges / h = 256) with the ring transform in the two-piece f16 operand format vs the exact one.
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "compact_transform or native_driver or large_graph or fullsize or full_size or random_model_shapes_any" 2>&1 | tail -4
for f in auto 3 auto 3; do echo "== GGNN_GRU_FMT=$f"; GGNN_GRU_FMT=$f timeout 300 python tools/bench_extra.py large 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k = d.get('kernels', {}); print(d.get('ms_per_step'), {n: round(v['avg_us'], 1) for n, v in k.items()})"; done
