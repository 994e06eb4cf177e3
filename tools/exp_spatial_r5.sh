#!/bin/bash
# Round 5: spatial sharing of the CUs between the streams of the headline loop -- the fused GRU (form 1) with ONE 4-wave workgroup per CU
# (GGNN_GRU_WG_PER_CU=1), the transform with one 8-wave workgroup per CU (GGNN_K1_WG_PER_CU=1), 2 / 3 / 4 streams.
export TMPDIR=/tmp
b() { echo "== $*"; env "$@" GGNN_BENCH_CHILD=1 timeout 300 python bench.py --streams $S --no-secondary --no-cpu-baseline --no-roofline --min-time 0.8 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value'] / 1e9, 4), round(d['ms_per_step'], 4), d.get('ms_per_step_one_stream'))"; }
for S in 2 3 4; do b A=1; b GGNN_GRU_WG_PER_CU=1; b GGNN_GRU_WG_PER_CU=1 GGNN_K1_WG_PER_CU=1; done
S=2; b A=1
