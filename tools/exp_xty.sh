#!/bin/bash
OUT=gpurun_out/${1:-exp_xty}; mkdir -p $OUT
export TMPDIR=/tmp
( python -m pytest tests/test_gpu_parity.py -x -q -k "xty" 2>&1 | tail -5 ) > $OUT/pytest.txt
for v in 1 0; do echo "== GGNN_XTY_PLANES=$v" >> $OUT/bench.txt; GGNN_XTY_PLANES=$v python tools/xty_bench.py 2>&1 | grep -v amdgpu >> $OUT/bench.txt; GGNN_XTY_PLANES=$v python tools/bench_extra.py train 2>/dev/null | tail -1 | cut -c1-170 >> $OUT/bench.txt; done
cat $OUT/pytest.txt $OUT/bench.txt
