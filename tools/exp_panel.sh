#!/bin/bash
# End of round 4: the column-panel GRU (h = 128 / 192 / 256) in the two-piece f16 format.  Panel tests first; when they are green the
# A/B of config 5 against GGNN_GRU_FMT=3, the whole GPU suite and the round's profiles (all four legs) of these sources.
OUT=gpurun_out/${1:-panel}; mkdir -p $OUT; export TMPDIR=/tmp
date +%s > $OUT/t0
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -k "test_gru or large_graph or fullsize or full_size" > $OUT/pytest_panel.txt 2>&1; RC=$?
echo "rc=$RC" >> $OUT/pytest_panel.txt; tail -5 $OUT/pytest_panel.txt
echo "== default" >> $OUT/large.txt; timeout 200 python tools/bench_extra.py large 2>&1 | tail -1 | cut -c1-2500 >> $OUT/large.txt
echo "== GGNN_GRU_FMT=3" >> $OUT/large.txt; GGNN_GRU_FMT=3 timeout 200 python tools/bench_extra.py large 2>&1 | tail -1 | cut -c1-2500 >> $OUT/large.txt
cat $OUT/large.txt | cut -c1-700
date +%s > $OUT/t1
if [ $RC -eq 0 ]; then
    timeout 420 python -m pytest tests -m gpu -x -q > $OUT/pytest_default.txt 2>&1; RC2=$?
    echo "rc=$RC2" >> $OUT/pytest_default.txt; tail -4 $OUT/pytest_default.txt
    date +%s > $OUT/t2
    if [ $RC2 -eq 0 ]; then bash tools/profile_round.sh r04 bench large dense train > $OUT/profile.log 2>&1; tail -12 $OUT/profile.log | cut -c1-300; fi
fi
date +%s > $OUT/t3
