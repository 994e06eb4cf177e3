#!/bin/bash
# Round 4, after the f16 x 2 format became the fused GRU forward's default: the whole GPU suite on the new library, the per-launch
# times of the three builds / modes on one box (default = hand-placed split; GGNN_GRU_FMT=3 = bf16 x 3; variant c = hipcc's split:
#   tools/variant_lib.sh c ggnn_gru_fused_split.hip -DGGNN_F16_SPLIT_ASM=0), and -- when the suite is green -- the round's profiles.
OUT=gpurun_out/${1:-fmt}; mkdir -p $OUT; export TMPDIR=/tmp
date +%s > $OUT/t0
timeout 420 python -m pytest tests -m gpu -x -q > $OUT/pytest_default.txt 2>&1; RC=$?
echo "rc=$RC" >> $OUT/pytest_default.txt
tail -4 $OUT/pytest_default.txt
run() { echo "== $*" >> $OUT/fwd.txt; env "$@" timeout 200 python tools/fwd_kernels.py >> $OUT/fwd.txt 2>&1; }
run A=0; run GGNN_GRU_FMT=3; run GGNN_LIB_VARIANT=c
echo "== default" >> $OUT/probe.txt; timeout 200 python tools/split_probe.py 2>&1 | tail -1 >> $OUT/probe.txt
echo "== variant c" >> $OUT/probe.txt; GGNN_LIB_VARIANT=c timeout 200 python tools/split_probe.py 2>&1 | tail -1 >> $OUT/probe.txt
grep -E "^==|^V =|one stream" $OUT/fwd.txt; python - <<PY
import json
L=[l for l in open("$OUT/probe.txt") if l.startswith("{")]
print("asm split == hipcc split (probe statistics identical):", len(L) == 2 and json.loads(L[0]) == json.loads(L[1]))
PY
date +%s > $OUT/t1
if [ $RC -eq 0 ]; then
    bash tools/profile_round.sh r04 bench train > $OUT/profile.log 2>&1; tail -30 $OUT/profile.log
else
    GGNN_LIB_VARIANT=c timeout 420 python -m pytest tests -m gpu -x -q > $OUT/pytest_c.txt 2>&1; tail -4 $OUT/pytest_c.txt
fi
date +%s > $OUT/t2
