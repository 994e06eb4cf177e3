#!/bin/bash
# Profiles of one round, run on the GPU box:  gpurun -- 'bash tools/profile_round.sh r01_final'
# Produces under gpurun_out/<tag>/ (copy what is to be judged into profiles/):
#   stats_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the bench command (kernel average durations)
#   {fetch,write,mfma}_counter_collection.csv   three SEPARATE --pmc passes (never combined with other trace domains)
#   pmc_summary.json         tools/pmc_summary.py over the three passes
#   bench.json               the plain bench line (no profiler attached)
set -u
TAG=${1:-round}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
# (the roofline leg stays on: it also launches the stand-alone scatter-add kernel, which the timed path fuses away)
BENCH="python $ROOT/bench.py --steps 12 --warmup 4 --streams 1 --no-cpu-baseline"
timeout 600 python "$ROOT/bench.py" > "$OUT/bench.json" 2> "$OUT/bench.err"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o stats -- $BENCH > "$OUT/stats.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT" -o fetch -- $BENCH > "$OUT/fetch.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT" -o write -- $BENCH > "$OUT/write.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT" -o mfma -- $BENCH > "$OUT/mfma.log" 2>&1
python "$ROOT/tools/pmc_summary.py" "$OUT" "$OUT/pmc_summary.json"
tail -1 "$OUT/bench.json"
head -12 "$OUT/stats_kernel_stats.csv"
# the raw traces are large; keep the per-dispatch counter files (small) and drop the traces
rm -f "$OUT"/*_kernel_trace.csv
