#!/bin/bash
# Profiles of one round, run on the GPU box:  gpurun -- 'bash tools/profile_round.sh r02 [bench|large|dense|train ...]'
# Produces under gpurun_out/<tag>/ (copy what is to be judged into profiles/ as <tag>_*):
#   bench.json                       the plain bench line (no profiler attached)
#   stats_kernel_stats.csv           rocprofv3 --kernel-trace --stats of the bench command (kernel average durations)
#   {fetch,write,mfma}_counter_collection.csv   three SEPARATE --pmc passes (never combined with other trace domains)
#   pmc_summary.json                 tools/pmc_summary.py over the three passes
#   config5_* / config3_*            the same four passes + summary for `tools/bench_extra.py large` / `dense`
#   train_kernel_stats.csv           --stats of `tools/bench_extra.py train`
set -u
TAG=${1:-round}; shift || true
LEGS=${*:-bench large dense train}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
# the library that travelled with the snapshot must be the one these sources build (a profile of a stale binary names the wrong tree)
python - <<PY || { echo "libggnn_hip.so is older than its sources: run __graft_entry__.build() before profiling"; exit 1; }
import importlib, sys
sys.path.insert(0, "$ROOT")
b = importlib.import_module("gated-graph-neural-network-samples_amd.build")
sys.exit(1 if b.needs_build() else 0)
PY
cd /tmp
four_passes() {   # four_passes <prefix> <command...>
    local P=$1; shift
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o ${P}stats -- "$@" > "$OUT/${P}stats.log" 2>&1
    timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT" -o ${P}fetch -- "$@" > "$OUT/${P}fetch.log" 2>&1
    timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT" -o ${P}write -- "$@" > "$OUT/${P}write.log" 2>&1
    timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT" -o ${P}mfma -- "$@" > "$OUT/${P}mfma.log" 2>&1
    python "$ROOT/tools/pmc_summary.py" "$OUT" "$OUT/${P}pmc_summary.json" "$P"
    head -8 "$OUT/${P}stats_kernel_stats.csv" | cut -c1-200
}
for leg in $LEGS; do
case $leg in
bench)
    # (the roofline leg stays on: it also launches the stand-alone scatter-add kernel, which the timed path fuses away)
    timeout 900 python "$ROOT/bench.py" > "$OUT/bench.json" 2> "$OUT/bench.err"
    tail -1 "$OUT/bench.json" | cut -c1-600
    four_passes "" python "$ROOT/bench.py" --steps 12 --warmup 4 --streams 1 --min-time 0 --no-cpu-baseline --no-secondary ;;
large) four_passes config5_ python "$ROOT/tools/bench_extra.py" large ;;
dense) four_passes config3_ python "$ROOT/tools/bench_extra.py" dense ;;
train)
    timeout 600 python "$ROOT/tools/bench_extra.py" train > "$OUT/train.json" 2> "$OUT/train.err"; tail -1 "$OUT/train.json"
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o train -- python "$ROOT/tools/bench_extra.py" train > "$OUT/train_stats.log" 2>&1
    head -12 "$OUT/train_kernel_stats.csv" | cut -c1-200 ;;
esac
done
# the raw traces are large; keep the per-dispatch counter files (small) and drop the traces
rm -f "$OUT"/*_kernel_trace.csv
