#!/bin/bash
# SQ / cache counter passes over one forward workload (tools/fwd_kernels.py), one rocprofv3 --pmc pass per counter group:
#   tools/sq_passes.sh <outdir> [env assignments...]     e.g.  tools/sq_passes.sh gpurun_out/sq_form4 GGNN_GRU_FORM=4
set -u
OUT=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
export TMPDIR=/tmp
cd /tmp
i=0
while read -r group; do
    [ -z "$group" ] && continue
    i=$((i+1))
    env "$@" timeout 300 rocprofv3 --kernel-trace --pmc $group --output-format csv -d "$OUT" -o p$i -- python "$ROOT/tools/fwd_kernels.py" > "$OUT/p$i.log" 2>&1
    echo "pass $i: $group -> $(ls "$OUT"/p${i}_counter_collection.csv 2>/dev/null | wc -l) file(s)"
done <<'G'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC
SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD
SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM
SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_WR SQ_INST_LEVEL_LDS SQ_INSTS_SALU
SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM
TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum
G
rm -f "$OUT"/*_kernel_trace.csv "$OUT"/*_agent_info.csv
