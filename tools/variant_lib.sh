#!/bin/bash
# Kernel experiments: build libggnn_hip_<tag>.so = the current objects with some sources recompiled under extra flags.
#   tools/variant_lib.sh <tag> <source.hip[,source2.hip,...]> [-DFLAG=...]      then run with GGNN_LIB_VARIANT=<tag>
# e.g. the fused GRU with its s_memtime stamps for tools/gru_timeline.py:
#   tools/variant_lib.sh tl ggnn_gru_fused.hip,ggnn_gru_fused_split.hip -DGGNN_GRU_STAMPS=1
set -e
tag=$1; srcs=$2; shift; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd); P=$ROOT/gated-graph-neural-network-samples_amd
objs=$(ls "$P"/build/*.o)
new=""
for src in ${srcs//,/ }; do
    base=$(basename "$src" .hip)
    extra=""
    case $base in   # the translation units build.py compiles without packed-f32 vector instructions
        *_split|ggnn_msg_compact|ggnn_panel|ggnn_bwd_gemm|ggnn_gru_wide) extra="-Xclang -target-feature -Xclang -packed-fp32-ops -mllvm -amdgpu-use-amdgpu-trackers=1" ;;
    esac
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -I "$ROOT/include" $extra "$@" -c "$P/csrc/$base.hip" -o "/tmp/${base}_$tag.o"
    objs=$(echo "$objs" | grep -v "/$base.o")
    new="$new /tmp/${base}_$tag.o"
done
hipcc --offload-arch=gfx950 -shared -fPIC $objs $new -o "$P/libggnn_hip_$tag.so"
echo "built $P/libggnn_hip_$tag.so"
