#!/bin/bash
# Kernel experiments: build libggnn_hip_<tag>.so = the current objects with ONE source recompiled under extra flags.
#   tools/variant_lib.sh <tag> <source.hip> [-DFLAG=...]      then run with GGNN_LIB_VARIANT=<tag>
set -e
tag=$1; src=$2; shift; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd); P=$ROOT/gated-graph-neural-network-samples_amd
base=$(basename "$src" .hip)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -I "$ROOT/include" "$@" -c "$P/csrc/$base.hip" -o "/tmp/${base}_$tag.o"
objs=$(ls "$P"/build/*.o | grep -v "/$base.o")
hipcc --offload-arch=gfx950 -shared -fPIC $objs "/tmp/${base}_$tag.o" -o "$P/libggnn_hip_$tag.so"
echo "built $P/libggnn_hip_$tag.so"
