#!/bin/bash
# Box-to-box variance probe: per-kernel times of one forward + clocks / power sampled while the GPU is under that load.
#   gpurun -- 'bash tools/box_probe.sh'
ROOT=$(cd "$(dirname "$0")/.." && pwd)
( for i in 1 2 3 4 5 6; do sleep 1.5; rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|Socket Graphics|Temperature \(Sensor junction\)" | tr '\n' ' '; echo; done ) > /tmp/smi.log 2>&1 &
SMI=$!
python "$ROOT/tools/fwd_kernels.py" 2>&1 | tail -2
python "$ROOT/tools/fwd_kernels.py" 2>&1 | tail -2
wait $SMI
cat /tmp/smi.log | cut -c1-220
