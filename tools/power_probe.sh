#!/bin/bash
# Clocks and socket power while the forward runs back to back, random data against all-zero data:  gpurun -- 'bash tools/power_probe.sh'
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for d in rand zero; do
    echo "== GGNN_FWD_DATA=$d"
    GGNN_FWD_DATA=$d python "$ROOT/tools/power_probe.py" 8 > /tmp/pp_$d.log 2>&1 &
    P=$!
    while ! grep -q LOAD-START /tmp/pp_$d.log 2>/dev/null; do sleep 0.5; kill -0 $P 2>/dev/null || break; done
    for i in 1 2 3 4 5; do
        sleep 1
        rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Socket|Sensor junction" | sed -E 's/^GPU\[[0-9]+\][ \t]*: //' | tr '\n' '|'; echo
    done
    wait $P
    grep -v "amdgpu.ids\|LOAD-START" /tmp/pp_$d.log | tail -1
done
