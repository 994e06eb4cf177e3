#!/bin/bash
# One-line digest of bench.py's JSON (value, ms/step, per-kernel average launch time): bash tools/bench_brief.sh [bench args]
python bench.py "$@" 2>&1 | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print('%.1f M/s  %.4f ms/step  streams=%s ' % (d['value'] / 1e6, d['ms_per_step'], d['config'].get('hip_streams')),
      {k: (round(v['avg_us'], 1), round(v.get('median_us', 0), 1), round(v.get('max_us', 0), 1)) for k, v in d.get('kernels', {}).items()})"
