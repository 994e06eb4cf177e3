#!/bin/bash
OUT=gpurun_out/${1:-exp_remat}; mkdir -p $OUT
export TMPDIR=/tmp
for v in "" rm; do
  echo "== variant '$v'" >> $OUT/res.txt
  GGNN_LIB_VARIANT=$v python tools/bench_extra.py large 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:(round(v,4) if isinstance(v,float) else v) for k,v in d.items() if not isinstance(v,(dict,list))}); print({k:round(v['avg_us'],1) for k,v in d.get('kernels',{}).items()})" >> $OUT/res.txt 2>&1
  GGNN_LIB_VARIANT=$v python tools/gru_bwd_bench.py 0 >> $OUT/res.txt 2>&1
  GGNN_LIB_VARIANT=$v python tools/bench_extra.py train 2>/dev/null | tail -1 | cut -c1-200 >> $OUT/res.txt
done
( GGNN_LIB_VARIANT=rm timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 ) > $OUT/pytest_rm.txt
cat $OUT/res.txt | grep -v amdgpu.ids; cat $OUT/pytest_rm.txt
