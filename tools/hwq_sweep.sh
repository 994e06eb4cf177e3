for q in 1 2 3 4 8; do
  GPU_MAX_HW_QUEUES=$q python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); e=d['end_to_end_fresh_batch']; print('HWQ=$q bench:', round(d['value']/1e6,1), 'e2e', round(e['value']/1e6,1), round(e['one_stream_value']/1e6,1), 'train', round(d['train']['ms_per_step'],3))"
done
