#!/bin/bash
# config #4 headline leg only (8 partitions on one GPU): default host threading vs one partition after the other
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT"
for thr in "" 1; do
  PPCSR_PP_THREADS="$thr" python3 bench.py --config 4 --steps 2 --no-secondary --no-cpu-baseline --no-check --no-profile 2>/dev/null | python3 -c "
import sys,json
j=json.loads(sys.stdin.read()); e=j['engine']
print('PP_THREADS=${thr:-default}', 'value', round(j['value']/1e6,1), 'ms', round(j['ms_per_step'],1), {k:e.get(k) for k in ('rounds','rollbacks','exclusive_ops','double_calls','round_syncs','device_ms_last_batch')})"
done
