#!/bin/bash
# A/B of engine options on the hot-vertex stream (config #5's shape on the config #2 graph): each argument one variant ("-" = defaults)
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT"
for var in "$@"; do
  opts=""
  if [ "$var" != "-" ]; then for kv in ${var//,/ }; do opts="$opts --opt $kv"; done; fi
  python3 bench.py --zipf --no-cpu-baseline --no-ref-cli --no-check --no-secondary --no-profile --steps 3 --warmup 1 $opts 2>/dev/null | python3 -c "
import sys,json
j=json.loads(sys.stdin.read()); e=j['engine']
print('[$var]', 'value', round(j['value']/1e6,2), 'ms', round(j['ms_per_step'],1), {k:e.get(k) for k in ('rounds','wasted_rounds','rollbacks','exclusive_ops','round_syncs')})"
done
