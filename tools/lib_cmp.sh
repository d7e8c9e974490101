#!/bin/bash
# A/B of two builds of the engine library on config #2: PPCSR_LIB selects the build (debugging hook of the binding)
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT"
for rep in 1 2 3; do
  for lib in "" "$1"; do
    PPCSR_LIB="$lib" python3 bench.py --no-secondary --no-cpu-baseline --no-ref-cli --steps 5 --warmup 2 2>/dev/null | python3 -c "
import sys,json
j=json.loads(sys.stdin.read()); k=j['roofline']['kernel_ms']; n=j['roofline']['launches']
print('${lib:-default}'.split('/')[-1], 'value', round(j['value']/1e6,1), 'ms', round(j['ms_per_step'],3), {a: round(b/n*1e3,1) for a,b in k.items()}, 'parity', j.get('parity_checked'))"
  done
done
