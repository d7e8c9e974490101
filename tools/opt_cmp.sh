#!/bin/bash
# A/B of engine options on config #2 / #3: each argument is one variant, a comma-separated list of key=value ("-" = defaults)
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT"
for rep in 1 2; do
  for var in "$@"; do
    opts=""
    if [ "$var" != "-" ]; then for kv in ${var//,/ }; do opts="$opts --opt $kv"; done; fi
    python3 bench.py --no-cpu-baseline --no-ref-cli --no-check --steps 5 --warmup 2 $EXTRA $opts 2>/dev/null > /tmp/oc.json
    echo "[$var]"; python3 tools/bench_brief.py /tmp/oc.json | head -2
  done
done
