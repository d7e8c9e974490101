#!/bin/bash
# config #2 rounds only: bench value (two runs) + per-kernel averages from a kernel trace of the same command
set -o pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/c2"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
ARGS="bench.py --no-cpu-baseline --no-ref-cli --no-secondary --no-profile --steps 5 --warmup 2"
for i in 1 2; do python3 $ARGS 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('value', round(j['value']/1e6,1), 'M/s', round(j['ms_per_step'],3), 'ms', 'parity', j.get('parity_checked'))"; done
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o run -- python3 $ARGS --no-check > "$OUT/b.json" 2> "$OUT/b.err" || { tail -3 "$OUT/b.err"; exit 1; }
S=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1)
python3 - "$S" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name'].split('(')[0]
    if n.startswith('ppcsr::o_') or 'snap' in n: print('%-32s %5s avg %7.2f us' % (n, r['Calls'], float(r['AverageNs'])/1e3))
PY
rm -rf "$OUT/trace"
