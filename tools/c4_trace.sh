#!/bin/bash
# kernel trace of the config #4 headline leg (8 partitions on one GPU); summary by tools/trace_overlap.py
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/c4trace"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
rocprofv3 --kernel-trace --output-format csv -d "$OUT/t" -o run -- python3 bench.py --config 4 --steps 2 --no-secondary --no-cpu-baseline --no-check --no-profile > "$OUT/b.json" 2> "$OUT/b.err" || { tail -3 "$OUT/b.err"; exit 1; }
T=$(find "$OUT/t" -name "*kernel_trace.csv" | head -1)
python3 tools/trace_overlap.py "$T" > "$OUT/summary.txt"
cat "$OUT/summary.txt"
rm -rf "$OUT/t"
