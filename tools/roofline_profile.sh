#!/bin/bash
# Reproducible roofline record (run on the GPU box: `gpurun -- bash tools/roofline_profile.sh`).
# Three passes of the SAME command, the program directly after `--` (no env / bash hop: the profiler has initialised the GPU):
#   1. rocprofv3 --kernel-trace --stats        -> per-dispatch durations (and the --stats summary)
#   2. rocprofv3 --pmc FETCH_SIZE              -> HBM read bytes per dispatch   (counters in runs of their own)
#   3. rocprofv3 --pmc WRITE_SIZE              -> HBM written bytes per dispatch
# bench.py --markers launches empty marker kernels (k_mark_0..7) around (i) the timed rounds, (ii) full-array rebalances,
# (iii) half-array rebalances, (iv) one neighbour scan; tools/roofline_summary.py cuts those sections out of all three passes
# and writes gpurun_out/roofline/${ROUND}_roofline.json (copy it to profiles/).  ROUND defaults to r04; PPCSR_COMMIT (the box has no
# .git) is recorded as the commit the record was measured on.
set -o pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
ROUND="${ROUND:-r04}"
OUT="$ROOT/gpurun_out/roofline"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
ARGS="bench.py --markers --no-cpu-baseline --no-ref-cli --no-check --steps 5 --warmup 2"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o run -- python3 $ARGS > "$OUT/bench_trace.json" 2> "$OUT/bench_trace.err" || { echo "trace pass failed"; tail -5 "$OUT/bench_trace.err"; exit 1; }
echo "trace pass done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o run -- python3 $ARGS > "$OUT/bench_fetch.json" 2> "$OUT/bench_fetch.err" || { echo "FETCH_SIZE pass failed"; tail -5 "$OUT/bench_fetch.err"; exit 1; }
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o run -- python3 $ARGS > "$OUT/bench_write.json" 2> "$OUT/bench_write.err" || { echo "WRITE_SIZE pass failed"; tail -5 "$OUT/bench_write.err"; exit 1; }
echo "write pass done"
T=$(find "$OUT/trace" -name "*kernel_trace.csv" | head -1)
S=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1)
F=$(find "$OUT/fetch" -name "*counter_collection.csv" | head -1)
W=$(find "$OUT/write" -name "*counter_collection.csv" | head -1)
python3 tools/roofline_summary.py "$T" "$F" "$W" "$OUT/bench_trace.json" "$OUT/${ROUND}_roofline.json" "python3 $ARGS" || exit 1
[ -n "$S" ] && cp "$S" "$OUT/${ROUND}_kernel_stats.csv"
# the raw per-dispatch files are large: keep only the summaries under gpurun_out/
rm -rf "$OUT/trace" "$OUT/fetch" "$OUT/write"
ls -la "$OUT"
