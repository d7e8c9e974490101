#!/bin/bash
# dynamic instruction cost of parts of o_plan: SQ counters of the diagnostics build with a part executed twice (option dbg_repeat)
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/sqd"
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
for rep in "$@"; do
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_BRANCH --output-format csv -d "$OUT/p$rep" -o run -- python3 tools/sq_profile.py 1 dbg_repeat=$rep > "$OUT/p$rep.log" 2>&1 || { tail -5 "$OUT/p$rep.log"; exit 1; }
  echo "dbg_repeat=$rep"; python3 tools/pmc_summary.py "$OUT/p$rep" | grep "^o_plan"
  rm -rf "$OUT/p$rep"
done
