#!/bin/bash
# SQ instruction counters of the rebalance kernels (two --pmc passes of tools/rb_bench.py 20)
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/sqrb"
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_BRANCH --output-format csv -d "$OUT/p1" -o run -- python3 tools/rb_bench.py 20 > "$OUT/p1.log" 2>&1 || { tail -5 "$OUT/p1.log"; exit 1; }
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_BUSY_CYCLES --output-format csv -d "$OUT/p2" -o run -- python3 tools/rb_bench.py 20 > "$OUT/p2.log" 2>&1 || { tail -5 "$OUT/p2.log"; exit 1; }
python3 tools/pmc_summary.py "$OUT/p1" "$OUT/p2" > "$OUT/summary.txt"
rm -rf "$OUT/p1" "$OUT/p2"
grep "^k_rb\|^k_scan" "$OUT/summary.txt"
