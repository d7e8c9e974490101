#!/bin/bash
# The engine's kernels + host code compiled for the CPU emulator WITH AddressSanitizer (CPU only: GPU sanitizers are not available
# on this pool) and driven through a few streams — out-of-bounds reads / writes of the emulated kernels show up here.
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SIM="$ROOT/tests/hostsim"; CS="$ROOT/parallel-packed-csr_amd/csrc"
OUT="${PPCSR_SIM_ASAN:-/tmp/libppcsr_sim_asan.so}"
g++ -O1 -g -std=c++17 -ffp-contract=off -Wno-unknown-pragmas -fPIC -shared -fsanitize=address -fno-omit-frame-pointer -I"$SIM" -I"$CS" \
  "$SIM/ppcsr_sim.cpp" "$SIM/sim_runtime.cpp" "$SIM/sim_xchg.cpp" -lrt -o "$OUT" || exit 1
PPCSR_SIM_ASAN="$OUT" ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:verify_asan_link_order=0 LD_PRELOAD="$(gcc -print-file-name=libasan.so)" \
  python3 "$ROOT/tools/sim_asan_run.py"
