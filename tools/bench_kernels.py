#!/usr/bin/env python3
"""Micro-benchmark of the HBM-bound kernels (whole-window rebalance, bulk neighbour scan) on the config #2 core graph.
usage: python tools/bench_kernels.py [scale] [core_edges]   (run under rocprofv3 --kernel-trace --stats for the breakdown)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import load_pkg, load_streams  # noqa: E402

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 20
m = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
pkg, st = load_pkg(), load_streams()
s, d = st.rmat_edges(scale, m, seed=1)
e = pkg.PCSR(1 << scale)
e.apply(st.adds(s, d))
N = e.geometry()[0]
for w in (N, N // 2, N // 16):
    ms = e.bench_rebalance(w, 10)
    print(f"rebalance window {w}: {ms*1e3:.1f} us  -> {24.0*w/ms/1e6:.0f} GB/s algorithmic ({24.0*w/ms/1e6/80:.1f} % of 8 TB/s)")
if os.environ.get("PPCSR_SWEEP"):
    for pf in (0, 1):
        for tile in (32, 64, 128, 256):
            e.set_option("rb_prefetch", pf)
            e.set_option("rb_tile", tile)
            r = [e.bench_rebalance(w, 10) * 1e3 for w in (N, N // 2, N // 16, N // 128)]
            print(f"prefetch={pf} tile={tile}: " + "  ".join(f"{x:.1f}us" for x in r))
    e.set_option("rb_tile", 0)
    e.set_option("rb_prefetch", 1)
for _ in range(3):
    ms, tot = e.bench_scan_all()
stt = e.stats()
b = 12.0 * stt["N"] + 12.0 * stt["n"] + 4.0 * tot
print(f"scan_all: {ms*1e3:.1f} us, {tot} edges -> {tot/ms/1e6:.1f} G edges/s, {b/ms/1e6:.0f} GB/s algorithmic ({b/ms/1e6/80:.1f} % of 8 TB/s)")
