#!/usr/bin/env python3
"""PPPCSR on ONE GPU: P vertex-range partitions (independent engines) fed through pppcsr_apply_batch, which drives the
partitions from host threads so that their round kernels overlap.  usage: python tools/bench_pppcsr.py [P] [scale] [core_edges]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import load_pkg, load_streams  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
scale = int(sys.argv[2]) if len(sys.argv) > 2 else 20
m = int(sys.argv[3]) if len(sys.argv) > 3 else 10_000_000
pkg, st = load_pkg(), load_streams()
n = 1 << scale
s, d = st.rmat_edges(scale, m, seed=1)
s = st.permute_labels(s, n)
e = pkg.PPPCSR(n, numDomain=1, partitionsPerDomain=P)
t0 = time.perf_counter()
e.apply(st.adds(s, d))
t1 = time.perf_counter()
print(f"P={P}: core load {m} edges in {t1-t0:.2f} s = {m/(t1-t0)/1e6:.1f} M/s (incl. host bucketing + H2D)")
for k in range(3):
    s2, d2 = st.rmat_edges(scale, 1_000_000, seed=2 + k)
    ops = st.adds(st.permute_labels(s2, n), d2)
    t0 = time.perf_counter()
    e.apply(ops)
    t1 = time.perf_counter()
    print(f"P={P}: batch of 1M in {(t1-t0)*1e3:.1f} ms = {1/(t1-t0):.1f} M updates/s (host buffers: bucketing + H2D inside)")
    if os.environ.get("PPCSR_PP_STATS"):
        for q in range(P):
            stq = e.partition(q).stats()
            print("   part", q, {k2: stq[k2] for k2 in ("N", "rounds", "rollbacks", "exclusive_ops", "double_calls", "last_batch_ms")})
