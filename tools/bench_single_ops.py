#!/usr/bin/env python3
"""Latency of the single-update API (ppcsr_add_edge / remove_edge / edge_exists: one call = one batch of one update) and of
small batches on the config #2 graph.  usage: python tools/bench_single_ops.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import load_pkg, load_streams  # noqa: E402

pkg, st = load_pkg(), load_streams()
s, d = st.rmat_edges(20, 10_000_000, seed=1)
e = pkg.PCSR(1 << 20)
e.bulk_build(st.adds(s, d)) if os.environ.get("PPCSR_BULK") else e.apply(st.adds(s, d))
s2, d2 = st.rmat_edges(20, 4096, seed=7)
t0 = time.perf_counter()
for i in range(300):
    e.add_edge(int(s2[i]), int(d2[i]), 1)
t1 = time.perf_counter()
print(f"add_edge: {(t1-t0)/300*1e6:.0f} us per call")
t0 = time.perf_counter()
for i in range(300):
    e.edge_exists(int(s2[i]), int(d2[i]))
t1 = time.perf_counter()
print(f"edge_exists: {(t1-t0)/300*1e6:.0f} us per call")
for bs in (16, 256, 4096):
    ops = st.adds(s2[:bs], d2[:bs] + 1)
    t0 = time.perf_counter()
    for _ in range(20):
        e.apply(ops)
    t1 = time.perf_counter()
    print(f"batch of {bs}: {(t1-t0)/20*1e6:.0f} us per batch")
