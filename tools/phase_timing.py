#!/usr/bin/env python3
"""Where does an o_plan wave spend its time?  Runs config #2 (core load + one 1 M batch) on the profiling build
(libppcsr_hip_timing.so, built by `python -c "import build; build.build_timing()"` in parallel-packed-csr_amd/) and prints
the per-phase wave clocks accumulated by the kernel.  usage: PPCSR_LIB=.../libppcsr_hip_timing.so python tools/phase_timing.py"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import load_pkg, load_streams  # noqa: E402

pkg, st = load_pkg(), load_streams()
L = pkg.load_library()
s, d = st.rmat_edges(20, 10_000_000, seed=1)
e = pkg.PCSR(1 << 20)
e.apply(st.adds(s, d))
out = (ctypes.c_ulonglong * 89)()
L.ppcsr_debug_phase_read(out, 1)
s2, d2 = st.rmat_edges(20, 1_000_000, seed=2)
e.apply(st.adds(s2, d2))
L.ppcsr_debug_phase_read(out, 1)
o = np.array(out[:], dtype=np.float64)
cnt = o[24]
names = ["", "ctl+carry", "op load", "nodes[src]", "search", "slot/leafcnt/gap batch", "plan_insert/remove", "sentinel range",
         "plan record", "reservations"]
print(f"waves: {int(cnt)}")
for i in range(1, 10):
    print(f"{i} {names[i]:26s} avg {o[i]/cnt/100:7.2f} us   cumulative max {o[12+i]/100:7.2f} us")
h = o[25:]
tot = h.sum()
acc = 0
for b, x in enumerate(h):
    acc += x
    if x:
        print(f"  wave duration {b*0.5:4.1f}-{b*0.5+0.5:4.1f} us: {x/tot*100:5.1f} %  (cum {acc/tot*100:5.1f} %)")
