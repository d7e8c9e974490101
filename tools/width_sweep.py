#!/usr/bin/env python3
"""Per-kernel time of a speculative round against the round width (fixed widths, adaptive off) on the config #2 graph:
separates a round's fixed cost from its cost per update.  usage: python tools/width_sweep.py [graph=rmat|uniform] [key=value ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from helpers import load_pkg, load_streams  # noqa: E402

pkg, st = load_pkg(), load_streams()
graph = "rmat"
opts = []
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    if k == "graph":
        graph = v
    else:
        opts.append((k, int(v)))
scale, m, u = 20, 10_000_000, 1_000_000
if graph == "rmat":
    s, d = st.rmat_edges(scale, m, seed=1)
    us, ud = st.rmat_edges(scale, u, seed=2)
else:
    s, d = st.uniform_ints(11, m, 1 << scale), st.uniform_ints(12, m, 1 << scale)
    us, ud = st.uniform_ints(13, u, 1 << scale), st.uniform_ints(14, u, 1 << scale)
n = int(max(s.max(), d.max())) + 1
e = pkg.PCSR(n)
e.apply(st.adds(s, d))
e.snapshot()
upd = st.adds(us, ud)
for k, v in opts:
    e.set_option(k, v)
widths = [int(x) for x in os.environ.get("WIDTHS", "1536,3072,6144,9216,12288,18432,24576").split(",")]
print(f"graph {graph}: n {n}, geometry {e.geometry()}")
for w in widths:
    e.set_option("adaptive", 0)
    e.set_option("opt_horizon", w)
    for prof in (0, 1):
        e.set_option("profile", prof)
        e.restore()
        a = e.stats()
        e.apply(upd)
        b = e.stats()
        r = b["rounds"] - a["rounds"]
        if not prof:
            print(f"width {w:6d}: {b['last_batch_ms']:.2f} ms = {u / b['last_batch_ms'] / 1e3:6.1f} M/s, rounds {r} (+{b['wasted_rounds'] - a['wasted_rounds']} wasted), "
                  f"commits/round {u / max(r, 1):.0f}, {b['last_batch_ms'] * 1e3 / max(r, 1):.1f} us/round", flush=True)
        else:
            L = max(b["prof_launches"], 1)
            print(f"              events: plan {b['prof_plan_ms'] * 1e3 / L:.1f} check {b['prof_check_ms'] * 1e3 / L:.1f} apply {b['prof_apply_ms'] * 1e3 / L:.1f} "
                  f"compact {b['prof_compact_ms'] * 1e3 / L:.1f} us per launch ({L} launches)", flush=True)
