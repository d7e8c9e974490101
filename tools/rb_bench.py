#!/usr/bin/env python3
"""Rebalance kernels alone (device time, HIP events): whole-array, half-array (in place) and quarter-array (through scratch)
windows at 2^24 and 2^25 slots; optional option variants ("k=v,k=v;k=v") are timed beside the defaults and a state digest shows
that all leave the same array.  usage: python tools/rb_bench.py [scales, e.g. 20,21] [variants]"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import load_pkg, load_streams  # noqa: E402

pkg, st = load_pkg(), load_streams()
out = {}
for scale in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "20,21").split(",")]:
    n = 1 << scale
    m = 10_000_000 << (scale - 20)
    s, d = st.rmat_edges(scale, m, seed=1)
    e = pkg.PCSR(n)
    e.bulk_build(st.adds(s, d))
    N = e.geometry()[0]
    row = {"N_slots": int(N)}
    variants = [("", {})]
    for spec in (sys.argv[2].split(";") if len(sys.argv) > 2 else []):
        kv = dict((a.split("=")[0], int(a.split("=")[1])) for a in spec.split(","))
        variants.append((":" + spec, kv))
    for g, opts in variants:
        for kk, vv in opts.items():
            e.set_option(kk, vv)
        for label, w in (("whole", N), ("half", N // 2), ("quarter_scratch", N // 4)):
            if label == "quarter_scratch":
                e.set_option("rb_inplace_min", 0)  # through the scratch stretch + copy-back
            e.bench_rebalance(w, 1)
            ms = e.bench_rebalance(w, 10)
            if label == "quarter_scratch":
                e.set_option("rb_inplace_min", 524288)
            row[f"{label}{g}"] = {"us": ms * 1e3, "alg_TBps": 24.0 * w / (ms * 1e-3) / 1e12, "frac": 24.0 * w / (ms * 1e-3) / 8e12}
        items, nodes = e.state()
        row[f"digest{g}"] = hashlib.sha256(items.tobytes() + nodes.tobytes()).hexdigest()[:16]
        assert e.check_invariants() == 0
    row["same_state"] = len({v for k, v in row.items() if k.startswith("digest")}) == 1
    out[f"scale{scale}"] = row
    print(json.dumps({f"scale{scale}": row}), flush=True)
    e.close()
