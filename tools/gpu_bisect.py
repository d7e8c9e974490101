#!/usr/bin/env python3
"""Debug aid (GPU box): find the first update after which the HIP engine's state differs from the oracle.
usage: python tools/gpu_bisect.py [scale] [core_edges] [max_horizon]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import load_pkg, load_streams  # noqa: E402
from oracle_lib import Oracle  # noqa: E402

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 20
m = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
maxh = int(sys.argv[3]) if len(sys.argv) > 3 else 0
pkg, st = load_pkg(), load_streams()
n = 1 << scale
s, d = st.rmat_edges(scale, m, seed=1)
ops = st.adds(s, d)


def mk():
    e = pkg.PCSR(n)
    if maxh:
        e.set_option("max_horizon", maxh)
    return e


def same(e, o):
    if e.geometry() != o.geometry():
        return False
    ei, en = e.state()
    oi, on = o.state()
    return np.array_equal(ei, oi) and np.array_equal(en, on)


def describe(e, o):
    print("geometry", e.geometry(), o.geometry())
    if e.geometry() != o.geometry():
        return
    ei, en = e.state()
    oi, on = o.state()
    bad = np.nonzero((ei != oi).any(1))[0]
    print("bad slots:", len(bad), bad[:16], "span", (bad.min(), bad.max()) if len(bad) else None)
    if len(bad):
        lo = max(0, bad[0] - 2)
        print("eng:", ei[lo:lo + 24].tolist())
        print("ora:", oi[lo:lo + 24].tolist())
    bn = np.nonzero((en != on).any(1))[0]
    print("bad nodes:", len(bn), bn[:8], en[bn[:4]].tolist(), on[bn[:4]].tolist())
    print("bad leafcnt:", e.check_invariants())


chunk = 250_000
e, o = mk(), Oracle(n)
t0 = time.time()
lo = 0
while lo < len(ops):
    hi = min(lo + chunk, len(ops))
    e.apply(ops[lo:hi])
    o.apply(ops[lo:hi])
    ok = same(e, o)
    print(f"ops [{lo},{hi}) {'ok' if ok else 'MISMATCH'} N={e.geometry()[0]} t={time.time()-t0:.1f}s", flush=True)
    if not ok:
        break
    lo = hi
else:
    print("no divergence", e.stats())
    sys.exit(0)
describe(e, o)
# bisect inside [lo,hi): replay prefix on fresh engines
step = chunk
base_lo = lo
while step > 1:
    step = max(1, step // 8)
    e2, o2 = mk(), Oracle(n)
    if base_lo:
        e2.apply(ops[:base_lo])
        o2.apply(ops[:base_lo])
    if not same(e2, o2):
        print("replay of the clean prefix is not clean -> nondeterministic (race)")
        describe(e2, o2)
        break
    p = base_lo
    found = False
    while p < hi:
        q = min(p + step, hi)
        e2.apply(ops[p:q])
        o2.apply(ops[p:q])
        if not same(e2, o2):
            print(f"  step {step}: first bad sub-chunk [{p},{q})", flush=True)
            base_lo, hi = p, q
            found = True
            if step == 1:
                print("first bad op:", ops[p], "stats", e2.stats())
                describe(e2, o2)
            break
        p = q
    if not found:
        print(f"  step {step}: sub-chunks all clean -> depends on batching (ops interact inside one batch)")
        # retry the whole bad chunk in one batch on the clean prefix
        e3, o3 = mk(), Oracle(n)
        if base_lo:
            e3.apply(ops[:base_lo]); o3.apply(ops[:base_lo])
        e3.apply(ops[base_lo:hi]); o3.apply(ops[base_lo:hi])
        print("  whole chunk again:", "ok" if same(e3, o3) else "MISMATCH")
        describe(e3, o3)
        break
