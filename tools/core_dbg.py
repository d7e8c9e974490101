"""debug aid: RMAT core load in chunks against the oracle, reporting num_neighbors differences"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
from helpers import load_pkg, load_streams
from oracle_lib import Oracle
pkg, st = load_pkg(), load_streams()
scale = int(os.environ.get("SCALE", 20)); n = 1 << scale
m = int(os.environ.get("EDGES", 10_000_000))
s, d = st.rmat_edges(scale, m, seed=1)
core = st.adds(s, d)
e = pkg.PCSR(n); o = Oracle(n)
for kv in sys.argv[1:]:
    k, v = kv.split("="); e.set_option(k, int(v))
chunk = int(os.environ.get("CHUNK", 500000))
prev = e.stats()
for lo in range(0, len(core), chunk):
    part = core[lo:lo + chunk]
    e.apply(part); o.apply(part)
    ei, en = e.state(); oi, on = o.state()
    st_ = e.stats()
    d_ = {k: st_[k] - prev[k] for k in ("rounds", "chained", "rollbacks", "exclusive_ops", "double_calls", "duplicates", "wasted_rounds")}
    prev = st_
    ok_items = e.geometry() == o.geometry() and np.array_equal(ei, oi)
    bn = np.nonzero(en[:, 2] != on[:, 2])[0] if e.geometry() == o.geometry() else []
    print("chunk", lo, "items", "ok" if ok_items else "MISMATCH", "nn diffs", len(bn), d_, flush=True)
    if len(bn) or not ok_items:
        for v in bn[:8]:
            cnt = int((part[:, 0] == v).sum())
            print("   vertex", int(v), "eng", int(en[v, 2]), "ora", int(on[v, 2]), "ops of this vertex in the chunk", cnt, "first positions", np.nonzero(part[:, 0] == v)[0][:6].tolist())
        break
