"""config #2 (or the zipf stream with ZIPF=1) once, after a marker kernel: for profiling"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
from helpers import load_pkg, load_streams
pkg, st = load_pkg(), load_streams()
n = 1 << 20
s, d = st.rmat_edges(20, 10_000_000, seed=1)
e = pkg.PCSR(n)
e.apply(st.adds(s, d))
for kv in sys.argv[1:]:
    k, v = kv.split("="); e.set_option(k, int(v))
if os.environ.get("ZIPF"):
    upd = st.adds(st.zipf_sources(n, 1_000_000, seed=4, alpha=1.2), st.uniform_ints(11, 1_000_000, n))
else:
    s2, d2 = st.rmat_edges(20, 1_000_000, seed=2); upd = st.adds(s2, d2)
e.snapshot()
e.set_option("marker", 1)
a = e.stats(); e.apply(upd); b = e.stats()
print("ms", round(b["last_batch_ms"], 2), {k: b[k]-a[k] for k in ("rounds","committed","planned","rollbacks","wasted_rounds","chained")}, flush=True)
