"""zipf stream with diag=2 trace dump (see tools/diag_chains.py)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
from helpers import load_pkg, load_streams
pkg, st = load_pkg(), load_streams()
n = 1 << 20
s, d = st.rmat_edges(20, 10_000_000, seed=1)
zs = st.zipf_sources(n, 1_000_000, seed=4, alpha=1.2)
zd = st.uniform_ints(11, 1_000_000, n)
e = pkg.PCSR(n)
e.apply(st.adds(s, d))
for kv in sys.argv[1:]:
    k, v = kv.split("="); e.set_option(k, int(v))
e.set_option("diag", 2)
a = e.stats()
e.apply(st.adds(zs, zd))
b = e.stats()
print("zipf ms", round(b["last_batch_ms"], 1), {k: b[k]-a[k] for k in ("rounds","committed","planned","exclusive_ops","rollbacks","wasted_rounds","chained")}, flush=True)
