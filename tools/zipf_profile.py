"""The hot-vertex stress stream (config #5's shape on the config #2 graph): 1 M inserts whose sources are Zipf(1.2) ranks —
18 % of them into ONE vertex.  python3 tools/zipf_profile.py [option=value ...]; PPCSR_TRACE_EXCL=1 lists the exclusive
updates, option diag=1 prints per epoch why planned updates did not commit."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
from helpers import load_pkg, load_streams
pkg, st = load_pkg(), load_streams()
n = 1 << 20
s, d = st.rmat_edges(20, 10_000_000, seed=1)
zs = st.zipf_sources(n, 1_000_000, seed=4, alpha=1.2)
zd = st.uniform_ints(11, 1_000_000, n)
z = st.adds(zs, zd)
e = pkg.PCSR(n)
e.apply(st.adds(s, d))
for kv in sys.argv[1:]:
    k, v = kv.split("="); e.set_option(k, int(v))
a = e.stats()
e.apply(z)
b = e.stats()
print("zipf ms", round(b["last_batch_ms"], 1), {k: b[k]-a[k] for k in ("rounds","committed","planned","exclusive_ops","rollbacks","wasted_rounds","big_redistributes","round_syncs")}, flush=True)
