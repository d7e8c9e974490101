import os, sys
sys.path.insert(0, "/root/repo/tests")
import numpy as np
from helpers import load_pkg, load_streams
pkg, st = load_pkg(), load_streams()
N4, P4, part = 10_000_000, 8, 3
ps = N4 // P4
def sub(s, d):
    s, d = st.permute_labels(s, N4), st.permute_labels(d, N4)
    m = np.minimum(s // np.uint32(ps), P4 - 1) == part
    return st.adds(s[m] - np.uint32(part * ps), d[m])
cs, cd = st.rmat_edges_folded(N4, 24, 100_000_000, seed=1)
core = sub(cs, cd); del cs, cd
us, ud = st.rmat_edges_folded(N4, 24, 10_000_000, seed=2)
upd = sub(us, ud)
e = pkg.PCSR(ps)
e.apply(core); e.snapshot()
for rep in range(4):
    e.set_option("profile", 1 if rep == 3 else 0)
    e.restore()
    a = e.stats(); e.apply(upd); b = e.stats()
    d = {k: b[k] - a[k] for k in ("rounds", "round_syncs", "exclusive_ops", "big_redistributes", "rollbacks", "planned", "committed", "double_calls")}
    print(rep, round(b["last_batch_ms"], 2), d, flush=True)
L = max(b["prof_launches"], 1)
print("events per launch us: plan %.1f check %.1f apply %.1f big %.1f (%d launches)" % (b["prof_plan_ms"]*1e3/L, b["prof_check_ms"]*1e3/L, b["prof_apply_ms"]*1e3/L, b["prof_compact_ms"]*1e3/L, L))
