import os, sys, time
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np
from helpers import load_pkg, load_streams
from oracle_lib import Oracle
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
def describe(e, o, label):
    ok = e.geometry() == o.geometry()
    print(label, "geometry", e.geometry(), o.geometry())
    if not ok: return False
    ei, en = e.state(); oi, on = o.state()
    bad = np.nonzero((ei != oi).any(1))[0]
    bn = np.nonzero((en != on).any(1))[0]
    print(label, "bad slots:", len(bad), bad[:12], "span", (bad.min(), bad.max()) if len(bad) else None, "bad nodes", len(bn), bn[:8], "bad leafcnt", e.check_invariants())
    if len(bad):
        lo = max(0, bad[0] - 2)
        print(" eng:", ei[lo:lo + 12].tolist()); print(" ora:", oi[lo:lo + 12].tolist())
    if len(bn):
        print(" eng nodes", en[bn[:4]].tolist(), "ora", on[bn[:4]].tolist())
    return len(bad) == 0 and len(bn) == 0
e = pkg.PCSR(ps)
for kv in sys.argv[1:]:
    k, v = kv.split("="); e.set_option(k, int(v))
o = Oracle(ps)
e.apply(core); o.apply(core)
describe(e, o, "core")
print("core stats", {k: v for k, v in e.stats().items() if k in ("rounds", "rollbacks", "exclusive_ops", "double_calls", "big_redistributes", "wasted_rounds", "chained")}, flush=True)
if os.environ.get("CORE_ONLY"): sys.exit(0)
e.snapshot()
for rep in range(2):
    e.restore()
    describe(e, o, f"rep{rep} after restore")
    e.apply(upd)
    o2 = o.clone(); o2.apply(upd)
    ok = describe(e, o2, f"rep{rep} after updates")
    print("stats", {k: v for k, v in e.stats().items() if k in ("rounds", "rollbacks", "exclusive_ops", "double_calls", "big_redistributes", "wasted_rounds")})
    o2.close()
# chunked
e.restore()
o3 = o.clone()
for lo in range(0, len(upd), 100000):
    e.apply(upd[lo:lo+100000]); o3.apply(upd[lo:lo+100000])
    ei, en = e.state(); oi, on = o3.state()
    ok = e.geometry() == o3.geometry() and np.array_equal(ei, oi) and np.array_equal(en, on)
    print("chunk", lo, "ok" if ok else "MISMATCH", e.geometry(), flush=True)
    if not ok:
        describe(e, o3, "chunk"); break
