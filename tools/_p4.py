import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
from helpers import load_pkg, load_streams
pkg, st = load_pkg(), load_streams()
N4, SCALE4, CORE4, UPD4, P4 = 10_000_000, 24, 100_000_000, 10_000_000, 8
t0 = time.time()
cs, cd = st.rmat_edges_folded(N4, SCALE4, CORE4, seed=1)
us, ud = st.rmat_edges_folded(N4, SCALE4, UPD4, seed=2)
print("gen", round(time.time() - t0, 1), flush=True)
def sub(s, d, part):
    s = st.permute_labels(s, N4); d = st.permute_labels(d, N4)
    ps = N4 // P4
    own = np.minimum(s // np.uint32(ps), P4 - 1)
    m = own == part
    return st.adds(s[m] - np.uint32(part * ps), d[m])
part = 3
core, upd = sub(cs, cd, part), sub(us, ud, part)
zs = st.zipf_sources(N4, UPD4, seed=4, alpha=1.2); zd = st.uniform_ints(11, UPD4, N4)
zupd = sub(zs, zd, part)
del cs, cd, us, ud
ref = None
for sb, ga, es in ((16384, 8, 8192), (16384, 2, 8192), (16384, 1, 8192), (16384, 2, 16384), (16384, 4, 16384)):
    e = pkg.PCSR(N4 // P4)
    e.set_option("soft_barrier", sb); e.set_option("epoch_grow_after", ga); e.set_option("epoch_short", es)
    e.apply(core)
    a = e.stats()
    e.snapshot()
    e.apply(upd)
    b = e.stats()
    st1 = e.state()
    e.restore()
    e.apply(zupd)
    c = e.stats()
    print({k: b[k]-a[k] for k in ("rounds","committed","planned","exclusive_ops","rollbacks","wasted_rounds","big_redistributes","round_syncs")}, {k: c[k]-b[k] for k in ("rounds","planned","exclusive_ops","rollbacks")})
    print("soft_barrier", sb, "grow_after", ga, "short", es, "core ms", round(a["last_batch_ms"], 1), "| inserts", len(upd), "ms", round(b["last_batch_ms"], 2), "=", round(len(upd) / b["last_batch_ms"] / 1e3, 1), "M/s rounds", b["rounds"] - a["rounds"],
          "| zipf", len(zupd), "ms", round(c["last_batch_ms"], 1), "=", round(len(zupd) / c["last_batch_ms"] / 1e3, 2), "M/s", flush=True)
    if ref is None: ref = st1
    else: print("identical:", np.array_equal(ref[0], st1[0]) and np.array_equal(ref[1], st1[1]))
    e.close()
