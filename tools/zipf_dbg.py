"""debug aid: the zipf stream applied in chunks against the oracle (options as key=value)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
from helpers import load_pkg, load_streams
from oracle_lib import Oracle
pkg, st = load_pkg(), load_streams()
n = 1 << 20
s, d = st.rmat_edges(20, 10_000_000, seed=1)
core = st.adds(s, d)
zs = st.zipf_sources(n, 1_000_000, seed=4, alpha=1.2)
zd = st.uniform_ints(11, 1_000_000, n)
upd = st.adds(zs, zd)
e = pkg.PCSR(n); o = Oracle(n)
early = os.environ.get("EARLY", "0") == "1"
if early:
    for kv in sys.argv[1:]:
        k, v = kv.split("="); e.set_option(k, int(v))
e.apply(core); o.apply(core)
ei, en = e.state(); oi, on = o.state()
print("core:", "ok" if (np.array_equal(ei, oi) and np.array_equal(en, on)) else "MISMATCH", "nn diffs", int((en[:, 2] != on[:, 2]).sum()), {k: v for k, v in e.stats().items() if k in ("chained", "rollbacks", "exclusive_ops", "double_calls")}, flush=True)
if not early:
    for kv in sys.argv[1:]:
        k, v = kv.split("="); e.set_option(k, int(v))
chunk = int(os.environ.get("CHUNK", 50000))
for lo in range(0, len(upd), chunk):
    e.apply(upd[lo:lo+chunk]); o.apply(upd[lo:lo+chunk])
    ei, en = e.state(); oi, on = o.state()
    ok = e.geometry() == o.geometry() and np.array_equal(ei, oi) and np.array_equal(en, on)
    print("chunk", lo, "ok" if ok else "MISMATCH", flush=True)
    if not ok:
        bad = np.nonzero((ei != oi).any(1))[0]
        bn = np.nonzero((en != on).any(1))[0]
        print("geometry", e.geometry(), o.geometry(), "bad slots", len(bad), "bad nodes", len(bn))
        if len(bn):
            for f, name in enumerate(("beginning", "end", "num_neighbors")):
                bf = np.nonzero(en[:, f] != on[:, f])[0]
                print("  field", name, "differs at", len(bf), bf[:10], "eng", en[bf[:10], f].tolist(), "ora", on[bf[:10], f].tolist())
        if not len(bad):
            print("stats", {k: v for k, v in e.stats().items() if k in ("duplicates", "not_found", "noops", "committed", "chained", "rollbacks")}, o.stats())
            break
        print("bad slots", len(bad), bad[:16], "span", bad.min(), bad.max(), "bad nodes", len(bn), bn[:8], "bad leafcnt", e.check_invariants())
        lo_ = max(0, int(bad[0]) - 4)
        print(" eng:", ei[lo_:lo_ + 16].tolist()); print(" ora:", oi[lo_:lo_ + 16].tolist())
        if len(bn): print(" eng nodes", en[bn[:4]].tolist(), "ora", on[bn[:4]].tolist())
        # multiset comparison of the live edges: lost / duplicated?
        le = ei[ei[:, 2] != 0]; lo2 = oi[oi[:, 2] != 0]
        print(" live", len(le), len(lo2))
        key = (le[:, 0].astype(np.uint64) << np.uint64(32)) | le[:, 1].astype(np.uint64)
        uk, cnt = np.unique(key, return_counts=True)
        dups = uk[cnt > 1]
        print(" duplicated (src,dst) in the engine:", len(dups), [(int(k >> np.uint64(32)), int(k & np.uint64(0xFFFFFFFF))) for k in dups[:12]])
        for k in dups[:6]:
            pos = np.nonzero((ei[:, 0] == (k >> np.uint64(32))) & (ei[:, 1] == (k & np.uint64(0xFFFFFFFF))) & (ei[:, 2] != 0))[0]
            print("   key", (int(k >> np.uint64(32)), int(k & np.uint64(0xFFFFFFFF))), "at slots", pos.tolist(), "leaves", (pos // 32).tolist(), "regions(1024)", (pos // 1024).tolist())
            uidx = np.nonzero((upd[:, 0] == (k >> np.uint64(32))) & (upd[:, 1] == (k & np.uint64(0xFFFFFFFF))))[0]
            print("      stream positions of that update:", uidx.tolist())
            p0 = int(pos[0]); print("      around:", ei[max(0, p0 - 3):p0 + 4].tolist())
        # sortedness inside vertex 0
        b, en_ = int(en[0, 0]), int(en[0, 1])
        seg = ei[b + 1:en_]; lv = seg[seg[:, 2] != 0][:, 1].astype(np.int64)
        bad_order = np.nonzero(np.diff(lv) <= 0)[0]
        print(" vertex 0: live", len(lv), "order violations", len(bad_order), bad_order[:8])
        break
