"""Full-size runs of the two scheduler stress regimes, with option sweeps and the scheduler's diagnostics:
  zipf : config #5's stream shape on the config #2 graph (1 M Zipf(1.2) inserts, 18 % into vertex 0)
  crit : config #4, partition 3 (permuted labels) alone: its 12.5 M-edge core, its share of the 10 M inserts
  c2   : config #2 itself (1 M RMAT inserts)
usage: python tools/exp_diag.py zipf,crit [option=v1/v2 ...]   ('/'-separated values are swept; EXP_DIAG=1: diag lines on stderr;
EXP_CHECK=1: the first combination of every workload is compared with the oracle, the others with the first)"""
import hashlib
import itertools
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
from helpers import load_pkg, load_streams

pkg, st = load_pkg(), load_streams()
which = (sys.argv[1] if len(sys.argv) > 1 else "zipf,crit").split(",")
sweeps = {}
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    sweeps[k] = [int(x) for x in v.split("/")]
KEYS = ("rounds", "committed", "planned", "exclusive_ops", "rollbacks", "wasted_rounds", "big_redistributes", "round_syncs", "double_calls")
check = os.environ.get("EXP_CHECK", "1") == "1"
reps = int(os.environ.get("EXP_REPS", "2"))


def dig(e):
    items, nodes = e.state()
    h = hashlib.sha256()
    h.update(np.array(e.geometry(), np.int64).tobytes())
    h.update(items.tobytes())
    h.update(nodes.tobytes())
    return h.hexdigest()


def run(name, n, core, upd):
    e = pkg.PCSR(n)
    e.apply(core)
    e.snapshot()
    ref = None
    keys = list(sweeps)
    for combo in itertools.product(*[sweeps[k] for k in keys]) if keys else [()]:
        for k, v in zip(keys, combo):
            e.set_option(k, v)
        best = None
        prof = None
        for rep in range(reps):
            e.restore()
            e.set_option("diag", 1 if rep == reps - 1 and os.environ.get("EXP_DIAG", "0") == "1" else 0)
            a = e.stats()
            e.apply(upd)
            b = e.stats()
            d = {k: b[k] - a[k] for k in KEYS}
            if best is None or b["last_batch_ms"] < best[0]:
                best = (b["last_batch_ms"], d)
                L = b["prof_launches"] - a["prof_launches"]
                prof = None if not L else tuple((b[k] - a[k]) / L * 1e3 for k in ("prof_plan_ms", "prof_check_ms", "prof_apply_ms", "prof_compact_ms")) + (L,)
        ok = ""
        if check:
            g = dig(e)
            if ref is None:
                from oracle_lib import Oracle
                o = Oracle(n)
                o.apply(core)
                o.apply(upd)
                oi, on = o.state()
                h = hashlib.sha256()
                h.update(np.array(o.geometry(), np.int64).tobytes())
                h.update(oi.tobytes())
                h.update(on.tobytes())
                ref = h.hexdigest()
                o.close()
            ok = "bit-exact" if g == ref else "MISMATCH"
        ms, d = best
        print(f"{name} {dict(zip(keys, combo))}: {len(upd)} updates {ms:.2f} ms = {len(upd) / ms / 1e3:.2f} M/s rounds {d['rounds']}+{d['wasted_rounds']}w commits/round "
              f"{d['committed'] / max(d['rounds'], 1):.0f} replan {d['planned'] / max(d['committed'], 1):.2f} excl {d['exclusive_ops']} "
              f"bigrb {d['big_redistributes']} rollbacks {d['rollbacks']} syncs {d['round_syncs']} dbl {d['double_calls']} {ok}", flush=True)
        if prof:
            print(f"     per launch set (us): plan(+sort) {prof[0]:.1f} check {prof[1]:.1f} apply(+chain) {prof[2]:.1f} compact {prof[3]:.1f} over {prof[4]} sets", flush=True)
    e.close()


if "zipf" in which or "c2" in which:
    n = 1 << 20
    s, d = st.rmat_edges(20, 10_000_000, seed=1)
    core = st.adds(s, d)
    if "c2" in which:
        s2, d2 = st.rmat_edges(20, 1_000_000, seed=2)
        run("c2", n, core, st.adds(s2, d2))
    if "zipf" in which:
        zs = st.zipf_sources(n, 1_000_000, seed=4, alpha=1.2)
        zd = st.uniform_ints(11, 1_000_000, n)
        run("zipf", n, core, st.adds(zs, zd))
if "crit" in which:
    N4, P4, part = 10_000_000, 8, 3
    ps = N4 // P4

    def sub(s, d):
        s, d = st.permute_labels(s, N4), st.permute_labels(d, N4)
        m = np.minimum(s // np.uint32(ps), P4 - 1) == part
        return st.adds(s[m] - np.uint32(part * ps), d[m])
    cs, cd = st.rmat_edges_folded(N4, 24, 100_000_000, seed=1)
    core = sub(cs, cd)
    del cs, cd
    us, ud = st.rmat_edges_folded(N4, 24, 10_000_000, seed=2)
    run("crit", ps, core, sub(us, ud))
