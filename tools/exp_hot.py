#!/usr/bin/env python3
"""Scheduler experiments on the two regimes that stress it (one GPU, small enough to sweep options in seconds):
  crit : config #4's permuted partition scaled by 1/8 — n = 156 250, 1.5625 M-edge core, 156 K inserts; the array sits at
         ~0.8 density, every level of the calibrator tree close to its bound, so leaf overflows cascade into big windows
  zipf : config #5's stream shape on the config #2 graph scaled down — n = 2^18, 2.5 M-edge core, 250 K Zipf(1.2) inserts
usage: python tools/exp_hot.py [crit|zipf|both] [key=value ...]  ('/'-separated values sweep: region_slots=4096/512)"""
import itertools
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import load_pkg, load_streams  # noqa: E402

pkg, st = load_pkg(), load_streams()
which = sys.argv[1] if len(sys.argv) > 1 else "both"
sweeps = {}
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    sweeps[k] = [int(x) for x in v.split("/")]
check = os.environ.get("EXP_CHECK", "1") == "1"


def workloads():
    if which in ("crit", "both"):
        n, sc = 156250, 21
        s, d = st.rmat_edges_folded(n, sc, 12_500_000, seed=1)
        s, d = st.permute_labels(s, 1_250_000), st.permute_labels(d, 10_000_000)  # (any fixed relabelling)
        m = s < n
        core = st.adds(s[m] % n, d[m])
        s2, d2 = st.rmat_edges_folded(n, sc, 1_250_000, seed=2)
        s2, d2 = st.permute_labels(s2, 1_250_000), st.permute_labels(d2, 10_000_000)
        m2 = s2 < n
        yield "crit", n, core, st.adds(s2[m2] % n, d2[m2])
    if which in ("zipf", "both"):
        n, sc = 1 << 18, 18
        s, d = st.rmat_edges(sc, 2_500_000, seed=1)
        zs = st.zipf_sources(n, 250_000, seed=4, alpha=1.2)
        zd = st.uniform_ints(11, 250_000, n)
        yield "zipf", n, st.adds(s, d), st.adds(zs, zd)


for name, n, core, upd in workloads():
    ref_state = None
    keys = list(sweeps)
    for combo in itertools.product(*[sweeps[k] for k in keys]) if keys else [()]:
        e = pkg.PCSR(n)
        for k, v in zip(keys, combo):
            e.set_option(k, v)
        e.apply(core)
        if os.environ.get("EXP_DIAG"):
            e.set_option("diag", 1)
        if os.environ.get("EXP_PROF"):
            e.set_option("profile", 1)
        s0 = e.stats()
        t0 = time.perf_counter()
        e.apply(upd)
        wall = (time.perf_counter() - t0) * 1e3
        s1 = e.stats()
        d = {k: s1[k] - s0[k] for k in ("rounds", "committed", "planned", "exclusive_ops", "rollbacks", "wasted_rounds", "big_redistributes", "round_syncs")}
        ok = ""
        if check:
            items, nodes = e.state()
            if ref_state is None:
                from oracle_lib import Oracle
                o = Oracle(n)
                o.apply(core)
                o.apply(upd)
                ref_state = (o.geometry(), *o.state())
                o.close()
            ok = "bit-exact" if (tuple(e.geometry()) == tuple(ref_state[0]) and np.array_equal(items, ref_state[1]) and np.array_equal(nodes, ref_state[2])) else "MISMATCH"
        print(f"{name} {dict(zip(keys, combo))}: core {len(core)} N={s1['N']} | {len(upd)} updates in {s1['last_batch_ms']:.1f} ms device ({wall:.1f} wall) = "
              f"{len(upd) / s1['last_batch_ms'] / 1e3:.2f} M/s | rounds {d['rounds']} (+{d['wasted_rounds']} wasted) commits/round {d['committed'] / max(d['rounds'], 1):.0f} "
              f"replan {d['planned'] / max(d['committed'], 1):.2f} excl {d['exclusive_ops']} bigrb {d['big_redistributes']} rollbacks {d['rollbacks']} syncs {d['round_syncs']} {ok}", flush=True)
        if os.environ.get("EXP_PROF") and s1["prof_launches"]:
            L = s1["prof_launches"]
            print(f"   per launch (us): plan {s1['prof_plan_ms'] / L * 1e3:.1f} check {s1['prof_check_ms'] / L * 1e3:.1f} apply+big {s1['prof_apply_ms'] / L * 1e3:.1f} "
                  f"compact {s1['prof_compact_ms'] / L * 1e3:.1f} over {L} launches", flush=True)
        e.close()
