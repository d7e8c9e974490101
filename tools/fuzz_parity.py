#!/usr/bin/env python3
"""Randomised parity campaign on the GPU: random graph sizes, stream shapes, scheduler options and batch splits, each
compared slot by slot with the oracle.  usage: python tools/fuzz_parity.py [cases] [seed0] [max_seconds]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import load_pkg, load_streams  # noqa: E402
from oracle_lib import Oracle  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
budget = float(sys.argv[3]) if len(sys.argv) > 3 else 400.0
pkg, st = load_pkg(), load_streams()
t_start = time.time()
bad = 0
for c in range(cases):
    if time.time() - t_start > budget:
        print(f"time budget reached after {c} cases")
        break
    rng = np.random.default_rng(seed0 + c)
    n = int(rng.choice([7, 60, 1000, 4096, 65536]))
    m = int(rng.choice([20_000, 60_000, 150_000, 300_000]))
    kind = str(rng.choice(["mixed", "zipf", "hub", "runs", "dupes", "rmat"]))
    lock = bool(rng.integers(0, 2))
    if kind == "mixed":
        ops = st.random_stream(n, m, seed=int(rng.integers(1 << 30)), p_delete=float(rng.choice([0.0, 0.3, 0.6])))
    elif kind == "zipf":
        src = st.zipf_sources(n, m, seed=int(rng.integers(1 << 30)))
        ops = np.stack([src, st.uniform_ints(int(rng.integers(1 << 30)), m, 1 << 20), rng.integers(0, 3, m)], 1).astype(np.uint32)
    elif kind == "hub":
        h = int(rng.integers(0, n))
        ops = np.stack([np.full(m, h), st.uniform_ints(int(rng.integers(1 << 30)), m, 1 << 16), rng.integers(0, 4, m)], 1).astype(np.uint32)
    elif kind == "runs":
        d = np.arange(m) if rng.integers(0, 2) else np.arange(m, 0, -1)
        ops = np.stack([rng.integers(max(n - 3, 0), n, m), d, np.ones(m)], 1).astype(np.uint32)
    elif kind == "dupes":
        ops = np.stack([rng.integers(0, n, m), rng.integers(0, 40, m), rng.integers(0, 3, m)], 1).astype(np.uint32)
    else:
        sc = max(int(np.log2(n)), 3)
        s, d = st.rmat_edges(sc, m, seed=int(rng.integers(1 << 30)))
        ops = st.adds(s % n, d)
        ops[rng.random(m) < 0.25, 2] = 0
    opts = {}
    if rng.integers(0, 2):
        opts = dict(opt_horizon=int(rng.choice([256, 1024, 6144, 16384])), region_slots=int(rng.choice([64, 1024, 4096])),
                    epoch_ops=int(rng.choice([4096, 65536, 1 << 20])), small_batch=int(rng.choice([0, 256, 5000])))
    if rng.integers(0, 2):  # round-2 knobs: width in multiples of a (pretended) chip-full, barrier, big windows, epochs
        opts.update(resident_waves=int(rng.choice([0, 512, 6144])), big_window=int(rng.choice([2048, 8192, 32768])),
                    soft_barrier=int(rng.choice([0, 1024, 1 << 30])), rb_inplace_min=int(rng.choice([0, 2048, 1 << 19])),
                    epoch_short=int(rng.choice([512, 16384])), epoch_grow_after=int(rng.choice([1, 2, 8])))
    if rng.integers(0, 3) == 0:  # diagnostics on (the *_x instantiations of the round kernels)
        opts.update(diag=int(rng.choice([0, 1])))
    if rng.integers(0, 3) == 0:  # round-3 policies: epoch length / region width that follow the rollback frequency
        opts.update(epoch_adapt=int(rng.choice([0, 2, 8, 32])), region_rare=int(rng.choice([0, 4096, 16384])),
                    region_rare_dist=int(rng.choice([1, 4096, 65536])), region_rare_cpr=int(rng.choice([0, 64, 1536])),
                    region_rare_calm=int(rng.choice([1, 8])))
    if rng.integers(0, 2):  # round-4: which o_check, when windows are queued for workgroups, how often the host looks
        opts.update(check_lanes=int(rng.choice([-1, 0, 1])), big_min=int(rng.choice([64, 512, 512])), rounds_per_sync=int(rng.choice([2, 8, 32])),
                    rb_defer_table=int(rng.choice([64, 4096, 1 << 22])))  # (64: every resize builds its position table inside the scatter launch)
    if rng.integers(0, 5) == 0:
        opts["mode"] = 0
    eng, o = pkg.PCSR(n, lock_search=lock), Oracle(n, lock_search=lock)
    for k, v in opts.items():
        eng.set_option(k, v)
    pos = 0
    snap_at = int(rng.integers(0, m)) if rng.integers(0, 3) == 0 else -1  # incremental snapshot / restore somewhere in the stream
    snapped = None
    while pos < m:
        step = int(rng.choice([1, 17, 300, 5000, 100_000]))
        if snapped is None and snap_at >= 0 and pos >= snap_at:
            eng.snapshot()
            snapped = pos
        eng.apply(ops[pos:pos + step])
        pos += step
    if snapped is not None:  # back to the snapshot: must equal the oracle at that point; then the rest of the stream again
        eng.restore()
        o.apply(ops[:snapped])
        ei, en = eng.state()
        oi, on = o.state()
        if not (eng.geometry() == o.geometry() and np.array_equal(ei, oi) and np.array_equal(en, on)):
            print(f"case {c}: MISMATCH after restore to {snapped}", flush=True)
            bad += 1
        eng.apply(ops[snapped:])
        o.apply(ops[snapped:])
    else:
        o.apply(ops)
    ei, en = eng.state()
    oi, on = o.state()
    ok = eng.geometry() == o.geometry() and np.array_equal(ei, oi) and np.array_equal(en, on)
    print(f"case {c}: n={n} m={m} kind={kind} lock={lock} opts={opts} -> {'ok' if ok else 'MISMATCH'}", flush=True)
    bad += 0 if ok else 1
    eng.close()
    o.close()
print(f"done: {bad} mismatches")
sys.exit(1 if bad else 0)
