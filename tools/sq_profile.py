#!/usr/bin/env python3
"""Workload for instruction-level counters of the round kernels (run under `rocprofv3 --pmc ...`): the config #2 core
(bulk-built: the load itself is not what is looked at), then `reps` batches of 1 M RMAT inserts through the exact path."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from helpers import load_pkg, load_streams  # noqa: E402

pkg, st = load_pkg(), load_streams()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
s, d = st.rmat_edges(20, 10_000_000, seed=1)
n = int(max(s.max(), d.max())) + 1
e = pkg.PCSR(n)
e.bulk_build(st.adds(s, d))
e.snapshot()
us, ud = st.rmat_edges(20, 1_000_000, seed=2)
upd = st.adds(us, ud)
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    e.set_option(k, int(v))
for r in range(reps):
    e.restore()
    e.apply(upd)
    b = e.stats()
    print(f"rep {r}: {b['last_batch_ms']:.2f} ms rounds {b['rounds']}", flush=True)
