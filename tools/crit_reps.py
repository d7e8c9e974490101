#!/usr/bin/env python3
"""One config #4 partition (partition 3, permuted labels) alone, as each of 8 GPUs runs it: restore + its 1.25 M inserts, repeated;
device time of every repetition (the scheduler's adaptive state carries over from one batch to the next)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from helpers import load_pkg, load_streams  # noqa: E402

pkg, st = load_pkg(), load_streams()
N4, P4, part = 10_000_000, 8, int(os.environ.get("PART", "3"))
ps = N4 // P4


def sub(s, d):
    s, d = st.permute_labels(s, N4), st.permute_labels(d, N4)
    m = np.minimum(s // np.uint32(ps), P4 - 1) == part
    return st.adds(s[m] - np.uint32(part * ps), d[m])


cs, cd = st.rmat_edges_folded(N4, 24, 100_000_000, seed=1)
core = sub(cs, cd)
del cs, cd
us, ud = st.rmat_edges_folded(N4, 24, 10_000_000, seed=2)
upd = sub(us, ud)
e = pkg.PCSR(ps)
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    e.set_option(k, int(v))
e.apply(core)
e.snapshot()
for rep in range(6):
    e.restore()
    a = e.stats()
    e.apply(upd)
    b = e.stats()
    print(f"rep {rep}: {b['last_batch_ms']:.2f} ms = {len(upd) / b['last_batch_ms'] / 1e3:.1f} M/s rounds {b['rounds'] - a['rounds']}+{b['wasted_rounds'] - a['wasted_rounds']}w "
          f"rollbacks {b['rollbacks'] - a['rollbacks']}", flush=True)
