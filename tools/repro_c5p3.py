#!/usr/bin/env python3
"""config #5 partition 3 (permuted) at full size under option variants, against the oracle (bisecting a parity failure).
usage: python tools/repro_c5p3.py "k=v,k=v" "k=v" ...   (each argument one variant; "" = defaults)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
from helpers import load_pkg, load_streams, digest
from oracle_lib import Oracle
pkg, st = load_pkg(), load_streams()
N4, P4, part = 10_000_000, 8, 3
ps = N4 // P4
def sub(s, d):
    s, d = st.permute_labels(s, N4), st.permute_labels(d, N4)
    m = np.minimum(s // np.uint32(ps), P4 - 1) == part
    return st.adds(s[m] - np.uint32(part * ps), d[m])
t0 = time.time()
cs, cd = st.rmat_edges_folded(N4, 24, 100_000_000, seed=1)
core = sub(cs, cd); del cs, cd
zupd = sub(st.zipf_sources(N4, 10_000_000, seed=4, alpha=1.2), st.uniform_ints(11, 10_000_000, N4))
print(f"streams {time.time() - t0:.0f} s: core {len(core)} zipf {len(zupd)}", flush=True)
o = Oracle(ps)
o.apply(core)
want_core = digest(*o.state(), o.geometry())
o.apply(zupd)
want = digest(*o.state(), o.geometry())
oi, on = o.state()
print(f"oracle done {time.time() - t0:.0f} s", flush=True)
for spec in (sys.argv[1:] or [""]):
    e = pkg.PCSR(ps)
    for kv in [x for x in spec.split(",") if x]:
        k, v = kv.split("=")
        e.set_option(k, int(v))
    e.apply(core)
    okc = digest(*e.state(), e.geometry()) == want_core
    a = e.stats()
    e.apply(zupd)
    b = e.stats()
    ei, en = e.state()
    ok = digest(ei, en, e.geometry()) == want
    bad = np.nonzero((ei != oi).any(1))[0]
    print(f"variant [{spec}]: core {'ok' if okc else 'BAD'}, zipf {'ok' if ok else 'BAD'} ({len(bad)} slots differ{', first ' + str(bad[:4]) + ' last ' + str(bad[-2:]) if len(bad) else ''}); "
          f"{b['last_batch_ms']:.0f} ms, rounds {b['rounds'] - a['rounds']}, excl {b['exclusive_ops'] - a['exclusive_ops']}, bigred {b['big_redistributes'] - a['big_redistributes']}, rollbacks {b['rollbacks'] - a['rollbacks']}", flush=True)
    e.close()
