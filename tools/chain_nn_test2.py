import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
from helpers import load_pkg
from oracle_lib import Oracle
pkg = load_pkg()
n, K, steps = 1000, 3000, 16
rng = np.random.default_rng(n + K)
src = (rng.integers(0, min(n, 4), K)).astype(np.uint32)
dst = rng.integers(0, 1000, K).astype(np.uint32)
ops = np.stack([src, dst, np.ones(K, np.uint32)], 1).astype(np.uint32)
def run(L, lo=1616):
    e = pkg.PCSR(n); o = Oracle(n)
    for k, v in dict(chain=2, chain_steps=steps, small_batch=0).items(): e.set_option(k, v)
    e.apply(ops[:lo]); o.apply(ops[:lo])
    e.apply(ops[lo:L]); o.apply(ops[lo:L])
    ei, en = e.state(); oi, on = o.state()
    s = e.stats()
    bad = np.nonzero(en[:, 2] != on[:, 2])[0]
    print(L, "items", "ok" if np.array_equal(ei, oi) else "MISMATCH", "nn diff at", bad.tolist(), (en[bad, 2].astype(np.int64) - on[bad, 2].astype(np.int64)).tolist(),
          {k: s[k] for k in ("rounds", "chained", "rollbacks", "duplicates")}, flush=True)
    return len(bad)
lo_, hi_ = 1617, 2837
if run(hi_) == 0:
    print("no repro with the split batches")
else:
    while hi_ - lo_ > 1:
        mid = (lo_ + hi_) // 2
        if run(mid): hi_ = mid
        else: lo_ = mid
    print("first bad prefix ends at", hi_, "op", ops[hi_ - 1].tolist())
