import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
from helpers import load_pkg
from oracle_lib import Oracle
pkg = load_pkg()
for n, K, steps in [(1000, 3000, 16)]:
    rng = np.random.default_rng(n + K)
    src = (rng.integers(0, min(n, 4), K)).astype(np.uint32)
    dst = rng.integers(0, 1000, K).astype(np.uint32)
    ops = np.stack([src, dst, np.ones(K, np.uint32)], 1).astype(np.uint32)
    e = pkg.PCSR(n); o = Oracle(n)
    for k, v in dict(chain=2, chain_steps=steps, small_batch=0).items(): e.set_option(k, v)
    e.apply(ops); o.apply(ops)
    ei, en = e.state(); oi, on = o.state()
    s = e.stats()
    print(n, K, steps, "items", "ok" if np.array_equal(ei, oi) else "MISMATCH", "nn eng", en[:4, 2].tolist(), "ora", on[:4, 2].tolist(),
          {k: s[k] for k in ("rounds", "chained", "rollbacks", "duplicates", "exclusive_ops", "double_calls")}, flush=True)
