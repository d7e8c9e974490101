"""Kernel-level anatomy of the window rebalance: python3 tools/rebalance_profile.py [window_fraction_denominator]
(run under `rocprofv3 --kernel-trace --stats`); rebalances the leftmost N/den slots of the config #2 graph 20 times."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from helpers import load_pkg, load_streams
pkg, st = load_pkg(), load_streams()
den = int(sys.argv[1]) if len(sys.argv) > 1 else 2
s, d = st.rmat_edges(20, 10_000_000, seed=1)
e = pkg.PCSR(1 << 20)
e.bulk_build(st.adds(s, d))
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    e.set_option(k, int(v))
N = e.geometry()[0]
ms = e.bench_rebalance(N // den, 20)
print("window", N // den, "ms per call", ms, "alg GB/s", 24.0 * (N // den) / (ms * 1e-3) / 1e9)
