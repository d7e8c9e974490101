"""Workload for tools/sim_asan.sh: the real kernels under the CPU emulator built with AddressSanitizer (speculative rounds with
rollbacks, a hub stream with the in-launch position table, strict rounds, whole-array rebalances), each compared with the oracle."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
from helpers import load_pkg, load_streams
from oracle_lib import Oracle
pkg, st = load_pkg(), load_streams()
lib = pkg.load_library(os.environ.get("PPCSR_SIM_ASAN", "/tmp/libppcsr_sim_asan.so"))
def make(n, **opts):
    e = pkg.PCSR(n, lock_search=True, lib=lib)
    base = dict(mode=1, opt_horizon=256, epoch_ops=1024, region_slots=64, max_horizon=32, min_horizon=4, init_horizon=8, rounds_per_sync=2, small_batch=0, big_grid=2, big_min=256, big_window=131072)
    base.update(opts)
    for k, v in base.items(): e.set_option(k, v)
    return e
for name, n, m, pd, opts in [("mixed", 200, 6000, 0.3, {}), ("hub", 40, 5000, 0.1, dict(rb_defer_table=64)), ("strict", 300, 4000, 0.3, dict(mode=0))]:
    ops = st.random_stream(n, m, seed=5, p_delete=pd)
    if name == "hub":
        ops[:, 0] = np.where(np.arange(m) % 2 == 0, 3, ops[:, 0])
    e, o = make(n, **opts), Oracle(n)
    for lo in range(0, m, 1500):
        e.apply(ops[lo:lo + 1500]); o.apply(ops[lo:lo + 1500])
    ei, en = e.state(); oi, on = o.state()
    print(name, "same" if (np.array_equal(ei, oi) and np.array_equal(en, on)) else "DIFF", e.stats()["rollbacks"], flush=True)
    N = e.geometry()[0]
    e.bench_rebalance(N, 1); o.debug_redistribute(0, N)
    ei, en = e.state(); oi, on = o.state()
    print(name, "rebalance", "same" if np.array_equal(ei, oi) else "DIFF", flush=True)
    e.close(); o.close()
print("asan run finished")
