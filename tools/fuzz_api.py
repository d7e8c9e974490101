#!/usr/bin/env python3
"""Randomised API-sequence campaign on the GPU: add_edge / remove_edge / add_node / edge_exists / get_neighbourhood /
batches / PPPCSR routing, every query and the final state compared with the oracle.
usage: python tools/fuzz_api.py [cases] [seed0] [max_seconds]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import load_pkg, load_streams  # noqa: E402
from oracle_lib import Oracle, OraclePPPCSR  # noqa: E402


def run_case(pkg, st, seed):
    """one random API sequence; returns (ok, description)"""
    rng = np.random.default_rng(seed)
    lock = bool(rng.integers(0, 2))
    if rng.integers(0, 4) == 0:  # PPPCSR: routing + per-partition states
        n = int(rng.choice([10, 1003, 5000]))
        P = int(rng.choice([2, 3, 8]))
        e = pkg.PPPCSR(n, lock_search=lock, numDomain=1, partitionsPerDomain=P)
        o = OraclePPPCSR(n, lock_search=lock, num_domains=1, parts_per_domain=P)
        ops = st.random_stream(n, int(rng.choice([2000, 30000])), seed=int(rng.integers(1 << 30)), p_delete=0.3)
        e.apply(ops)
        o.apply(ops)
        ok = True
        for k in range(P):
            ei, en = e.partition(k).state()
            oi, on = o.partition(k).state()
            ok = ok and np.array_equal(ei, oi) and np.array_equal(en, on)
        for _ in range(20):
            s, d = int(rng.integers(0, n)), int(rng.integers(0, 1 << 16))
            part = o.get_partition(s)
            ok = ok and (e.edge_exists(s, d) == o.partition(part).edge_exists(s - o.partition_start(part), d))
        e.close()
        o.close()
        return ok, f"PPPCSR n={n} P={P} lock={lock}"
    if rng.integers(0, 5) == 0:  # non-parity bulk build + exact updates + consumers: edge sets, invariants, BFS, PageRank
        from helpers import check_pma_invariants, edge_view, reference_consumers
        n = int(rng.choice([3, 300, 20000]))
        m = int(rng.choice([50, 5000, 200000]))
        adds = st.random_stream(n, m, seed=int(rng.integers(1 << 30)), p_delete=0.1)
        adds[:, 1] %= max(n, 2) + 3  # a few dests beyond n
        e, o = pkg.PCSR(n, lock_search=lock), Oracle(n, lock_search=lock)
        e.bulk_build(adds)
        o.apply(adds[adds[:, 2] != 0])
        upd = st.random_stream(n, int(rng.choice([10, 3000])), seed=int(rng.integers(1 << 30)), p_delete=0.4)
        upd[:, 1] %= max(n, 2) + 3
        e.apply(upd)
        o.apply(upd)
        ei, en = e.state()
        check_pma_invariants(ei, en)
        ok = e.check_invariants() == 0
        for a, b in zip(edge_view(ei, en), edge_view(*o.state())):
            ok = ok and np.array_equal(a, b)
        vals = (rng.random(n) * 3).astype(np.float32)
        start = int(rng.integers(0, n))
        lv, pr = reference_consumers(o, start, vals)
        ok = ok and np.array_equal(e.bfs(start), lv) and e.pagerank(vals).tobytes() == pr.tobytes()
        e.close()
        o.close()
        return ok, f"bulk_build n={n} m={m} lock={lock}"
    n = int(rng.choice([1, 5, 64, 2000]))
    e, o = pkg.PCSR(n, lock_search=lock), Oracle(n, lock_search=lock)
    if rng.integers(0, 2):
        e.set_option("small_batch", int(rng.choice([0, 8, 256])))
    ok = True
    undefined = False
    steps = int(rng.choice([300, 1500]))
    for _ in range(steps):
        if o.geometry()[2] == 0:  # H == 0: the array shrank to ONE leaf, where the reference's full-leaf rebalance reads
            undefined = True      # past the end of its array (undefined behaviour; the engine refuses loudly): stop here
            break
        cur_n = o.get_n()
        r = rng.random()
        if r < 0.45:
            s, d, v = int(rng.integers(0, cur_n + 2)), int(rng.integers(0, 200)), int(rng.integers(0, 4))
            e.add_edge(s, d, v)
            o.add_edge(s, d, v)
        elif r < 0.65:
            s, d = int(rng.integers(0, cur_n)), int(rng.integers(0, 200))
            e.remove_edge(s, d)
            o.remove_edge(s, d)
        elif r < 0.70:
            e.add_node()
            o.add_node()
        elif r < 0.80:
            s, d = int(rng.integers(0, cur_n)), int(rng.integers(0, 200))
            ok = ok and (e.edge_exists(s, d) == o.edge_exists(s, d))
        elif r < 0.88:
            s = int(rng.integers(0, cur_n))
            ok = ok and np.array_equal(e.get_neighbourhood(s), o.get_neighbourhood(s))
        else:
            m = int(rng.choice([3, 40, 700]))
            ops = st.random_stream(cur_n, m, seed=int(rng.integers(1 << 30)), p_delete=0.35)
            ops[:, 1] %= 200
            e.apply(ops)
            o.apply(ops)
    ei, en = e.state()
    oi, on = o.state()
    ok = ok and e.geometry() == o.geometry() and e.get_n() == o.get_n() and np.array_equal(ei, oi) and np.array_equal(en, on)
    sst = e.stats()
    desc = (f"PCSR n0={n} -> n={o.get_n()} steps={steps} lock={lock}{' (stopped at a one-leaf array)' if undefined else ''}"
            f" sequential-regime events {sst.get('narrow_lost', 0)}, regular again: {bool(sst.get('narrow', 1))}")
    e.close()
    o.close()
    return ok, desc


def run_case_tolerant(pkg, st, seed):
    """run_case, except that the engine's loud refusal of the one-leaf regime (where the reference itself reads out of
    bounds, so there is nothing defined to compare with) ends the case instead of failing it"""
    try:
        return run_case(pkg, st, seed)
    except pkg.PpcsrError as ex:
        if "one-leaf array" in str(ex):
            return True, "stopped: array shrank to one leaf mid-batch (reference UB, refused loudly)"
        raise


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    budget = float(sys.argv[3]) if len(sys.argv) > 3 else 300.0
    pkg, st = load_pkg(), load_streams()
    t_start = time.time()
    bad = 0
    for c in range(cases):
        if time.time() - t_start > budget:
            print(f"time budget reached after {c} cases")
            break
        ok, desc = run_case_tolerant(pkg, st, seed0 + c)
        print(f"case {c}: {desc} -> {'ok' if ok else 'MISMATCH'}", flush=True)
        bad += 0 if ok else 1
    print(f"done: {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
