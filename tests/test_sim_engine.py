"""CPU: the engine's ACTUAL kernel + scheduler source, compiled for the fiber SIMT emulator
(tests/hostsim, test infrastructure), replayed against the golden fixtures and the oracle.  This is a
debugging aid for the kernel logic on machines without a GPU; it proves nothing about the GPU build
(the `-m gpu` tests do) and the product never loads this library."""
import os
import subprocess

import numpy as np
import pytest

from helpers import GOLDEN_SINGLE, ROOT, digest, golden, load_pkg, replay_golden
from oracle_lib import Oracle

SIM_DIR = os.path.join(ROOT, "tests", "hostsim")
SIM_SO = os.path.join(SIM_DIR, "libppcsr_sim.so")
CSRC = os.path.join(ROOT, "parallel-packed-csr_amd", "csrc")


def build_sim():
    srcs = [os.path.join(SIM_DIR, "ppcsr_sim.cpp"), os.path.join(SIM_DIR, "sim_runtime.cpp"), os.path.join(SIM_DIR, "sim_xchg.cpp")]
    deps = srcs + [os.path.join(SIM_DIR, "sim_runtime.h")] + [os.path.join(CSRC, f) for f in os.listdir(CSRC)
                                                             if f.endswith((".h", ".cc"))]
    if os.path.exists(SIM_SO) and os.path.getmtime(SIM_SO) >= max(os.path.getmtime(d) for d in deps):
        return
    subprocess.run(["g++", "-O2", "-g", "-std=c++17", "-ffp-contract=off", "-Wno-unknown-pragmas", "-fPIC", "-shared",
                    "-I" + SIM_DIR, "-I" + CSRC] + srcs + ["-lrt", "-o", SIM_SO], check=True)


@pytest.fixture(scope="module")
def sim():
    build_sim()
    pkg = load_pkg()
    lib = pkg.load_library(SIM_SO)

    def make(n, lock=True, **opts):
        e = pkg.PCSR(n, lock_search=lock, lib=lib)
        e.set_option("mode", opts.get("mode", 0))
        e.set_option("opt_horizon", opts.get("opt_horizon", 64))
        e.set_option("epoch_ops", opts.get("epoch_ops", 1024))
        e.set_option("region_slots", opts.get("region_slots", 64))
        e.set_option("max_horizon", opts.get("max_horizon", 32))
        e.set_option("min_horizon", opts.get("min_horizon", 4))
        e.set_option("init_horizon", opts.get("init_horizon", 8))
        e.set_option("rounds_per_sync", opts.get("rounds_per_sync", 2))
        e.set_option("small_batch", opts.get("small_batch", 0))  # keep small streams on the scheduler under test
        e.set_option("big_grid", opts.get("big_grid", 2))        # (every emulated workgroup of o_big is 1024 fibers)
        e.set_option("big_min", opts.get("big_min", 512))
        e.set_option("big_window", opts.get("big_window", 131072))
        if "rb_defer_table" in opts:
            e.set_option("rb_defer_table", opts["rb_defer_table"])
        return e
    return make


def _same(eng, o, label=""):
    assert eng.geometry() == o.geometry(), label
    ei, en = eng.state()
    oi, on = o.state()
    np.testing.assert_array_equal(en, on, err_msg=label + " nodes")
    np.testing.assert_array_equal(ei, oi, err_msg=label + " items")
    assert eng.check_invariants() == 0


def test_sim_small_inserts(sim, streams):
    ops = streams.random_stream(200, 3000, seed=1)
    e, o = sim(200), Oracle(200)
    e.apply(ops)
    o.apply(ops)
    _same(e, o)


@pytest.mark.parametrize("defer_min", [1 << 22, 64])
def test_sim_mixed_with_resizes(sim, streams, defer_min):
    """(defer_min = 64: double_list / half_list build their position table inside the scatter launch, tiles from the top down)"""
    a = streams.random_stream(40, 6000, seed=2)
    d = a.copy()
    d[:, 2] = 0
    ops = np.concatenate([a, d[::-1]])
    e, o = sim(40, rb_defer_table=defer_min), Oracle(40)
    for lo in range(0, len(ops), 1500):
        e.apply(ops[lo:lo + 1500])
        o.apply(ops[lo:lo + 1500])
        _same(e, o, f"after {lo + 1500}")
    so, se = o.stats(), e.stats()
    assert se["redistribute_calls"] == so["redistribute_calls"] and se["redistribute_slots"] == so["redistribute_slots"]
    assert se["double_calls"] == so["double_calls"] and se["half_calls"] == so["half_calls"]


@pytest.mark.parametrize("name", ["add_node_empty_then_edges", "hub_1e4_insert_then_delete"])
def test_sim_golden_small(sim, name):
    eng = replay_golden(lambda n, lock: sim(n, lock), name, check_every=True)
    assert eng.check_invariants() == 0


@pytest.mark.parametrize("name,region", [("random_2e4_n1000", 128), ("random_2e4_n1000", 16), ("dense_n40_grow_shrink", 64)])
def test_sim_speculative_rounds(sim, name, region):
    """speculative scheduler (mode 1) incl. validation failures -> rollback -> cut / strict replay"""
    eng = replay_golden(lambda n, lock: sim(n, lock, mode=1, opt_horizon=256, epoch_ops=4096, region_slots=region), name)
    st = eng.stats()
    assert eng.check_invariants() == 0
    assert st["rollbacks"] > 0  # the small regions / doublings of these fixtures must exercise the rollback path


@pytest.mark.parametrize("big_min,big_window", [(64, 131072), (128, 1024)])
def test_sim_big_windows_inside_the_round(sim, big_min, big_window):
    """windows above big_min slots are rebalanced by a workgroup of o_big instead of the update's
    own wave; windows above big_window make the update exclusive, and it runs in the middle of the epoch (stamp-validated,
    its slot commits as nothing).  A hub stream drives windows up to the whole array; its delete phase shrinks them again.
    (Every emulated workgroup is 1024 fibers, hence the short stream.)"""
    n = 12
    ins = np.stack([np.zeros(1800, np.uint32), 1 + (np.arange(1800, dtype=np.uint32) * 7919) % 5003, np.ones(1800, np.uint32)], 1)
    other = np.stack([1 + np.arange(400, dtype=np.uint32) % 11, np.arange(400, dtype=np.uint32) * 3 % 97, np.ones(400, np.uint32)], 1)
    dele = ins[::-1][:1200].copy()
    dele[:, 2] = 0
    ops = np.concatenate([ins[:900], other, ins[900:], dele]).astype(np.uint32)
    e = sim(n, True, mode=1, opt_horizon=128, epoch_ops=4096, region_slots=64, big_min=big_min, big_window=big_window, big_grid=1)
    o = Oracle(n)
    for lo in range(0, len(ops), 1300):
        e.apply(ops[lo:lo + 1300])
        o.apply(ops[lo:lo + 1300])
        _same(e, o, f"after {lo + 1300}")
    st = e.stats()
    assert st["exclusive_ops"] > 0


@pytest.mark.parametrize("rps,seed,big_min", [(8, 1, 32), (8, 2, 32), (3, 3, 32), (8, 4, 64)])
def test_sim_queued_windows_between_exclusive_updates(sim, rps, seed, big_min):
    """The launch that rebalances a round's queued windows (o_big) is left out while a stream queues few, and put back by the
    round that needs it (need_big); exclusive updates, finished epochs and such rounds leave launches behind them that must do
    nothing — one of those once marked its round number as served, and the round that took the number afterwards lost its
    window (found by the full-size config #5 partition, never by the small tests).  Hub streams with small windows made
    exclusive (big_window 1024) and windows queued from 32 / 64 slots on (every two-leaf window: the launch goes in and out), several
    rounds per host look."""
    rng = np.random.default_rng(seed)
    n = 40
    m = 2600
    src = np.where(rng.random(m) < 0.8, rng.integers(0, 3, m), rng.integers(0, n, m)).astype(np.uint32)
    ops = np.stack([src, rng.integers(0, 6000, m).astype(np.uint32), (rng.random(m) >= 0.15).astype(np.uint32)], 1).astype(np.uint32)
    e = sim(n, True, mode=1, opt_horizon=64, epoch_ops=512, region_slots=64, big_min=big_min, big_window=1024, big_grid=1, rounds_per_sync=rps)
    o = Oracle(n)
    for lo in range(0, m, 650):
        e.apply(ops[lo:lo + 650])
        o.apply(ops[lo:lo + 650])
        _same(e, o, f"after {lo + 650}")
    st = e.stats()
    assert st["exclusive_ops"] > 0 and st["big_redistributes"] >= 0


@pytest.mark.parametrize("soft", [64, 128, 1 << 30])
def test_sim_soft_barrier_with_ignored_adds(sim, soft):
    """A soft barrier (planned window >= soft_barrier slots) used to be published as key + 1 in the word that carries the
    K_EXCL barrier: when it and everything before it committed, the NEXT update was sent to the exclusive executor whatever
    its kind — and an add with src >= n (accepted input, silently ignored: PCSR.cpp:1375) then indexed nodes[] out of bounds.
    Hub stream (windows climb past the small barrier) with every 2nd op such an add."""
    n = 60
    m = 3000
    hub = np.stack([np.full(m, 3, np.uint32), 1 + (np.arange(m, dtype=np.uint32) * 7919) % 100003, np.ones(m, np.uint32)], 1)
    junk = np.stack([np.full(m, n + 7, np.uint32), np.arange(m, dtype=np.uint32), np.ones(m, np.uint32)], 1)
    ops = np.empty((2 * m, 3), np.uint32)
    ops[0::2] = hub
    ops[1::2] = junk
    e = sim(n, True, mode=1, opt_horizon=8, epoch_ops=4096, region_slots=64)
    e.set_option("soft_barrier", soft)
    o = Oracle(n)
    e.apply(ops)
    o.apply(ops)
    _same(e, o)
    se = e.stats()
    assert se["noops"] == m
    if soft < (1 << 30):  # a barrier alone never makes an update exclusive
        e2 = sim(n, True, mode=1, opt_horizon=8, epoch_ops=4096, region_slots=64)
        e2.set_option("soft_barrier", 1 << 30)
        e2.apply(ops)
        assert se["exclusive_ops"] == e2.stats()["exclusive_ops"]


def test_sim_snapshot_restore_incremental(sim, streams):
    """snapshot() / restore() through dirty tags across batches with rollbacks and doublings (see the GPU test of the same name)"""
    n = 40
    core = streams.random_stream(n, 1500, seed=100, p_delete=0.1)
    rng = np.random.default_rng(0)
    m = 2500
    src = np.where(rng.random(m) < 0.5, rng.integers(0, 4, m), rng.integers(0, n, m)).astype(np.uint32)
    upd = np.stack([src, rng.integers(0, 100000, m).astype(np.uint32), (rng.random(m) >= 0.1).astype(np.uint32)], 1).astype(np.uint32)
    upd2 = streams.random_stream(n, 1500, seed=7, p_delete=0.5)
    e, o = sim(n, True, mode=1, opt_horizon=256, epoch_ops=1024, region_slots=64), Oracle(n)
    e.apply(core)
    o.apply(core)
    e.snapshot()
    for rep in range(3):
        e.restore()
        _same(e, o, f"rep {rep}: after restore")
        which = upd if rep != 1 else upd2
        e.apply(which)
        o2 = o.clone()
        o2.apply(which)
        _same(e, o2, f"rep {rep}: after the batch")
        o2.close()
    assert e.stats()["rollbacks"] > 0 and e.stats()["double_calls"] > 0


def test_sim_speculative_stats_survive_rollback(sim, streams):
    ops = streams.random_stream(1000, 12000, seed=5, p_delete=0.2)
    e, o = sim(1000, mode=1, opt_horizon=256, epoch_ops=2048, region_slots=32), Oracle(1000)
    e.apply(ops)
    o.apply(ops)
    _same(e, o)
    se, so = e.stats(), o.stats()
    for k in ("redistribute_calls", "redistribute_slots", "double_calls", "not_found", "duplicates"):
        assert se[k] == so[k], k


def test_sim_horizon_options_changed_between_batches(sim, streams):
    """the shared plan-record buffer follows max(opt_horizon, max_horizon) whenever either option changes"""
    ops = streams.random_stream(400, 9000, seed=12, p_delete=0.2)
    e, o = sim(400, mode=1, opt_horizon=256, epoch_ops=2048, region_slots=64, max_horizon=64), Oracle(400)
    e.apply(ops[:3000])
    e.set_option("max_horizon", 8)
    e.apply(ops[3000:6000])
    e.set_option("mode", 0)
    e.apply(ops[6000:7500])
    e.set_option("mode", 1)
    e.set_option("opt_horizon", 512)
    e.apply(ops[7500:])
    o.apply(ops)
    _same(e, o)


def test_sim_leaves_the_sequential_regime_when_ranges_are_sane(sim, streams):
    """narrow == 0 (set by add_node after a doubling, here by hand) makes every batch run one update per round; the range
    check that precedes such a batch finds sorted, disjoint, consistent vertex ranges and switches back"""
    ops = streams.random_stream(300, 4000, seed=21, p_delete=0.2)
    e, o = sim(300, mode=1, opt_horizon=128, epoch_ops=2048, region_slots=64), Oracle(300)
    e.apply(ops[:2000])
    e.set_option("search_narrow", 0)
    assert e.stats()["narrow"] == 0
    r0 = e.stats()["rounds"]
    e.apply(ops[2000:])
    o.apply(ops)
    _same(e, o)
    assert e.stats()["narrow"] == 1
    assert e.stats()["rounds"] - r0 < 1000  # (2000 updates one per round would be 2000 rounds)


@pytest.mark.parametrize("defer_min", [1 << 22, 64])
def test_sim_big_window_rebalance(sim, streams, defer_min):
    """the multi-workgroup rebalance (rank scan + exact position table + fused scatter/fill) on whole-array and
    partial windows, against the reference's redistribute() run by the oracle on the same window; defer_min = 64: the position
    table is built inside the scatter launch (published segment by segment, tiles taken from the top of the window down)"""
    ops = streams.random_stream(300, 6000, seed=9)
    e, o = sim(300, rb_defer_table=defer_min), Oracle(300)
    e.apply(ops)
    o.apply(ops)
    N = e.geometry()[0]
    for w in (N, N // 2, N // 8):
        e.bench_rebalance(w, 1)
        o.debug_redistribute(0, w)
        _same(e, o, f"window {w}")
    more = streams.random_stream(300, 2000, seed=10, p_delete=0.3)
    e.apply(more)
    o.apply(more)
    _same(e, o, "updates after a rebalance")


@pytest.mark.parametrize("shape", ["uniform", "dense_left", "dense_right", "dense_middle", "sparse_middle"])
def test_sim_partial_window_rebalance_in_place(sim, streams, shape):
    """partial windows rebalanced inside the array (k_rb_order + k_rb_inplace: tiles held in registers, written once every
    tile whose source they cover has been read), on windows whose elements move left, right, outward and inward, against
    the reference's redistribute() (PCSR.cpp:207-247) run by the oracle on the same window"""
    n = 4096
    base = streams.random_stream(n, 30000, seed=21)
    lo, hi = {"uniform": (0, 0), "dense_left": (0, n // 8), "dense_right": (n // 4, n // 2), "dense_middle": (n // 5, n // 4),
              "sparse_middle": (0, 0)}[shape]
    extra = []
    if hi > lo:  # pile edges onto a stretch of vertices: that part of the window is dense, the rest must make room
        src = streams.uniform_ints(31, 12000, hi - lo, lo)
        extra = [streams.adds(src, streams.uniform_ints(32, 12000, n))]
    ops = np.concatenate([base] + extra)
    # (big_window 2048: every window above it is rebalanced by the host-driven path, i.e. in place, during the loads too)
    e, o = sim(n, True, mode=1, opt_horizon=64, big_window=2048), Oracle(n)
    e.set_option("rb_inplace_min", 2048)
    e.apply(ops)
    o.apply(ops)
    if shape == "sparse_middle":  # empty the middle of the left half: its neighbours move inward
        dele = base[(base[:, 0] >= n // 6) & (base[:, 0] < n // 3)].copy()
        dele[:, 2] = 0
        e.apply(dele)
        o.apply(dele)
    _same(e, o, "before")
    N = e.geometry()[0]
    assert N >= 65536
    for w in (N // 2, N // 4, 8192, 2048):
        e.bench_rebalance(w, 1)
        o.debug_redistribute(0, w)
        _same(e, o, f"{shape}: window {w}")
    more = streams.random_stream(n, 3000, seed=22, p_delete=0.3)
    e.apply(more)
    o.apply(more)
    _same(e, o, "updates after the in-place rebalances")


def test_sim_scan_all_and_queries(sim, streams):
    ops = streams.random_stream(120, 4000, seed=11, p_delete=0.2)
    e, o = sim(120), Oracle(120)
    e.apply(ops)
    o.apply(ops)
    rows, dests = e.scan_all()
    for v in range(120):
        ref = o.get_neighbourhood(v)
        np.testing.assert_array_equal(e.get_neighbourhood(v), ref)
        np.testing.assert_array_equal(dests[int(rows[v]):int(rows[v + 1])], ref)
    for s, d in [(0, 1), (5, 77), (119, 3)]:
        assert e.edge_exists(s, d) == o.edge_exists(s, d)


def test_sim_scan_all_repeated_across_resizes(sim, streams):
    """the bulk scan keeps per-array-size scratch state between calls: scan, grow (double_list), scan, shrink, scan;
    isolated vertices (runs of adjacent sentinels longer than a wave) included"""
    n = 300
    e, o = sim(n), Oracle(n)

    def check():
        rows, dests = e.scan_all()
        for v in range(n):
            np.testing.assert_array_equal(dests[int(rows[v]):int(rows[v + 1])], o.get_neighbourhood(v))
        assert int(rows[n]) == len(dests)

    check()  # empty graph: nothing but sentinels
    src = np.repeat(np.arange(100, 140, dtype=np.uint32), 60)
    dst = np.tile(np.arange(60, dtype=np.uint32), 40)
    adds = np.stack([src, dst, np.ones_like(src)], 1).astype(np.uint32)
    for part in (adds[:700], adds[700:]):
        e.apply(part)
        o.apply(part)
        check()
        check()
    dels = adds.copy()
    dels[:, 2] = 0
    e.apply(dels[:2300])
    o.apply(dels[:2300])
    check()
    assert e.geometry() == o.geometry()


def test_sim_sparse_hub_searches(sim, streams):
    """64-ary bracket narrowing on a hub that is mostly gaps (built, then 95 % deleted), probed through every exit of the
    search, under both schedulers and both lock_search settings"""
    m, n = 2500, 16
    d = streams.uniform_ints(11, m, 1 << 20) * 4 + 2
    hub = np.stack([np.full(m, 5), d, np.ones(m)], 1).astype(np.uint32)
    filler = np.stack([streams.uniform_ints(12, 3000, n), streams.uniform_ints(13, 3000, 1 << 30), np.ones(3000)], 1).astype(np.uint32)
    dele = hub[streams.uniform_ints(14, m, 100) < 95].copy()
    dele[:, 2] = 0
    probe_d = np.concatenate([d[:500] + 1, d[:500], np.arange(1, 301, dtype=np.uint64), d[:300] - 1]).astype(np.uint32)
    probes = np.stack([np.full(len(probe_d), 5), probe_d, np.ones(len(probe_d))], 1).astype(np.uint32)
    probes[1::3, 2] = 0
    ops = np.concatenate([hub, filler, dele, probes])
    for lock, mode in ((True, 1), (False, 0)):
        e, o = sim(n, lock, mode=mode), Oracle(n, lock_search=lock)
        e.apply(ops)
        o.apply(ops)
        assert e.geometry() == o.geometry()
        ei, en = e.state()
        oi, on = o.state()
        np.testing.assert_array_equal(en, on)
        np.testing.assert_array_equal(ei, oi)


def test_sim_slide_off_the_end(sim, streams):
    """slide_right runs off the end of the array (PCSR.cpp:347-351): the reference slides back and then to the LEFT
    (slide_left, PCSR.cpp:360-390, + insert():541-544); oracle counts 2 slide_left calls on this stream"""
    from helpers import slide_off_end_stream
    ops = slide_off_end_stream()
    more = streams.random_stream(4096, 1500, seed=77, p_delete=0.2)
    more[:, 0] = 4096 - 1 - (more[:, 0] % 40)  # keep working at the end of the array
    for lock, mode in ((True, 1), (False, 0), (False, 1)):
        e, o = sim(4096, lock, mode=mode), Oracle(4096, lock_search=lock)
        for part in (ops, more):
            e.apply(part)
            o.apply(part)
            assert e.geometry() == o.geometry()
            ei, en = e.state()
            oi, on = o.state()
            np.testing.assert_array_equal(en, on)
            np.testing.assert_array_equal(ei, oi)
        assert o.stats()["slide_left_calls"] >= 2


def test_sim_bulk_build(sim, streams):
    """non-parity bulk build: same edge set / values / num_neighbors as the one-by-one build, valid PMA
    invariants, and ordinary updates afterwards keep both"""
    from helpers import check_pma_invariants as _check_invariants_numpy, edge_view as _edge_view
    n = 150
    ops = streams.random_stream(n, 4000, seed=41, p_delete=0.0)
    ops[::9, 2] = 0          # ignored entries (value 0)
    ops[5::13, 0] = n + 7    # ignored entries (src >= n)
    ops[100:200] = ops[0:100]  # duplicates (count in num_neighbors; last value wins)
    ops[100:200, 2] += 3
    e, o = sim(n, mode=1), Oracle(n)
    e.bulk_build(ops)
    for r in ops:
        if r[2] != 0:
            o.add_edge(int(r[0]), int(r[1]), int(r[2]))
    ei, en = e.state()
    _check_invariants_numpy(ei, en)
    assert e.check_invariants() == 0
    for a, b in zip(_edge_view(ei, en), _edge_view(*o.state())):
        np.testing.assert_array_equal(a, b)
    more = streams.random_stream(n, 3000, seed=42, p_delete=0.4)
    e.apply(more)
    o.apply(more)
    ei, en = e.state()
    _check_invariants_numpy(ei, en)
    for a, b in zip(_edge_view(ei, en), _edge_view(*o.state())):
        np.testing.assert_array_equal(a, b)
    with pytest.raises(Exception):
        e.bulk_build(ops)  # only an empty graph can be bulk-built


def test_sim_consumers_bfs_pagerank(sim, streams):
    """device BFS / PageRank over the gapped array == the reference's templates (bfs.h, pagerank.h) on the oracle state;
    PageRank bit for bit (same order of fp32 additions)"""
    from helpers import reference_consumers
    n = 200
    ops = streams.random_stream(n, 3000, seed=21, p_delete=0.15)
    ops[:, 1] %= n
    e, o = sim(n, mode=1), Oracle(n)
    e.apply(ops)
    o.apply(ops)
    vals = (np.arange(n, dtype=np.float32) % 7 + 0.25).astype(np.float32)
    for start in (0, 17, n - 1):
        lv, pr = reference_consumers(o, start, vals)
        np.testing.assert_array_equal(e.bfs(start), lv)
    got = e.pagerank(vals)
    assert got.tobytes() == pr.tobytes()


def test_chain_table_matches_serial_fp64_chain():
    """the piecewise-linear position table (built with the fp64-reciprocal division) == the oracle's serial `x -= step`
    chain (PCSR.cpp:237-247), for windows up to 2^31 slots and densities across the PMA's range"""
    import ctypes
    from oracle_lib import oracle_lib
    build_sim()
    lib = ctypes.CDLL(SIM_SO)
    c_u64 = ctypes.c_uint64
    lib.ppcsr_sim_chain_positions.argtypes = [c_u64, c_u64, c_u64, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int),
                                              ctypes.POINTER(ctypes.c_int)]
    L = oracle_lib()
    rng = np.random.default_rng(5)
    cases = [(0, 1 << 24, 10_000_000), (0, 1 << 24, 4_194_305), (1 << 30, 1 << 30, 3_000_001), (0, 1 << 31, 2_500_000),
             (4096, 4096, 4095), (0, 8, 1), (0, 8, 2), (64, 64, 64), (12288, 8192, 3), (16384, 8192, 3), (8192, 8192, 7000)]
    for _ in range(60):
        lg = int(rng.integers(3, 25))
        ln = 1 << lg
        idx = int(rng.integers(0, 1 << (30 - lg))) * ln
        j = int(rng.integers(1, ln + 1))
        cases.append((idx, ln, j))
    # the windows ONE wave / one workgroup rebalances (closed form without the division): small windows at every kind of
    # start — slot 0, powers of two (the window fills its binade from the bottom), other multiples — and every fill
    for _ in range(2500):
        lg = int(rng.integers(3, 11))
        ln = 1 << lg
        kind = int(rng.integers(0, 4))
        idx = 0 if kind == 0 else ((1 << int(rng.integers(lg, 31))) if kind == 1 else int(rng.integers(1, 1 << (31 - lg))) * ln)
        j = int(rng.integers(1, ln + 1)) if rng.integers(0, 4) else int(rng.choice([1, 2, 3, 4, ln - 1, ln]))
        cases.append((idx, ln, max(j, 1)))
    for idx, ln, j in cases:
        ref = np.zeros(j, np.uint64)
        got = np.zeros(j, np.uint64)
        L.po_redistribute_positions(idx, ln, j, ref.ctypes.data)
        nseg, lin = ctypes.c_int(0), ctypes.c_int(0)
        rc = lib.ppcsr_sim_chain_positions(idx, ln, j, got.ctypes.data, ctypes.byref(nseg), ctypes.byref(lin))
        assert rc == 0, (idx, ln, j, nseg.value)
        np.testing.assert_array_equal(got, ref, err_msg=f"window ({idx},{ln}) j={j}")
        assert lin.value & 1, (idx, ln, j)
        if idx >= ln and idx % ln == 0 and j >= 2:  # aligned windows away from slot 0 stay in one binade: the in-wave closed form applies
            assert lin.value & 2, (idx, ln, j)


def test_sim_bucket_kernels(streams):
    """the HIP owner-bucketing kernels (multi-GPU exchange) under the emulator vs the host routing routine"""
    import ctypes
    build_sim()
    pkg = load_pkg()
    lib = pkg.load_library(SIM_SO)
    for n_global, P, m in [(1000, 2, 5000), (1003, 8, 9000), (64, 64, 3000), (5, 8, 100)]:
        ops = streams.random_stream(n_global, m, seed=3 + P, p_delete=0.3)
        ref, ref_counts = pkg.bucket_ops(n_global, P, ops, lib=lib)
        out = np.empty_like(ops)
        counts = np.zeros(P, np.uint64)
        rc = lib.pppcsr_bucket_ops_device(n_global, P, ops.ctypes.data, len(ops), out.ctypes.data, counts.ctypes.data, None)
        assert rc == 0
        np.testing.assert_array_equal(counts, ref_counts)
        np.testing.assert_array_equal(out, ref)


def test_sim_xchg_steps_and_repartition(streams):
    """one rank of the exchange (pack -> layout -> apply, the rows copied by hand) and pppcsr_repartition under the emulator:
    unequal starts go through the boundary table of the bucketing kernels; the repartition rule is checked by
    tests/helpers.py check_repartitioned, updates afterwards against oracles started from the rebuilt state"""
    import ctypes
    from helpers import check_repartitioned
    from oracle_lib import OraclePPPCSR
    build_sim()
    pkg = load_pkg()
    lib = pkg.load_library(SIM_SO)
    n, P = 300, 4
    pp = pkg.PPPCSR(n, numDomain=1, partitionsPerDomain=P, lib=lib)

    def tune():
        for k in range(P):
            e = pp.partition(k)
            for key, v in dict(mode=1, opt_horizon=64, epoch_ops=1024, region_slots=64, small_batch=0, big_grid=2, big_min=512,
                               big_window=131072, max_horizon=32, min_horizon=4, init_horizon=8, rounds_per_sync=2).items():
                e.set_option(key, v)

    def same(parts, label):
        for k in range(P):
            a, b = pp.partition(k), parts[k]
            assert a.get_n() == b.get_n(), label
            assert digest(*a.state(), a.geometry()) == digest(*b.state(), b.geometry()), f"{label}: partition {k}"

    tune()
    ops = streams.random_stream(n, 1200, seed=5, p_delete=0.2)
    pp.xchg_create(1, 0)
    cnt, d_send = pp.xchg_pack(ops.ctypes.data, len(ops))
    assert int(cnt.sum()) == len(ops)
    with pytest.raises(pkg.PpcsrError):  # a rank knows what it sends to itself
        pp.xchg_layout(cnt[::-1].copy() + np.uint64(1))
    dst = pp.xchg_layout(cnt)
    off = 0
    for q in range(P):
        ctypes.memmove(dst[q], d_send + off * 12, int(cnt[q]) * 12)
        off += int(cnt[q])
    pp.xchg_apply()
    with pytest.raises(pkg.PpcsrError):  # nothing packed
        pp.xchg_apply()
    o = OraclePPPCSR(n, True, 1, P)
    o.apply(ops)
    parts = [o.partition(k) for k in range(P)]
    same(parts, "exchange")
    old = np.array([pp.partition_start(k) for k in range(P)], np.uint64)
    st = pp.balanced_starts()
    assert st[0] == 0 and np.all(np.diff(st.astype(np.int64)) >= 0) and st[-1] <= n
    with pytest.raises(pkg.PpcsrError):
        pp.repartition(np.array([0, 150, 100, 200], np.uint64))

    def build_bulk(size, adds):  # the same bulk path on a fresh engine
        e = pkg.PCSR(size, lib=lib)
        e.bulk_build(adds)
        out = e.state()
        e.close()
        return out

    for new in (np.array([0, 50, 50, 299], np.uint64), st):
        before = [pp.partition(k).state() for k in range(P)]
        (d_moved, n_moved), (d_nn, n_nn) = pp.repartition_export(new)  # (pppcsr_repartition = these three steps; the recreated
        tune()                                                           #  engines get the emulator-sized options before they run)
        pp.bulk_build_device(d_moved, n_moved)
        pp.set_num_neighbors_device(d_nn, n_nn)
        tune()
        assert [pp.partition_start(k) for k in range(P)] == [int(x) for x in new]
        after = [pp.partition(k).state() for k in range(P)]
        check_repartitioned(after, before, old, new, n, build_bulk)
        # updates after the (non-parity) rebuild are exact again: oracles started from the engine's state
        parts = [Oracle.from_state(*after[k]) for k in range(P)]
        ops2 = streams.random_stream(n, 500, seed=6 + int(new[1]), p_delete=0.3)
        pp.apply(ops2)
        own = np.searchsorted(new, ops2[:, 0], side="right") - 1
        assert [pp.get_partiton(int(v)) for v in ops2[:50, 0]] == [int(x) for x in own[:50]]
        for k in range(P):
            sub = ops2[own == k].copy()
            sub[:, 0] -= np.uint32(new[k])
            parts[k].apply(sub)
        same(parts, f"updates after {new}")
        old = new
