"""GPU (MI355X): bit-exact parity of the HIP engine, driven through the C ABI, against
(a) the golden fixtures generated from the real reference and (b) the oracle on seeded streams.
Integer / index work: the bar is byte-for-byte equality of edges[] and nodes[]."""
import numpy as np
import pytest

from helpers import GOLDEN_SINGLE, digest, golden, load_pkg, replay_golden
from oracle_lib import Oracle, OraclePPPCSR

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    p = load_pkg()
    p.load_library()  # must be the in-tree HIP build; raises if missing
    return p


# every parity test runs under both schedulers: strict prefix rounds (mode 0) and speculative rounds with
# validated rollback (mode 1, the default); small horizons/epochs/regions force rollbacks to happen in the tests
MODES = {
    "strict": dict(mode=0),
    "speculative": dict(mode=1, small_batch=0),  # (small_batch=0: even tiny batches go through the speculative rounds)
    "speculative-small": dict(mode=1, opt_horizon=1024, epoch_ops=4096, region_slots=256, small_batch=0),
    "default": dict(),  # engine defaults: speculative rounds, batches of <= 256 updates through the strict rounds
    # round 4: the wave-per-update o_check forced (by default the engine picks it for streams with long footprints)
    "wave-check": dict(mode=1, small_batch=0, check_lanes=0, opt_horizon=2048),
    # the diagnostics build of the round kernels with every measurement aid on: o_plan executes the plan, the search and the
    # round's bookkeeping twice (tools/sq_delta.sh) — it must still plan the same thing
    "diag-repeat": dict(mode=1, small_batch=0, dbg_repeat=7, opt_horizon=4096),
}


@pytest.fixture(params=list(MODES))
def mk(request, pkg):
    def make(n, lock=True):
        e = pkg.PCSR(n, lock_search=lock)
        for k, v in MODES[request.param].items():
            e.set_option(k, v)
        return e
    make.mode = request.param
    return make


def _same(eng, o, label=""):
    assert eng.geometry() == o.geometry(), f"{label}: geometry {eng.geometry()} vs {o.geometry()}"
    ei, en = eng.state()
    oi, on = o.state()
    if not np.array_equal(en, on):
        bad = np.nonzero((en != on).any(1))[0]
        raise AssertionError(f"{label}: nodes[] differ at {bad[:8]}: {en[bad[:4]]} vs {on[bad[:4]]}")
    if not np.array_equal(ei, oi):
        bad = np.nonzero((ei != oi).any(1))[0]
        raise AssertionError(f"{label}: edges[] differ at {len(bad)} slots, first {bad[:8]}")
    assert eng.check_invariants() == 0, label


def _stats_match(eng, o):
    se, so = eng.stats(), o.stats()
    for k in ("redistribute_calls", "redistribute_slots", "double_calls", "half_calls", "not_found", "duplicates"):
        assert se[k] == so[k], (k, se[k], so[k])


@pytest.mark.parametrize("name", GOLDEN_SINGLE)
def test_golden(mk, name):
    eng = replay_golden(lambda n, lock: mk(n, lock), name)
    assert eng.check_invariants() == 0


def test_golden_one_op_at_a_time(pkg):
    """same fixture through the single-op entry points (add_edge / remove_edge)"""
    g = golden("add_node_empty_then_edges")
    eng = pkg.PCSR(0)
    for _ in range(5):
        eng.add_node()
    for s, d, op in g["ops"]:
        if op:
            eng.add_edge(int(s), int(d), int(op))
        else:
            eng.remove_edge(int(s), int(d))
    items, nodes = eng.state()
    assert digest(items, nodes, eng.geometry()) == str(g["digests"][-1])


def test_pppcsr_golden(pkg):
    g = golden("pppcsr_p8_n1000")
    pp = pkg.PPPCSR(1000, numDomain=1, partitionsPerDomain=8)
    assert pp.num_partitions() == 8
    np.testing.assert_array_equal([pp.get_partiton(v) for v in range(0, 1000, 7)], g["part_of_vertex"][::7])
    pp.apply(g["ops"])
    for k in range(8):
        p = pp.partition(k)
        assert p.get_n() == int(g["sizes"][k])
        items, nodes = p.state()
        assert digest(items, nodes, p.geometry()) == str(g["digests"][k]), f"partition {k}"
    assert pp.get_n() == 1000
    # the state the reference's forwarding calls (PPPCSR.cpp:46-52) produced, through the same public calls here
    for v in range(1000):
        np.testing.assert_array_equal(np.asarray(pp.getNode(v), np.uint32), g["fwd_nodes"][v])
    for v in range(0, 1000, 3):
        a, b = g["fwd_adj_ptr"][v], g["fwd_adj_ptr"][v + 1]
        np.testing.assert_array_equal(pp.get_neighbourhood(v), g["fwd_adj"][a:b])


@pytest.mark.parametrize("seed,n", [(0, 30), (1, 300), (2, 3000), (3, 20000), (4, 100000)])
@pytest.mark.parametrize("lock", [True, False])
def test_random_mixed_vs_oracle(mk, streams, seed, n, lock):
    core = streams.random_stream(n, 60000, seed=100 + seed)
    fresh = streams.random_stream(n, 20000, seed=200 + seed)
    ops = np.concatenate([core, streams.mixed_existing_stream(core, fresh, seed=300 + seed)])
    eng, o = mk(n, lock), Oracle(n, lock_search=lock)
    for lo in range(0, len(ops), 25000):
        eng.apply(ops[lo:lo + 25000])
        o.apply(ops[lo:lo + 25000])
        _same(eng, o, f"seed {seed} after {lo + 25000}")
    _stats_match(eng, o)


def test_snapshot_restore_incremental(pkg, streams):
    """snapshot() / restore() and the epoch rollback point are kept in step with the live state through dirty tags (only
    what was written since is copied): restore after batches with rollbacks and doublings, then continue"""
    for seed in range(3):
        rng = np.random.default_rng(seed)
        n = int(rng.choice([40, 300, 5000]))
        core = streams.random_stream(n, 4000, seed=seed + 100, p_delete=0.1)
        m = 6000
        src = np.where(rng.random(m) < 0.5, rng.integers(0, 4, m), rng.integers(0, n, m)).astype(np.uint32)
        upd = np.stack([src, rng.integers(0, 100000, m).astype(np.uint32), (rng.random(m) >= 0.1).astype(np.uint32)], 1).astype(np.uint32)
        upd2 = streams.random_stream(n, 1500, seed=seed + 7, p_delete=0.5)
        e, o = pkg.PCSR(n), Oracle(n)
        for k, v in dict(small_batch=0, epoch_ops=1024, opt_horizon=1024, region_slots=256).items():
            e.set_option(k, v)
        e.apply(core)
        o.apply(core)
        e.snapshot()
        g0 = e.geometry()
        for rep in range(3):
            e.restore()
            assert e.geometry() == g0
            _same(e, o, f"seed {seed} rep {rep}: after restore")
            which = upd if rep != 1 else upd2
            e.apply(which)
            o2 = o.clone()
            o2.apply(which)
            _same(e, o2, f"seed {seed} rep {rep}: after the batch")
            o2.close()


def test_horizon_options_changed_between_batches(pkg, streams):
    """the plan-record buffer is shared by both schedulers; shrinking max_horizon after a speculative batch
    (or growing opt_horizon) must never leave it smaller than the widest grid either may launch"""
    n = 5000
    ops = streams.random_stream(n, 90000, seed=77, p_delete=0.15)
    eng, o = pkg.PCSR(n), Oracle(n)
    eng.set_option("small_batch", 0)
    eng.apply(ops[:30000])               # speculative, rounds up to opt_horizon wide (3 x the resident waves)
    eng.set_option("max_horizon", 1024)  # used to reallocate the shared buffer with 1024 records
    eng.apply(ops[30000:60000])          # speculative again: o_plan writes plans[wid] for wid < opt_horizon
    eng.set_option("mode", 0)
    eng.apply(ops[60000:75000])          # strict rounds on the same buffer
    eng.set_option("mode", 1)
    eng.set_option("opt_horizon", 8192)  # wider than anything allocated so far
    eng.apply(ops[75000:])
    o.apply(ops)
    _same(eng, o, "after option changes")


@pytest.mark.parametrize("opts", [
    {"resident_waves": 0, "opt_horizon": 6144},                       # round 1's fixed width, no quantisation
    {"resident_waves": 6144, "opt_horizon": 24576},                   # up to four passes of resident waves
    {"resident_waves": 1024, "opt_horizon": 3072, "start_horizon": 512},
    {"soft_barrier": 1 << 30, "epoch_short": 1024, "epoch_grow_after": 1},
    {"soft_barrier": 2048, "epoch_short": 4096, "epoch_grow_after": 8, "rb_inplace_min": 4096, "big_window": 8192},
])
def test_scheduler_knobs_do_not_change_the_result(pkg, streams, opts):
    """round width in multiples of the resident waves, soft barrier, epoch lengths, in-place windows: every setting is a
    schedule of the same sequential semantics — RMAT hubs + a hot-vertex tail, slot by slot against the oracle"""
    n = 1 << 15
    s, d = streams.rmat_edges(15, 400000, seed=8)
    zs = streams.zipf_sources(n, 60000, seed=9, alpha=1.2)
    ops = np.concatenate([streams.adds(s, d), streams.adds(zs, streams.uniform_ints(10, 60000, n))])
    ops = np.concatenate([ops, streams.mixed_existing_stream(ops[:400000], streams.adds(*streams.rmat_edges(15, 30000, seed=11)), seed=12)])
    eng, o = pkg.PCSR(n), Oracle(n)
    for k, v in opts.items():
        eng.set_option(k, v)
    eng.apply(ops)
    o.apply(ops)
    _same(eng, o, str(opts))
    st = eng.stats()
    assert st["committed"] >= len(ops) - st["exclusive_ops"] - 1


def test_hubs_and_last_vertex(mk, streams):
    m = 20000
    for src_mode in ("last", "first", "tail"):
        hub = np.stack([np.full(m, 9), streams.uniform_ints(3, m, 1 << 30), np.ones(m)], 1).astype(np.uint32)
        if src_mode == "first":
            hub[:, 0] = 0
        elif src_mode == "tail":
            hub[:, 0] = streams.uniform_ints(4, m, 3) + 7
        dele = hub.copy()
        dele[:, 2] = 0
        ops = np.concatenate([hub, dele[::-1]])
        eng, o = mk(10), Oracle(10)
        eng.apply(ops)
        o.apply(ops)
        _same(eng, o, src_mode)
        _stats_match(eng, o)


@pytest.mark.parametrize("seed", [100, 123, 128, 137, 151, 2395, 2554, 3663])
def test_api_sequences_found_by_the_fuzzer(pkg, streams, seed):
    """random API sequences (add_edge / remove_edge / add_node / queries / batches) that tools/fuzz_api.py caught diverging:
    a one-vertex graph shrunk to a single leaf (the reference reads out of bounds there: the run stops before it), and
    add_node after edges + a doubling, which drops the new sentinel into another vertex's range (PCSR.cpp:533-540,
    681-703) so that later searches run over inverted / unsorted ranges — the engine then follows the literal walk and
    runs one update per round (footprints computed from disjoint ranges no longer hold)"""
    import importlib.util
    import os
    from helpers import ROOT
    spec = importlib.util.spec_from_file_location("fuzz_api", os.path.join(ROOT, "tools", "fuzz_api.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    ok, desc = fz.run_case_tolerant(pkg, streams, seed)
    assert ok, desc


def test_leaves_the_sequential_regime_when_ranges_are_sane(pkg, streams):
    """the sequential regime (one update per round) entered when add_node hits the reference's
    re-search-after-doubling path is left again as soon as the device-side range check finds sorted, disjoint, consistent
    vertex ranges (here the regime is forced by hand on a regular structure); a structure the reference has really
    corrupted stays sequential (fuzzer seed 2554 above)"""
    n = 20000
    ops = streams.random_stream(n, 120000, seed=23, p_delete=0.2)
    eng, o = pkg.PCSR(n), Oracle(n)
    eng.apply(ops[:60000])
    eng.set_option("search_narrow", 0)
    assert eng.stats()["narrow"] == 0
    r0 = eng.stats()["rounds"]
    eng.apply(ops[60000:])
    o.apply(ops)
    _same(eng, o, "after leaving the sequential regime")
    assert eng.stats()["narrow"] == 1
    assert eng.stats()["rounds"] - r0 < 2000


def test_config1_plumbing_case(mk, streams):
    """BASELINE config #1: scale-14 RMAT, 200 000-edge core (seed 1) + 100 000 inserts (seed 2) — the
    reference's own CPU-runnable case, as a slot-by-slot parity test"""
    n = 1 << 14
    s, d = streams.rmat_edges(14, 200_000, seed=1)
    s2, d2 = streams.rmat_edges(14, 100_000, seed=2)
    eng, o = mk(n), Oracle(n)
    for part in (streams.adds(s, d), streams.adds(s2, d2)):
        eng.apply(part)
        o.apply(part)
    _same(eng, o, "config #1")
    _stats_match(eng, o)


def test_slide_off_the_end_of_the_array(mk, streams):
    """slide_right runs off the end of the array (PCSR.cpp:347-351) and the reference recovers through slide_left
    (PCSR.cpp:360-390, 541-544) — a path tools/fuzz_parity.py found on ascending inserts into the last vertices"""
    from helpers import slide_off_end_stream
    ops = slide_off_end_stream()
    more = streams.random_stream(4096, 20000, seed=78, p_delete=0.2)
    more[:, 0] = 4096 - 1 - (more[:, 0] % 40)
    for lock in (True, False):
        eng, o = mk(4096, lock), Oracle(4096, lock_search=lock)
        for part in (ops, more):
            eng.apply(part)
            o.apply(part)
            _same(eng, o, f"lock={lock}")
        _stats_match(eng, o)
        assert o.stats()["slide_left_calls"] >= 2


def test_sparse_hub_searches(mk, streams):
    """the 64-ary bracket narrowing on a hub whose range is mostly gaps: build a 30 K-edge hub, delete 97 % of it (the
    array never shrinks below its doubled size while the other vertices keep it dense enough), then look keys up through
    every exit of the search — exact hits, misses between far-apart survivors, below the first and above the last edge,
    duplicates — interleaved with inserts that land in the gaps"""
    m = 30000
    n = 64
    d = streams.uniform_ints(11, m, 1 << 28) * 4 + 2
    hub = np.stack([np.full(m, 5), d, np.ones(m)], 1).astype(np.uint32)
    filler_s = streams.uniform_ints(12, 60000, n)
    filler = np.stack([filler_s, streams.uniform_ints(13, 60000, 1 << 30), np.ones(60000)], 1).astype(np.uint32)
    dele = hub[streams.uniform_ints(14, m, 100) < 97].copy()
    dele[:, 2] = 0
    probe_d = np.concatenate([d[:4000] + 1, d[:4000], np.arange(1, 4001, dtype=np.uint64), d[:2000] - 1]).astype(np.uint32)
    probes = np.stack([np.full(len(probe_d), 5), probe_d, np.ones(len(probe_d))], 1).astype(np.uint32)
    probes[1::3, 2] = 0  # every third probe is a delete (hit or miss)
    ops = np.concatenate([hub, filler, dele, probes])
    for lock in (True, False):
        eng, o = mk(n, lock), Oracle(n, lock_search=lock)
        eng.apply(ops)
        o.apply(ops)
        _same(eng, o, f"lock={lock}")
        _stats_match(eng, o)
        for dd in (int(d[7]), int(d[7]) + 1, 1, (1 << 30) + 7):
            assert eng.edge_exists(5, dd) == o.edge_exists(5, dd)


def test_ascending_and_descending_runs(mk):
    """long slides (descending dests) and end-of-array inserts (ascending dests into the last vertex)"""
    m = 12000
    for order in ("asc", "desc"):
        d = np.arange(m) + 5 if order == "asc" else np.arange(m, 0, -1) + 5
        ops = np.stack([np.full(m, 9), d, np.ones(m)], 1).astype(np.uint32)
        eng, o = mk(10), Oracle(10)
        eng.apply(ops)
        o.apply(ops)
        _same(eng, o, order)


def test_rmat_core_plus_updates(mk, streams):
    s, d = streams.rmat_edges(16, 600000, seed=1)
    core = streams.adds(s, d)
    s2, d2 = streams.rmat_edges(16, 100000, seed=2)
    upd = streams.mixed_existing_stream(core, streams.adds(s2, d2), seed=3)
    n = 1 << 16
    eng, o = mk(n), Oracle(n)
    eng.apply(core)
    o.apply(core)
    _same(eng, o, "core")
    eng.apply(upd)
    o.apply(upd)
    _same(eng, o, "updates")
    _stats_match(eng, o)
    # neighbour lists and the bulk scan (get_neighbourhood for all vertices)
    rows, dests = eng.scan_all()
    for v in list(range(0, 64)) + list(range(64, n, 997)) + [n - 1]:
        ref = o.get_neighbourhood(v)
        np.testing.assert_array_equal(eng.get_neighbourhood(v), ref)
        np.testing.assert_array_equal(dests[int(rows[v]):int(rows[v + 1])], ref)
    items, _ = o.state()
    live = (items[:, 2] != 0) & (items[:, 1] != 0xFFFFFFFF)
    live[-1] = False
    assert rows[-1] == live.sum()


def test_api_semantics(pkg):
    """DataStructureTest.cpp:12-49 restated against the HIP engine (through PPPCSR with one partition)"""
    p = pkg.PPPCSR(10, numDomain=1, partitionsPerDomain=1)
    assert p.get_n() == 10
    p.add_edge(11, 1, 1)  # no such source: silently ignored
    p.add_edge(0, 1, 1)
    assert p.edge_exists(0, 1)
    assert len(p.get_neighbourhood(0)) == 1 and len(p.get_neighbourhood(2)) == 0
    p.add_node()
    assert p.get_n() == 11
    p.remove_edge(0, 1)
    assert not p.edge_exists(0, 1)
    p.remove_edge(0, 1)  # miss
    assert p.getNode(0)[2] == 0xFFFFFFFF  # num_neighbors underflow quirk (PCSR.cpp:747)
    e = pkg.PPPCSR(0, numDomain=1, partitionsPerDomain=1)
    assert e.get_n() == 0
    e.add_node()
    assert e.get_n() == 1 and len(e.get_neighbourhood(0)) == 0
    assert len(p.get_neighbourhood(500)) == 0  # out of range source -> empty (PCSR.cpp:903)


def test_seq_stress_edge_exists(pkg):
    """add_remove_edge_1E4_seq (DataStructureTest.cpp:51-79), with edge_exists sampled"""
    eng = pkg.PCSR(10)
    m = 10000
    ins = np.stack([np.zeros(m), np.arange(1, m + 1), np.arange(1, m + 1)], 1).astype(np.uint32)
    eng.apply(ins)
    assert eng.getNode(0)[2] == m
    for i in (1, 2, 777, 5000, m):
        assert eng.edge_exists(0, i)
    assert not eng.edge_exists(0, m + 1)
    dele = ins.copy()
    dele[:, 2] = 0
    eng.apply(dele)
    assert len(eng.get_neighbourhood(0)) == 0 and eng.get_n() == 10
    assert not eng.edge_exists(0, 5)


@pytest.mark.parametrize("variant,tile,batch,defer", [(2, 0, 1, 1 << 22), (2, 32, 1, 1 << 22), (2, 256, 0, 1 << 22), (1, 0, 1, 1 << 22), (0, 0, 1, 1 << 22),
                                                      (2, 0, 1, 64), (2, 32, 1, 4096), (2, 256, 0, 64)])
def test_big_window_rebalance(pkg, streams, variant, tile, batch, defer):
    """multi-workgroup rebalance kernels on 2^21..2^15-slot windows vs the oracle's redistribute(): the default pipeline
    (tile sums + in-tile scan + four chunks in flight) at several tile sizes, and the two older scatter variants that stay
    selectable for A/B measurements; array doublings go through the same pipeline (out of place).  defer = 64 / 4096: windows
    from slot 0 build their position table INSIDE the scatter launch (published segment by segment, tiles taken top-down) —
    by default only windows of >= 2^22 slots do"""
    n = 1 << 16
    s, d = streams.rmat_edges(16, 500000, seed=4)
    ops = streams.adds(s, d)
    e, o = pkg.PCSR(n), Oracle(n)
    e.set_option("scatter_variant", variant)
    e.set_option("rb_tile", tile)
    e.set_option("rb_prefetch", batch)
    e.set_option("rb_defer_table", defer)
    e.apply(ops)
    o.apply(ops)
    _same(e, o, "load (with doublings)")
    N = e.geometry()[0]
    for w in (N, N // 2, N // 8, N // 64):
        e.bench_rebalance(w, 1)
        o.debug_redistribute(0, w)
        _same(e, o, f"window {w}")
    s2, d2 = streams.rmat_edges(16, 100000, seed=5)
    more = streams.adds(s2, d2)
    e.apply(more)
    o.apply(more)
    _same(e, o, "updates after the rebalances")


@pytest.mark.parametrize("shape,lists", [("uniform", 0), ("dense_left", 0), ("dense_right", 0), ("dense_middle", 0), ("sparse_middle", 0),
                                         ("dense_middle", 1), ("dense_right", 2), ("sparse_middle", 4)])
def test_partial_window_rebalance_in_place(pkg, streams, shape, lists):
    """partial windows rebalanced inside the array (k_rb_order + k_rb_inplace; hundreds to thousands of 2048-slot tiles,
    elements moving left, right, outward, inward — several tiles far) against the reference's redistribute() run by the oracle"""
    n = 1 << 16
    s, d = streams.rmat_edges(16, 500000, seed=4)
    base = streams.adds(s, d)
    lo, hi = {"uniform": (0, 0), "dense_left": (0, n // 16), "dense_right": (n // 4, n // 2), "dense_middle": (n // 6, n // 5),
              "sparse_middle": (0, 0)}[shape]
    extra = []
    if hi > lo:
        src = streams.uniform_ints(31, 400000, hi - lo, lo)
        extra = [streams.adds(src, streams.uniform_ints(32, 400000, n))]
    ops = np.concatenate([base] + extra)
    e, o = pkg.PCSR(n), Oracle(n)
    e.set_option("rb_inplace_min", 2048)
    if lists:  # ticket lists: by default as many as the XCD ids the engine saw at creation (8 on an MI355X), 1 = one counter
        e.set_option("rb_inplace_lists", lists)
    e.set_option("big_window", 4096)  # windows above it go to the host-driven path (in place) during the loads as well
    e.apply(ops)
    o.apply(ops)
    if shape == "sparse_middle":
        dele = base[(base[:, 0] >= n // 64) & (base[:, 0] < n // 8)].copy()
        dele[:, 2] = 0
        e.apply(dele)
        o.apply(dele)
    _same(e, o, "before")
    N = e.geometry()[0]
    for w in (N // 2, N // 4, N // 32, 4096):
        e.bench_rebalance(w, 1)
        o.debug_redistribute(0, w)
        _same(e, o, f"{shape}: window {w}")
    s2, d2 = streams.rmat_edges(16, 100000, seed=5)
    more = streams.adds(s2, d2)
    e.apply(more)
    o.apply(more)
    _same(e, o, "updates after the in-place rebalances")


def test_bulk_build_fast_path(pkg, streams):
    """non-parity bulk build of a 1 M-edge RMAT graph: valid PMA invariants, same edge set / values /
    num_neighbors as the one-by-one build (oracle), consumers agree, and ordinary updates afterwards keep all of it"""
    from helpers import check_pma_invariants, edge_view, reference_consumers
    scale, m = 16, 1_000_000
    n = 1 << scale
    s, d = streams.rmat_edges(scale, m, seed=51)
    ops = streams.adds(s, d)
    ops[::11, 2] = 0            # ignored: value 0
    ops[3::17, 0] = n + 5       # ignored: src >= n
    ops[1000:2000] = ops[0:1000]  # duplicates: counted, last value wins
    ops[1000:2000, 2] = 9
    eng, o = pkg.PCSR(n), Oracle(n)
    ms = eng.bulk_build(ops, with_ms=True)
    o.apply(ops[ops[:, 2] != 0])
    ei, en = eng.state()
    check_pma_invariants(ei, en)
    assert eng.check_invariants() == 0
    for a, b in zip(edge_view(ei, en), edge_view(*o.state())):
        np.testing.assert_array_equal(a, b)
    lv, _ = reference_consumers(o, 0, np.ones(n, np.float32))
    np.testing.assert_array_equal(eng.bfs(0), lv)
    s2, d2 = streams.rmat_edges(scale, 200_000, seed=52)
    more = streams.mixed_existing_stream(ops[(ops[:, 2] != 0) & (ops[:, 0] < n)][:300_000], streams.adds(s2, d2), seed=53)
    eng.apply(more)
    o.apply(more)
    ei, en = eng.state()
    check_pma_invariants(ei, en)
    for a, b in zip(edge_view(ei, en), edge_view(*o.state())):
        np.testing.assert_array_equal(a, b)
    with pytest.raises(pkg.PpcsrError):
        eng.bulk_build(ops)


def test_consumers_bfs_pagerank(pkg, streams):
    """GPU BFS / PageRank push over the gapped array == the reference's bfs.h / pagerank.h templates
    evaluated on the oracle's state; PageRank compared bit for bit (fp32 additions in the reference's order), incl. a
    vertex whose num_neighbors is 0 or wrapped (division quirks) and dests beyond n (skipped)"""
    from helpers import reference_consumers
    scale, m = 14, 300_000
    n = 1 << scale
    s, d = streams.rmat_edges(scale, m, seed=31)
    ops = streams.adds(s, d)
    dele = ops[::7].copy()
    dele[:, 2] = 0
    extra = np.array([[5, n + 3, 1], [9, 1, 0], [9, 1, 0]], np.uint32)  # dest beyond n; deletes of a missing edge (count wraps)
    ops = np.concatenate([ops, dele, extra])
    eng, o = pkg.PCSR(n), Oracle(n)
    eng.apply(ops)
    o.apply(ops)
    vals = (streams.uniform_ints(32, n, 1000).astype(np.float32) / np.float32(7.0)).astype(np.float32)
    for start in (0, 1, int(s[12345])):
        lv, pr = reference_consumers(o, start, vals)
        got, ms = eng.bfs(start, with_ms=True)
        np.testing.assert_array_equal(got, lv)
    got, ms = eng.pagerank(vals, with_ms=True)
    assert got.tobytes() == pr.tobytes(), f"pagerank differs at {np.nonzero(got.view(np.uint32) != pr.view(np.uint32))[0][:8]}"
    with pytest.raises(pkg.PpcsrError):
        eng.bfs(n)


def test_consumers_golden_from_reference_templates(pkg):
    """ppcsr_bfs / ppcsr_pagerank against golden vectors produced by the reference's own bfs.h / pagerank.h templates on
    the reference PCSR (tests/golden/consumers_rmat12.npz): identical levels, PageRank identical bit for bit"""
    g = golden("consumers_rmat12")
    eng = pkg.PCSR(int(g["n"]))
    eng.apply(g["ops"])
    for i, start in enumerate(g["starts"]):
        np.testing.assert_array_equal(eng.bfs(int(start)), g["levels"][i])
    assert eng.pagerank(g["node_values"]).tobytes() == g["pagerank"].tobytes()
    assert eng.pagerank(np.ones(int(g["n"]), np.float32)).tobytes() == g["pagerank_ones"].tobytes()


def test_bucket_ops_device_matches_host_routing(pkg, streams):
    """HIP counting-sort bucketing (the multi-GPU exchange's device side) == torch stable sort == host routine"""
    import importlib.util
    import os
    import torch
    from helpers import ROOT
    spec = importlib.util.spec_from_file_location("ppcsr_exchange", os.path.join(ROOT, "parallel-packed-csr_amd", "exchange.py"))
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    for n_global, P, m in [(1 << 20, 8, 1_000_000), (1003, 8, 9000), (1 << 23, 2, 300_000), (77, 64, 5000)]:
        ops = streams.random_stream(n_global, m, seed=P + 1, p_delete=0.3)
        t = torch.from_numpy(ops.view(np.int32)).cuda()
        out, counts = ex.bucket_ops_device(t, n_global, P)
        ref, ref_counts = ex.bucket_ops(t, n_global, P)
        torch.cuda.synchronize()
        assert torch.equal(counts, ref_counts)
        assert torch.equal(out, ref)
        hb, hc = pkg.bucket_ops(n_global, P, ops)
        np.testing.assert_array_equal(out.cpu().numpy().view(np.uint32), hb)


def test_native_rccl_exchange_single_rank(pkg, streams):
    """pppcsr_exchange_apply with a one-rank RCCL communicator (all a one-GPU box can run: RCCL refuses two ranks on one
    device): device bucketing -> counts to self -> rows to self (grouped ncclSend / ncclRecv, landing where the partition
    streams want them) -> concurrent per-partition apply, against the oracle's PPPCSR partition by partition.  Past one rank
    the same pack / layout / apply code runs in tests/test_exchange_gloo.py (two emulator processes, gloo as the carrier)."""
    import torch
    n, P = 50000, 8
    o = OraclePPPCSR(n, True, 1, P)
    pp = pkg.PPPCSR(n, numDomain=1, partitionsPerDomain=P, local=(0, P, 0))
    pp.comm_create(pkg.PPPCSR.comm_unique_id(), 1, 0, 0)
    for k in range(4):
        ops = streams.random_stream(n, [200000, 1, 0, 70000][k], seed=60 + k, p_delete=0.25)
        t = torch.from_numpy(ops.view(np.int32).copy()).cuda()
        torch.cuda.synchronize()
        pp.exchange_apply(t.data_ptr() if len(ops) else 0, len(ops))
        o.apply(ops)
    for k in range(P):
        a, b = pp.partition(k), o.partition(k)
        assert a.geometry() == b.geometry()
        ei, en = a.state()
        oi, on = b.state()
        assert np.array_equal(ei, oi) and np.array_equal(en, on), f"partition {k}"
    pp.close()
    # a handle that does not hold exactly the communicator's range is refused (here: 4 of 8 partitions, one rank)
    half = pkg.PPPCSR(n, numDomain=2, partitionsPerDomain=4, local=(0, 4, 0))
    half.comm_create(pkg.PPPCSR.comm_unique_id(), 1, 0, 0)
    t = torch.zeros((4, 3), dtype=torch.int32).cuda()
    with pytest.raises(pkg.PpcsrError):
        half.exchange_apply(t.data_ptr(), 4)
    half.close()


def test_xchg_steps_single_rank(pkg, streams):
    """pack -> layout -> apply with the rows moved by the caller (device-to-device copies through torch): the carrier-free form
    of the exchange, on the GPU build"""
    import torch
    n, P = 30000, 4
    o = OraclePPPCSR(n, True, 1, P)
    pp = pkg.PPPCSR(n, numDomain=1, partitionsPerDomain=P)
    pp.xchg_create(1, 0)
    for k in range(2):
        ops = streams.random_stream(n, 120000, seed=80 + k, p_delete=0.3)
        t = torch.from_numpy(ops.view(np.int32).copy()).cuda()
        torch.cuda.synchronize()
        cnt, d_send = pp.xchg_pack(t.data_ptr(), len(ops))
        ref, ref_cnt = pkg.bucket_ops(n, P, ops)
        np.testing.assert_array_equal(cnt, ref_cnt)
        dst = pp.xchg_layout(cnt)
        off = 0
        for q in range(P):
            c = int(cnt[q])
            rows = torch.from_numpy(ref[off:off + c].view(np.int32).copy()).cuda()  # what the bucketed block holds for q
            torch.cuda.synchronize()
            pkg.load_library().ppcsr_device_count()
            _d2d(dst[q], rows.data_ptr(), c * 12)
            off += c
        pp.xchg_apply()
        o.apply(ops)
    for k in range(P):
        a, b = pp.partition(k), o.partition(k)
        assert digest(*a.state(), a.geometry()) == digest(*b.state(), b.geometry()), f"partition {k}"
    pp.close()


def _d2d(dst, src, nbytes):
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    assert nbytes == 0 or hip.hipMemcpy(dst, src, nbytes, 3) == 0  # hipMemcpyDeviceToDevice
    assert hip.hipDeviceSynchronize() == 0


@pytest.mark.parametrize("P", [4, 8])
def test_repartition(pkg, streams, P):
    """pppcsr_repartition: partitions whose range stays are untouched bit for bit; partitions that change equal
    what the same bulk path builds from the edges of their new range, with every vertex's num_neighbors carried over
    (tests/helpers.py check_repartitioned: edge set, vertex ranges, PMA invariants); neighbourhoods survive; updates applied
    afterwards are bit-exact against oracles started from the rebuilt states.  Skewed graph (RMAT labels) to balanced starts
    and back; duplicates and deletes of missing edges in the load make num_neighbors differ from the degree."""
    from helpers import check_repartitioned
    n = 1 << 14
    s_, d_ = streams.rmat_edges(14, 150000, seed=7)
    core = np.concatenate([streams.adds(s_, d_), streams.random_stream(n, 5000, seed=8, p_delete=1.0)])  # (RMAT repeats edges; + misses)
    pp = pkg.PPPCSR(n, numDomain=1, partitionsPerDomain=P)
    pp.apply(core)
    old = np.array([pp.partition_start(k) for k in range(P)], np.uint64)
    st0 = [pp.partition(k).state() for k in range(P)]
    deg = np.concatenate([np.diff(np.append(np.nonzero(it[:, 1] == 0xFFFFFFFF)[0], len(it))) for it, _ in st0])
    assert (np.concatenate([nd[:, 2] for _, nd in st0]).astype(np.int64) != (deg - 1)).any()  # counters are not degrees here
    sizes_before = [int((it[:, 2] != 0).sum()) for it, _ in st0]
    adj_before = {v: pp.get_neighbourhood(v).copy() for v in range(0, n, 97)}
    st = pp.balanced_starts()
    assert st[0] == 0 and np.all(np.diff(st.astype(np.int64)) >= 0)

    def build_bulk(size, adds):
        e = pkg.PCSR(size)
        e.bulk_build(adds)
        out = e.state()
        e.close()
        return out

    for new in (st, old):  # to the balanced layout and back to the uniform one
        before = [pp.partition(k).state() for k in range(P)]
        pp.repartition(new)
        assert pp.get_n() == n and [pp.partition_start(k) for k in range(P)] == [int(x) for x in new]
        after = [pp.partition(k).state() for k in range(P)]
        check_repartitioned(after, before, old, new, n, build_bulk)
        for k in range(P):
            assert pp.partition(k).check_invariants() == 0
        for v, adj in adj_before.items():
            np.testing.assert_array_equal(pp.get_neighbourhood(v), adj)
        parts = [Oracle.from_state(*after[k]) for k in range(P)]
        upd = streams.mixed_existing_stream(core[:50000], streams.random_stream(n, 30000, seed=int(new[1]) % 1000), seed=11)
        pp.apply(upd)
        own = np.searchsorted(new, upd[:, 0], side="right") - 1
        for k in range(P):
            sub = upd[own == k].copy()
            sub[:, 0] -= np.uint32(new[k])
            parts[k].apply(sub)
            a = pp.partition(k)
            assert digest(*a.state(), a.geometry()) == digest(*parts[k].state(), parts[k].geometry()), f"updates after {new}: {k}"
        adj_before = {v: pp.get_neighbourhood(v).copy() for v in range(0, n, 97)}
        old = new
    assert max(sizes_before) > 2 * min(sizes_before)  # (the raw RMAT labels were skewed to begin with)
    pp.close()


def test_scan_all_repeated_across_resizes(pkg, streams):
    """bulk scan called repeatedly while the array doubles and halves (its scratch state is per array size); long runs
    of isolated vertices (adjacent sentinels spanning several waves) included"""
    n = 5000
    eng, o = pkg.PCSR(n), Oracle(n)

    def check():
        rows, dests = eng.scan_all()
        items, nodes = o.state()
        live = (items[:, 2] != 0) & (items[:, 1] != 0xFFFFFFFF)
        live[-1] = False
        np.testing.assert_array_equal(dests, items[live, 1].astype(np.int32))
        deg = np.add.reduceat(live.astype(np.int64), nodes[:, 0]) if len(dests) else np.zeros(n, np.int64)
        np.testing.assert_array_equal(np.diff(rows.astype(np.int64)), deg)

    check()
    s, d = streams.rmat_edges(12, 60_000, seed=9)
    adds = streams.adds(s + 700, d)  # vertices 0..699 and 4796.. stay isolated
    for part in (adds[:3000], adds[3000:30000], adds[30000:]):
        eng.apply(part)
        o.apply(part)
        check()
        check()
    dels = adds.copy()
    dels[:, 2] = 0
    eng.apply(dels[:55000])
    o.apply(dels[:55000])
    check()
    assert eng.geometry() == o.geometry()


def test_large_graph_properties(pkg, streams):
    """BASELINE-sized shape (scale-19 RMAT, 4 M-edge core + 1 M mixed updates) checked through size-independent
    properties of the reference's data structure instead of a slot-by-slot oracle comparison:
    canonical nulls, sentinel <-> nodes[] consistency, sorted neighbourhoods, exact edge set, call-count semantics."""
    scale, m = 19, 4_000_000
    n = 1 << scale
    s, d = streams.rmat_edges(scale, m, seed=1)
    core = streams.adds(s, d)
    s2, d2 = streams.rmat_edges(scale, 500_000, seed=2)
    upd = streams.mixed_existing_stream(core, streams.adds(s2, d2), seed=3)
    eng = pkg.PCSR(n)
    eng.apply(core)
    eng.apply(upd)
    assert eng.check_invariants() == 0
    items, nodes = eng.state()
    N = len(items)
    live = items[:, 2] != 0
    # (1) every null slot is exactly {0xFFFFFFFF, 0, 0}
    nul = items[~live]
    assert (nul[:, 0] == 0xFFFFFFFF).all() and (nul[:, 1] == 0).all()
    # (2) sentinels: slot nodes[v].beginning holds {v, MAX, v} (v = 0: value MAX); end = next beginning; last end = N-1
    sent = live & (items[:, 1] == 0xFFFFFFFF)
    pos = np.nonzero(sent)[0]
    assert len(pos) == n
    np.testing.assert_array_equal(items[pos, 0], np.arange(n, dtype=np.uint32))
    np.testing.assert_array_equal(nodes[:, 0], pos.astype(np.uint32))
    np.testing.assert_array_equal(nodes[:-1, 1], nodes[1:, 0])
    assert nodes[-1, 1] == N - 1
    # (3) edges between sentinel v and v+1 have src == v and strictly increasing dest
    e = items[live & ~sent]
    assert (np.diff(e[:, 0].astype(np.int64)) >= 0).all()
    same = e[1:, 0] == e[:-1, 0]
    assert (e[1:, 1][same] > e[:-1, 1][same]).all()
    owner = np.searchsorted(pos, np.nonzero(live & ~sent)[0], side="right") - 1
    np.testing.assert_array_equal(owner.astype(np.uint32), e[:, 0])
    # (4) the edge SET equals the stream's net effect
    key = lambda a: (a[:, 0].astype(np.uint64) << np.uint64(32)) | a[:, 1].astype(np.uint64)
    expect = set(key(core).tolist())
    for row_k, op in zip(key(upd).tolist(), upd[:, 2].tolist()):
        if op:
            expect.add(row_k)
        else:
            expect.discard(row_k)
    got = key(e)
    assert len(got) == len(expect) and set(got.tolist()) == expect
    # (5) num_neighbors counts add calls minus delete calls per source (PCSR.cpp:1392, :747), mod 2^32
    allops = np.concatenate([core, upd])
    cnt = np.bincount(allops[allops[:, 2] != 0, 0], minlength=n).astype(np.int64) - \
        np.bincount(allops[allops[:, 2] == 0, 0], minlength=n).astype(np.int64)
    np.testing.assert_array_equal(nodes[:, 2], (cnt % (1 << 32)).astype(np.uint32))
    # (6) the bulk scan agrees with the array
    rows, dests = eng.scan_all()
    assert rows[-1] == len(e) - (1 if (live[-1] and not sent[-1]) else 0)
