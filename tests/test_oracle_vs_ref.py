"""CPU, build container only: the oracle restatement against the UNMODIFIED reference compiled
into oracle/_ref (skipped where that library is absent)."""
import numpy as np
import pytest

from oracle_lib import Oracle, RefPCSR, have_ref

pytestmark = [pytest.mark.ref, pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built (no /root/reference)")]


def _same(o, r, label):
    assert o.geometry() == r.geometry(), label
    oi, on = o.state()
    ri, rn = r.state()
    np.testing.assert_array_equal(oi, ri, err_msg=label)
    np.testing.assert_array_equal(on, rn, err_msg=label)


def _run(n, ops, lock=True, label=""):
    o, r = Oracle(n, lock_search=lock), RefPCSR(n, lock_search=lock)
    o.apply(ops)
    r.apply(ops)
    _same(o, r, label)
    for v in range(0, n, max(1, n // 50)):
        np.testing.assert_array_equal(o.get_neighbourhood(v), r.get_neighbourhood(v))
    o.close()
    r.close()


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("lock", [True, False])
def test_random_mixed(streams, seed, lock):
    n = [30, 200, 1000, 3000, 5000, 20000][seed]
    core = streams.random_stream(n, 40000, seed=100 + seed)
    fresh = streams.random_stream(n, 15000, seed=200 + seed)
    ops = np.concatenate([core, streams.mixed_existing_stream(core, fresh, seed=300 + seed)])
    _run(n, ops, lock, f"seed {seed}")


def test_delete_everything_then_reinsert(streams):
    n = 300
    a = streams.random_stream(n, 30000, seed=5)
    d = a.copy()
    d[:, 2] = 0
    _run(n, np.concatenate([a, d, a[::-1]]))


def test_hub_and_last_vertex(streams):
    m = 20000
    hub = np.stack([np.full(m, 9), streams.uniform_ints(3, m, 1 << 30), np.ones(m)], 1).astype(np.uint32)  # last vertex
    _run(10, hub, label="last-vertex hub")
    hub[:, 0] = 0
    _run(10, hub, label="first-vertex hub")
    hub[:, 0] = streams.uniform_ints(4, m, 3) + 7
    _run(10, hub, label="tail hubs")


def test_rmat_skew(streams):
    s, d = streams.rmat_edges(13, 150000, seed=1)
    core = streams.adds(s, d)
    s2, d2 = streams.rmat_edges(13, 40000, seed=2)
    ops = np.concatenate([core, streams.mixed_existing_stream(core, streams.adds(s2, d2), seed=3)])
    _run(1 << 13, ops)


def test_edge_values_and_add_node(streams):
    o, r = Oracle(0), RefPCSR(0)
    for _ in range(20):
        o.add_node()
        r.add_node()
    ops = streams.random_stream(20, 5000, seed=9, p_delete=0.3)
    ops[:, 2] *= streams.uniform_ints(10, 5000, 1000) + 1
    o.apply(ops)
    r.apply(ops)
    for _ in range(7):
        o.add_node()
        r.add_node()
    ops2 = streams.random_stream(27, 5000, seed=10, p_delete=0.1)
    o.apply(ops2)
    r.apply(ops2)
    _same(o, r, "add_node interleaved")


@pytest.mark.ref
@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built (reference tree absent)")
def test_consumer_restatement_vs_reference_templates_rmat(streams):
    """the reference's bfs.h / pagerank.h templates on the reference PCSR vs tests/helpers.py:reference_consumers on the
    oracle, on a larger graph than the committed fixture (RMAT scale 14, 200 K edges, deletes)"""
    from helpers import reference_consumers
    scale, m = 14, 200_000
    n = 1 << scale
    s, d = streams.rmat_edges(scale, m, seed=51)
    ops = streams.adds(s, d)
    dele = ops[::6].copy()
    dele[:, 2] = 0
    ops = np.concatenate([ops, dele])
    r, o = RefPCSR(n), Oracle(n)
    r.apply(ops)
    o.apply(ops)
    vals = (streams.uniform_ints(52, n, 1000).astype(np.float32) / np.float32(3.0)).astype(np.float32)
    for start in (0, 5, int(s[4321])):
        lv, pr = reference_consumers(o, start, vals)
        np.testing.assert_array_equal(lv, r.bfs(start))
    assert pr.tobytes() == r.pagerank(vals).tobytes()
