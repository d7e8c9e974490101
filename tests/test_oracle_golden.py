"""CPU: the oracle (oracle/ppcsr_oracle.c) against the golden fixtures generated from the real
reference (tests/golden/make_golden.py).  This is what pins the oracle on machines without
/root/reference."""
import numpy as np
import pytest

from helpers import GOLDEN_SINGLE, digest, golden, replay_golden
from oracle_lib import Oracle, OraclePPPCSR


@pytest.mark.parametrize("name", GOLDEN_SINGLE)
def test_oracle_matches_golden(name):
    eng = replay_golden(lambda n, lock: Oracle(n, lock_search=lock), name)
    eng.close()


def test_oracle_pppcsr_matches_golden():
    g = golden("pppcsr_p8_n1000")
    pp = OraclePPPCSR(1000, True, 1, 8)
    assert pp.num_partitions() == 8
    part = np.array([pp.get_partition(v) for v in range(1000)])
    np.testing.assert_array_equal(part, g["part_of_vertex"])
    np.testing.assert_array_equal([pp.partition_start(k) for k in range(8)], g["starts"])
    pp.apply(g["ops"])
    for k in range(8):
        p = pp.partition(k)
        assert p.get_n() == int(g["sizes"][k])
        items, nodes = p.state()
        assert digest(items, nodes, p.geometry()) == str(g["digests"][k])
    pp.close()


def test_oracle_api_semantics():
    """DataStructureTest.cpp:12-49 restated (Initialization / add_node / add_edge / remove_edge)."""
    o = Oracle(10)
    assert o.get_n() == 10
    o.add_edge(11, 1, 1)  # silently ignored: no such source (PCSR.cpp:1375)
    o.add_edge(0, 1, 1)
    assert o.edge_exists(0, 1)
    assert len(o.get_neighbourhood(0)) == 1 and len(o.get_neighbourhood(2)) == 0
    o.remove_edge(0, 1)
    assert not o.edge_exists(0, 1)
    o.remove_edge(0, 1)  # miss: num_neighbors underflows (PCSR.cpp:747)
    _, nodes = o.state()
    assert nodes[0, 2] == 0xFFFFFFFF
    o.add_node()
    assert o.get_n() == 11
    e = Oracle(0)
    assert e.get_n() == 0
    e.add_node()
    assert e.get_n() == 1 and len(e.get_neighbourhood(0)) == 0


def test_oracle_duplicate_counts_and_value_overwrite():
    o = Oracle(10)
    for v in (5, 6, 7):
        o.add_edge(3, 4, v)
    items, nodes = o.state()
    assert nodes[3, 2] == 3  # num_neighbors counts calls, not distinct edges (PCSR.cpp:1392)
    live = items[(items[:, 2] != 0) & (items[:, 1] != 0xFFFFFFFF) & (items[:, 0] == 3)]
    assert len(live) == 1 and live[0, 2] == 7


def test_redistribute_positions_properties():
    from oracle_lib import oracle_lib
    L = oracle_lib()
    rng = np.random.default_rng(0)
    for _ in range(300):
        ln = 1 << int(rng.integers(3, 12))
        idx = int(rng.integers(0, 1 << 20)) * ln
        j = int(rng.integers(1, ln + 1))
        out = np.zeros(j, np.uint64)
        L.po_redistribute_positions(idx, ln, j, out.ctypes.data)
        assert out[0] == idx and (np.diff(out.astype(np.int64)) >= 1).all() and out[-1] < idx + ln
