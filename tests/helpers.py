"""Shared test helpers: fixture loading, state digests, package loading."""
import hashlib
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLDEN = os.path.join(HERE, "golden")
PKG_DIR = os.path.join(ROOT, "parallel-packed-csr_amd")


def _load(name, path):
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=[os.path.dirname(path)]
                                                  if path.endswith("__init__.py") else None)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_streams():
    return _load("ppcsr_streams", os.path.join(PKG_DIR, "streams.py"))


def load_pkg():
    """the product package (directory name has a hyphen, so it is loaded by path as `ppcsr_amd`)."""
    return _load("ppcsr_amd", os.path.join(PKG_DIR, "__init__.py"))


def digest(items, nodes, geom):
    h = hashlib.sha256()
    h.update(np.array(geom, np.int64).tobytes())
    h.update(np.ascontiguousarray(items).tobytes())
    h.update(np.ascontiguousarray(nodes).tobytes())
    return h.hexdigest()


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


GOLDEN_SINGLE = [
    "hub_1e4_insert_then_delete", "random_2e4_n1000", "insert_50k_n2000", "mixed_existing_80k_n2000",
    "rmat12_core60k_mixed40k", "random_2e4_n1000_lockfree", "dense_n40_grow_shrink", "add_node_empty_then_edges",
]


def replay_golden(make_engine, name, check_every=True):
    """Apply fixture `name` to make_engine(n, lock_search) checkpoint by checkpoint; return the engine.
    Asserts geometry + digest at every checkpoint and the raw state when stored."""
    g = golden(name)
    n, lock = int(g["n"]), bool(int(g["lock_search"]))
    eng = make_engine(n, lock)
    if name == "add_node_empty_then_edges":
        for _ in range(5):
            eng.add_node()
    ops, cps = g["ops"], g["checkpoints"]
    prev = 0
    for i, c in enumerate(cps):
        eng.apply(ops[prev:c])
        prev = int(c)
        if check_every or i == len(cps) - 1:
            geom = eng.geometry()
            assert tuple(geom) == tuple(int(x) for x in g["geoms"][i]), f"{name}: geometry at op {c}: {geom} vs {g['geoms'][i]}"
            items, nodes = eng.state()
            assert digest(items, nodes, geom) == str(g["digests"][i]), f"{name}: state digest differs at checkpoint {i} (op {c})"
    if "items" in g.files:
        items, nodes = eng.state()
        np.testing.assert_array_equal(items, g["items"])
        np.testing.assert_array_equal(nodes, g["nodes"])
    return eng


def reference_consumers(oracle, start, node_values):
    """src/utility/bfs.h:15-36 and src/utility/pagerank.h:15-29 evaluated on an oracle's state with numpy: BFS levels
    (uint32, 0xFFFFFFFF = unreachable) and the one-step PageRank push with fp32 additions in the reference's order
    (np.add.at applies its updates one by one in index order = ascending source, neighbours in slot order)."""
    import numpy as np
    items, nodes = oracle.state()
    n = len(nodes)
    live = (items[:, 2] != 0) & (items[:, 1] != 0xFFFFFFFF)
    live[-1] = False  # slot N-1 is never part of a neighbourhood
    src = items[live, 0].astype(np.int64)
    dst = items[live, 1].astype(np.int64)
    # BFS by levels
    lv = np.full(n, 0xFFFFFFFF, np.uint32)
    lv[start] = 0
    order = np.argsort(src, kind="stable")
    s_sorted, d_sorted = src[order], dst[order]
    rows = np.searchsorted(s_sorted, np.arange(n + 1))
    front = np.array([start], np.int64)
    level = 0
    while len(front):
        nb = np.concatenate([d_sorted[rows[u]:rows[u + 1]] for u in front]) if len(front) else np.empty(0, np.int64)
        nb = nb[nb < n]
        nb = np.unique(nb[lv[nb] == 0xFFFFFFFF])
        level += 1
        lv[nb] = level
        front = nb
    # PageRank push: edges are visited in array order == (source ascending, slot order)
    vals = np.asarray(node_values, np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        contrib = (vals / nodes[:, 2].astype(np.float32)).astype(np.float32)
    out = np.zeros(n, np.float32)
    ok = dst < n
    np.add.at(out, dst[ok], contrib[src[ok]])
    return lv, out


import numpy as np  # noqa: E402


def edge_view(items, nodes):
    """layout-independent view of a PCSR state: CSR (row lengths, dests, values) + num_neighbors"""
    live = (items[:, 2] != 0) & (items[:, 1] != 0xFFFFFFFF)
    live[-1] = False
    deg = np.bincount(items[live, 0], minlength=len(nodes))
    return deg, items[live, 1], items[live, 2], nodes[:, 2]


def check_pma_invariants(items, nodes):
    N, n = len(items), len(nodes)
    live = items[:, 2] != 0
    nul = items[~live]
    assert (nul[:, 0] == 0xFFFFFFFF).all() and (nul[:, 1] == 0).all()
    sent = live & (items[:, 1] == 0xFFFFFFFF)
    pos = np.nonzero(sent)[0]
    assert len(pos) == n
    np.testing.assert_array_equal(items[pos, 0], np.arange(n, dtype=np.uint32))
    np.testing.assert_array_equal(nodes[:, 0], pos.astype(np.uint32))
    np.testing.assert_array_equal(nodes[:-1, 1], nodes[1:, 0])
    assert nodes[-1, 1] == N - 1
    e = np.nonzero(live & ~sent)[0]
    owner = np.searchsorted(pos, e, side="right") - 1
    np.testing.assert_array_equal(items[e, 0], owner.astype(np.uint32))
    key = items[e, 0].astype(np.uint64) << np.uint64(32) | items[e, 1].astype(np.uint64)
    assert (np.diff(key.astype(np.int64)) > 0).all()  # sorted and unique inside every neighbourhood


# sources of 80 ascending-dest inserts into the last three of 4096 vertices: the 47th makes slide_right run off the end
# of the array (PCSR.cpp:347-351) — found by tools/fuzz_parity.py (seed 5069); the reference recovers through slide_left
SLIDE_OFF_END_SRC = [4094, 4094, 4094, 4093, 4094, 4095, 4094, 4095, 4094, 4093, 4093, 4094, 4095, 4095, 4094, 4095, 4093, 4095, 4094, 4094, 4094, 4093, 4093, 4095, 4094, 4093, 4094, 4093, 4095, 4095, 4093, 4094, 4094, 4095, 4094, 4093, 4095, 4095, 4093, 4093, 4094, 4093, 4095, 4094, 4095, 4093, 4094, 4093, 4095, 4093, 4095, 4095, 4095, 4093, 4093, 4095, 4093, 4093, 4095, 4095, 4093, 4094, 4094, 4094, 4094, 4093, 4095, 4094, 4094, 4093, 4095, 4094, 4094, 4095, 4094, 4094, 4095, 4094, 4094, 4093]


def slide_off_end_stream():
    import numpy as np
    src = np.array(SLIDE_OFF_END_SRC, np.uint32)
    return np.stack([src, np.arange(len(src), dtype=np.uint32), np.ones(len(src), np.uint32)], 1)


def live_triples(items, base=0):
    """every edge of a raw edges[] export as (src + base, dest, value), array order (sentinels and empty slots dropped; the
    last slot included — see Engine::export_triples_device)"""
    it = np.asarray(items, np.uint32).reshape(-1, 3)
    live = (it[:, 2] != 0) & (it[:, 1] != 0xFFFFFFFF) & (it[:, 2] != 0xFFFFFFFF)
    out = it[live].copy()
    out[:, 0] += np.uint32(base)
    return out


def check_repartitioned(new_states, old_states, old_starts, new_starts, total_n, build_bulk):
    """the rule of pppcsr_repartition (include/ppcsr.h) checked on raw states: new_states[k] / old_states[k] = (items, nodes) of
    partition k after / before.  A partition whose vertex range did not change is untouched bit for bit.  A changed one
    equals what the SAME bulk path builds from the edges that fall into its new range — build_bulk(size, adds) -> (items,
    nodes) of a fresh engine of `size` vertices bulk-built from `adds` (partition-local src, ascending (src, dest)) — except
    for num_neighbors, which every vertex carries over unchanged (it is a counter of calls, not the degree)."""
    P = len(old_states)
    end = lambda st, k: int(st[k + 1]) if k + 1 < P else int(total_n)
    changed = [int(old_starts[k]) != int(new_starts[k]) or end(old_starts, k) != end(new_starts, k) for k in range(P)]
    moved = [live_triples(old_states[k][0], int(old_starts[k])) for k in range(P) if changed[k]]
    moved = np.concatenate(moved) if moved else np.zeros((0, 3), np.uint32)
    nn_old = np.concatenate([np.asarray(old_states[k][1], np.uint32).reshape(-1, 3)[:, 2] for k in range(P)])  # by global vertex
    for k in range(P):
        it, nd = new_states[k]
        nd = np.asarray(nd, np.uint32).reshape(-1, 3)
        lo, hi = int(new_starts[k]), end(new_starts, k)
        assert len(nd) == hi - lo, f"partition {k}: {len(nd)} vertices, range [{lo}, {hi})"
        if not changed[k]:
            assert np.array_equal(it, old_states[k][0]) and np.array_equal(nd, np.asarray(old_states[k][1], np.uint32).reshape(-1, 3)), f"partition {k} was to stay untouched"
            continue
        sub = moved[(moved[:, 0] >= lo) & (moved[:, 0] < hi)].copy()
        sub[:, 0] -= np.uint32(lo)
        if hi > lo:
            eit, end_ = build_bulk(hi - lo, sub)
            end_ = np.asarray(end_, np.uint32).reshape(-1, 3)
            assert np.array_equal(it, eit), f"partition {k}: edges[] differ from the bulk build of its new range"
            assert np.array_equal(nd[:, :2], end_[:, :2]), f"partition {k}: vertex ranges differ from the bulk build"
        np.testing.assert_array_equal(nd[:, 2], nn_old[lo:hi], err_msg=f"partition {k}: num_neighbors not carried over")
        if hi > lo:
            check_pma_invariants(np.asarray(it, np.uint32).reshape(-1, 3), nd)
        if len(sub):
            np.testing.assert_array_equal(live_triples(it, lo), sub + np.array([lo, 0, 0], np.uint32))
