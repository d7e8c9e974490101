"""GPU: adversarial streams for the speculative scheduler — heavy skew (Zipf sources, config #5's shape), duplicate
storms, delete/re-insert churn on hubs, runs of sorted dests — each compared bit-for-bit with the oracle.  These target
the validation logic (per-vertex sentinel tracking, weak duplicate writes, region / growth-zone rules, rollback)."""
import numpy as np
import pytest

from helpers import load_pkg
from oracle_lib import Oracle

pytestmark = pytest.mark.gpu

CONFIGS = {
    "default": {},
    "tiny-regions": dict(region_slots=64, opt_horizon=2048, epoch_ops=65536),
    "wide": dict(opt_horizon=16384),
}


@pytest.fixture(scope="module")
def pkg():
    p = load_pkg()
    p.load_library()
    return p


def _run(pkg, n, ops, cfg, chunks=1):
    eng, o = pkg.PCSR(n), Oracle(n)
    for k, v in CONFIGS[cfg].items():
        eng.set_option(k, v)
    step = (len(ops) + chunks - 1) // chunks
    for lo in range(0, len(ops), step):
        eng.apply(ops[lo:lo + step])
        o.apply(ops[lo:lo + step])
        assert eng.geometry() == o.geometry()
        ei, en = eng.state()
        oi, on = o.state()
        assert np.array_equal(en, on), f"nodes differ after {lo + step}"
        assert np.array_equal(ei, oi), f"edges differ after {lo + step}"
    assert eng.check_invariants() == 0
    se, so = eng.stats(), o.stats()
    for k in ("redistribute_calls", "redistribute_slots", "not_found", "duplicates", "double_calls", "half_calls"):
        assert se[k] == so[k], k
    return se


@pytest.mark.parametrize("cfg", list(CONFIGS))
def test_zipf_sources(pkg, streams, cfg):
    n, m = 1 << 16, 600_000
    src = streams.permute_labels(streams.zipf_sources(n, m, seed=4, alpha=1.2), n)
    dst = streams.uniform_ints(5, m, n)
    ops = streams.adds(src, dst)
    _run(pkg, n, ops, cfg, chunks=2)


@pytest.mark.parametrize("cfg", list(CONFIGS))
def test_zipf_unpermuted_hot_prefix(pkg, streams, cfg):
    """every hot vertex sits at the start of the array (config #5's cascade stress)"""
    n, m = 1 << 15, 400_000
    src = streams.zipf_sources(n, m, seed=6, alpha=1.2)
    dst = streams.uniform_ints(7, m, 1 << 20)
    ops = streams.adds(src, dst)
    dele = ops[::3].copy()
    dele[:, 2] = 0
    _run(pkg, n, np.concatenate([ops, dele]), cfg, chunks=2)


@pytest.mark.parametrize("cfg", ["default", "tiny-regions"])
def test_duplicate_storm_and_small_dests(pkg, streams, cfg):
    """hot (src, dst) pairs with small dsts: duplicates and inserts right behind the sentinels of hub vertices"""
    n, m = 4096, 300_000
    src = streams.uniform_ints(11, m, 8) * 7  # 8 hubs
    dst = streams.uniform_ints(12, m, 40)     # 40 distinct small dests: mostly duplicates
    val = streams.uniform_ints(13, m, 1000) + 1
    a = np.stack([src, dst, val], 1).astype(np.uint32)
    b = np.stack([streams.uniform_ints(14, m, n), streams.uniform_ints(15, m, n), np.ones(m, np.uint32)], 1).astype(np.uint32)
    ops = np.empty((2 * m, 3), np.uint32)
    ops[0::2], ops[1::2] = a, b
    dele = a[::5].copy()
    dele[:, 2] = 0
    _run(pkg, n, np.concatenate([ops, dele, a[::7]]), cfg, chunks=3)


@pytest.mark.parametrize("cfg", ["default", "tiny-regions"])
def test_sorted_runs_into_many_vertices(pkg, streams, cfg):
    """ascending and descending dest runs interleaved over many vertices (long slides, appends next to sentinels)"""
    n, per = 512, 400
    v = np.repeat(np.arange(n, dtype=np.uint32), per)
    up = np.tile(np.arange(per, dtype=np.uint32) * 3 + 5, n)
    down = np.tile((np.arange(per, dtype=np.uint32)[::-1]) * 3 + 6, n)
    order = np.argsort(streams.splitmix64(np.arange(n * per, dtype=np.uint64)), kind="stable")
    a = np.stack([v, up, np.ones(n * per, np.uint32)], 1)[order]
    b = np.stack([v, down, np.ones(n * per, np.uint32)], 1)[order[::-1]]
    _run(pkg, n, np.concatenate([a, b]).astype(np.uint32), cfg, chunks=2)
