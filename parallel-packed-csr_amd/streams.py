"""Synthetic update streams.  The reference bundles no graph data, so every input
is synthesised here with a counter-based generator: element i of a stream depends only on
(seed, i), so the C++ CLI generator, the Python tests and bench.py produce identical streams on
any machine / numpy version (no reliance on numpy's Generator bit streams).

An update stream is an (n,3) uint32 array of (src, dst, op): op 0 = delete, op >= 1 = add with
edge value `op` (the reference pools always add with value 1, thread_pool.cpp:44-48).
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    """Vectorised splitmix64 finaliser of the uint64 array/scalar `x` (counter-based hash)."""
    with np.errstate(over="ignore"):
        z = (np.asarray(x, dtype=np.uint64) + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def _u01(seed, idx):
    """uniform [0,1) doubles from (seed, idx) with 53 random bits."""
    with np.errstate(over="ignore"):
        h = splitmix64(np.uint64(seed) * np.uint64(0xD1342543DE82EF95) + np.asarray(idx, dtype=np.uint64))
    return (h >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def uniform_ints(seed, count, hi, offset=0):
    """count integers in [0,hi) from counters offset..offset+count-1."""
    idx = np.arange(offset, offset + count, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = splitmix64(np.uint64(seed) * np.uint64(0xD1342543DE82EF95) + idx)
    return (h % np.uint64(hi)).astype(np.uint32)


def rmat_edges(scale, count, seed, a=0.57, b=0.19, c=0.19, offset=0, chunk=1 << 20):
    """`count` RMAT edges over 2**scale vertices: one uniform draw per (edge, bit)."""
    src = np.zeros(count, np.uint32)
    dst = np.zeros(count, np.uint32)
    for lo in range(0, count, chunk):
        m = min(chunk, count - lo)
        e = np.arange(offset + lo, offset + lo + m, dtype=np.uint64)
        s = np.zeros(m, np.uint32)
        d = np.zeros(m, np.uint32)
        for bit in range(scale):
            u = _u01(seed, e * np.uint64(64) + np.uint64(bit))
            sb = (u >= a + b).astype(np.uint32)                       # quadrants c,d set the src bit
            db = (((u >= a) & (u < a + b)) | (u >= a + b + c)).astype(np.uint32)  # quadrants b,d set the dst bit
            s = (s << np.uint32(1)) | sb
            d = (d << np.uint32(1)) | db
        src[lo:lo + m] = s
        dst[lo:lo + m] = d
    return src, dst


def rmat_edges_folded(n, scale, count, seed, offset=0):
    """config #4/#5 graph: RMAT ids of 2**scale >= n vertices folded into [0, n) by `% n`"""
    s, d = rmat_edges(scale, count, seed=seed, offset=offset)
    return (s % np.uint32(n)).astype(np.uint32), (d % np.uint32(n)).astype(np.uint32)


def adds(src, dst, value=1):
    return np.stack([src, dst, np.full(len(src), value, np.uint32)], 1).astype(np.uint32)


def random_stream(n, count, seed, p_delete=0.0):
    """uniform endpoints in [0,n); each op is a delete with probability p_delete (mostly misses)."""
    s = uniform_ints(seed, count, n)
    d = uniform_ints(seed + 1000003, count, n)
    op = (_u01(seed + 2000003, np.arange(count, dtype=np.uint64)) >= p_delete).astype(np.uint32)
    return np.stack([s, d, op], 1).astype(np.uint32)


def mixed_existing_stream(core, fresh, seed):
    """alternate ADD (rows of `fresh`) / DELETE (distinct rows of `core`, sampled without replacement)."""
    m = len(fresh)
    key = splitmix64(np.uint64(seed) + np.arange(len(core), dtype=np.uint64))
    pick = np.argsort(key, kind="stable")[:m]
    out = np.empty((2 * m, 3), np.uint32)
    out[0::2] = fresh
    out[1::2, 0] = core[pick, 0]
    out[1::2, 1] = core[pick, 1]
    out[1::2, 2] = 0
    return out


_ZIPF_CDF = {}


def zipf_sources(n, count, seed, alpha=1.2, offset=0):
    """Zipf(alpha) ranks in [0,n) by inverse CDF on a precomputed table (config #5); counters offset..offset+count-1."""
    cdf = _ZIPF_CDF.get((n, alpha))
    if cdf is None:
        w = 1.0 / np.power(np.arange(1, n + 1, dtype=np.float64), alpha)
        cdf = np.cumsum(w)
        cdf /= cdf[-1]
        _ZIPF_CDF.clear()  # (one table at a time: 80 MB at n = 10 M)
        _ZIPF_CDF[(n, alpha)] = cdf
    u = _u01(seed, np.arange(offset, offset + count, dtype=np.uint64))
    return np.minimum(np.searchsorted(cdf, u, side="right"), n - 1).astype(np.uint32)


def permute_labels(v, n):
    """v -> (v * 2654435761) mod n  (bijective when gcd(n, 2654435761) == 1)."""
    return ((v.astype(np.uint64) * np.uint64(2654435761)) % np.uint64(n)).astype(np.uint32)
