"""Model check behind the engine's 64-ary search (pma_device.h: pma_search).

The reference's gap-aware binary search (PCSR.cpp:427-502) is restated here as a scalar walk — the same statement as the
engine's device routine, which is bit-exact against the reference through the parity suites — and two properties are
checked on random sorted neighbourhoods with gaps:

  1. the value it returns equals the walk started from the TIGHT bracket (last live slot with dest < key, first live slot
     with dest >= key), i.e. it does not depend on the path;
  2. it equals the walk started from ANY bracket that satisfies the walk's own invariants (start: first slot of the range
     or a live slot with dest < key; end: end of the range or a live slot with dest > key).

Property 2 is what allows the engine to tighten the bracket 64 samples at a time before running the walk."""
import random


def walk(vals, dests, start, end, key):
    while start + 1 < end:
        mid = (start + end) // 2
        found = False
        check = idest = None
        for p in range(2 * (end - start) + 4):  # probe order: mid, mid+1, mid-1, mid+2, ...
            d = (p + 1) >> 1
            if p & 1:
                slot, valid = mid + d, d < end - mid
            else:
                slot, valid = mid - d, d <= mid - start
            if valid and vals[slot] != 0:
                found, check, idest = True, slot, dests[slot]
                break
        if not found or check == start:
            if found and key <= idest:
                return check
            return mid
        if key == idest:
            return check
        if key < idest:
            end = check
        else:
            start = check
    if end < start:
        start = end
    if vals[start] != 0 and key <= dests[start]:
        return start
    return end


def random_range(rng):
    n = rng.randint(1, 60)
    start0 = rng.randint(0, 3)
    end0 = start0 + n
    size = end0 + 2
    dens = rng.choice([0.1, 0.3, 0.6, 0.9, 1.0])
    vals, dests = [0] * size, [0] * size
    cur = 0
    for s in range(start0, end0):
        if rng.random() < dens:
            cur += rng.randint(1, 3)
            vals[s], dests[s] = 1, cur
    if rng.random() < 0.8:  # the next vertex's sentinel sits at `end0`
        vals[end0], dests[end0] = 1, 10 ** 9
    return vals, dests, start0, end0, cur


def test_walk_result_is_that_of_the_tight_bracket():
    rng = random.Random(1)
    for _ in range(30000):
        vals, dests, start0, end0, cur = random_range(rng)
        key = rng.randint(0, cur + 2)
        L, R = start0, end0
        for s in range(start0, end0):
            if vals[s]:
                if dests[s] < key:
                    L = s
                else:
                    R = s
                    break
        want = walk(vals, dests, start0, end0, key)
        if R < end0 and vals[R] and dests[R] == key:
            assert want == R
        else:
            assert walk(vals, dests, L, R, key) == want


def test_walk_result_is_independent_of_the_starting_bracket():
    rng = random.Random(2)
    checked = 0
    for _ in range(40000):
        vals, dests, start0, end0, cur = random_range(rng)
        key = rng.randint(0, cur + 2)
        if any(vals[s] and dests[s] == key for s in range(start0, end0)):
            continue  # an exact hit is returned as soon as any probe meets it
        lows = [start0] + [s for s in range(start0, end0) if vals[s] and dests[s] < key]
        highs = [end0] + [s for s in range(start0, end0) if vals[s] and dests[s] > key]
        want = walk(vals, dests, start0, end0, key)
        assert walk(vals, dests, rng.choice(lows), rng.choice(highs), key) == want
        checked += 1
    assert checked > 10000
