"""CPU, world_size 2 (gloo): the N > 1 routing path — owner bucketing + all-to-all — delivers to every rank
exactly its partition's subsequence of the global stream, in stream order, with partition-local sources.
Checked against the oracle's PPPCSR routing rule and, end to end, against per-partition oracle states."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import ROOT, digest, load_streams

sys.path.insert(0, os.path.join(ROOT, "tests"))


def _load_exchange():
    import importlib.util
    spec = importlib.util.spec_from_file_location("ppcsr_exchange", os.path.join(ROOT, "parallel-packed-csr_amd", "exchange.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _worker(rank, world, port, n_global, blocks, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ex = _load_exchange()
    outs = []
    for blk in blocks[rank]:
        t = torch.from_numpy(blk.view(np.int32))
        outs.append(ex.exchange_ops(t, n_global, world).numpy().view(np.uint32).copy())
    q.put((rank, outs))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_global", [1000, 1003])
def test_exchange_world2(n_global):
    from oracle_lib import Oracle, OraclePPPCSR
    streams = load_streams()
    world, steps, m = 2, 3, 4000
    # global stream per step = concat over ranks of their blocks (rank r holds block r)
    blocks = [[streams.random_stream(n_global, m, seed=50 + 7 * k + r, p_delete=0.3) for k in range(steps)] for r in range(world)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_global, blocks, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    pp = OraclePPPCSR(n_global, True, 1, world)
    parts = [Oracle(int(pp.partition(k).get_n())) for k in range(world)]
    for k in range(steps):
        glob = np.concatenate([blocks[r][k] for r in range(world)])
        owner = np.array([pp.get_partition(int(s)) for s in glob[:, 0]])
        pp.apply(glob)
        for r in range(world):
            exp = glob[owner == r].copy()
            exp[:, 0] -= np.uint32(pp.partition_start(r))
            np.testing.assert_array_equal(got[r][k], exp)
            parts[r].apply(got[r][k])
    for r in range(world):
        a, b = parts[r], pp.partition(r)
        assert digest(*a.state(), a.geometry()) == digest(*b.state(), b.geometry())


def _worker_parts(rank, world, port, n_global, n_parts, blocks, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ex = _load_exchange()
    outs = []
    for blk in blocks[rank]:
        t = torch.from_numpy(blk.view(np.int32))
        parts, cnts = ex.exchange_parts(t, n_global, n_parts, world, cap=3000)  # blocks are ragged: agree on the capacity
        assert [p.shape[0] for p in parts] == cnts
        outs.append([p.numpy().view(np.uint32).copy() for p in parts])
    q.put((rank, outs))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_global,n_parts", [(1000, 4), (1003, 8), (1000, 2)])
def test_exchange_parts_world2(n_global, n_parts):
    """strong-scaling form (config #4/#5): n_parts partitions over 2 ranks, ONE padded all-to-all per batch; every local
    partition receives exactly its subsequence of the global stream (block 0 then block 1), partition-local sources"""
    from oracle_lib import OraclePPPCSR
    streams = load_streams()
    world, steps = 2, 2
    sizes = [3000, 1]  # ragged: an almost empty block on one rank
    blocks = [[streams.random_stream(n_global, sizes[(r + k) % 2] if k else 2500, seed=90 + 5 * k + r, p_delete=0.25) for k in range(steps)]
              for r in range(world)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker_parts, args=(r, world, port, n_global, n_parts, blocks, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    pp = OraclePPPCSR(n_global, True, world, n_parts // world)
    ppr = n_parts // world
    for k in range(steps):
        glob = np.concatenate([blocks[r][k] for r in range(world)])
        owner = np.array([pp.get_partition(int(s)) for s in glob[:, 0]])
        for r in range(world):
            for ql in range(ppr):
                part = r * ppr + ql
                exp = glob[owner == part].copy()
                exp[:, 0] -= np.uint32(pp.partition_start(part))
                np.testing.assert_array_equal(got[r][k][ql], exp, err_msg=f"step {k} partition {part}")


# ---- the engine library's own exchange (pppcsr_xchg_pack / _layout / _apply) past one rank: two CPU-emulator processes, gloo as
# ---- the carrier.  Everything but the ncclSend/ncclRecv calls of pppcsr_exchange_apply is the code the GPU build runs.
def _sim_tune(pp, ks):
    for k in ks:
        e = pp.partition(k)
        for key, v in dict(mode=1, opt_horizon=64, epoch_ops=1024, region_slots=64, small_batch=0, big_grid=2, big_min=512,
                           big_window=131072, max_horizon=32, min_horizon=4, init_horizon=8, rounds_per_sync=2).items():
            e.set_option(key, v)


def _gloo_exchange(pp, world, rank, ppr, blk, finish="apply"):
    """one batch through pack -> (gloo) -> layout -> apply.  blk: this rank's block, a (n,3) uint32 array or a (address, n) pair"""
    import ctypes
    addr, n = (blk.ctypes.data, len(blk)) if isinstance(blk, np.ndarray) else blk
    counts, d_send = pp.xchg_pack(addr, n)
    send = np.ctypeslib.as_array((ctypes.c_uint32 * (3 * max(n, 1))).from_address(d_send)).reshape(-1, 3)[:n] if n else np.zeros((0, 3), np.uint32)
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    # carrier step 1: the counts (every rank learns what every source holds for its partitions)
    allc = [torch.zeros(world * ppr, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(allc, torch.from_numpy(counts.astype(np.int64)))
    recv_counts = np.concatenate([allc[r].numpy()[rank * ppr:(rank + 1) * ppr] for r in range(world)]).astype(np.uint64)
    dst = pp.xchg_layout(recv_counts)
    # carrier step 2: the rows, one message per peer, scattered to the addresses the layout asked for
    reqs, inbox = [], {}
    for r in range(world):
        if r == rank:
            continue
        out = torch.from_numpy(send[off[r * ppr]:off[(r + 1) * ppr]].copy().view(np.int32).reshape(-1))
        inbox[r] = torch.zeros(int(recv_counts[r * ppr:(r + 1) * ppr].sum()) * 3, dtype=torch.int32)
        if out.numel():
            reqs.append(dist.isend(out, r))
        if inbox[r].numel():
            reqs.append(dist.irecv(inbox[r], r))
    for q in reqs:
        q.wait()
    for r in range(world):
        rows = send[off[r * ppr]:off[(r + 1) * ppr]] if r == rank else inbox[r].numpy().view(np.uint32).reshape(-1, 3)
        rows = np.ascontiguousarray(rows)
        o = 0
        for q in range(ppr):
            c = int(recv_counts[r * ppr + q])
            if c:
                ctypes.memmove(dst[r * ppr + q], rows[o:o + c].ctypes.data, c * 12)
            o += c
    {"apply": pp.xchg_apply, "set_nn": pp.xchg_set_num_neighbors, "bulk": pp.xchg_bulk_build}[finish]()


def _worker_native(rank, world, port, n_global, n_parts, blocks, new_starts, after, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from helpers import load_pkg
    from test_sim_engine import SIM_SO
    pkg = load_pkg()
    lib = pkg.load_library(SIM_SO)
    ppr = n_parts // world
    mine = range(rank * ppr, (rank + 1) * ppr)
    pp = pkg.PPPCSR(n_global, numDomain=world, partitionsPerDomain=ppr, lib=lib, local=(rank * ppr, ppr, 0))
    _sim_tune(pp, mine)
    pp.xchg_create(world, rank)
    for blk in blocks[rank]:
        _gloo_exchange(pp, world, rank, ppr, blk)
    snap = [pp.partition(k).state() for k in mine]
    # repartition across the ranks: the edges on the move and the counters beside them are two more blocks of the exchange
    moved, nn_recs = pp.repartition_export(new_starts)
    _sim_tune(pp, mine)
    _gloo_exchange(pp, world, rank, ppr, moved, finish="bulk")
    _gloo_exchange(pp, world, rank, ppr, nn_recs, finish="set_nn")
    _sim_tune(pp, mine)
    mid = [pp.partition(k).state() for k in mine]
    _gloo_exchange(pp, world, rank, ppr, after[rank])
    end = [pp.partition(k).state() for k in mine]
    q.put((rank, [snap, mid, end]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_global,n_parts", [(600, 4), (403, 2)])
def test_native_exchange_and_repartition_world2_hostsim(n_global, n_parts):
    from oracle_lib import Oracle, OraclePPPCSR
    from helpers import check_repartitioned
    from test_sim_engine import build_sim
    build_sim()
    streams = load_streams()
    world, ppr = 2, n_parts // 2
    sizes = [[1500, 1], [0, 1200]]  # ragged and empty blocks
    blocks = [[streams.random_stream(n_global, sizes[k][r], seed=70 + 5 * k + r, p_delete=0.25) for k in range(2)] for r in range(world)]
    after = [streams.random_stream(n_global, 700, seed=170 + r, p_delete=0.4) for r in range(world)]
    pp = OraclePPPCSR(n_global, True, world, ppr)
    old = np.array([pp.partition_start(k) for k in range(n_parts)], np.uint64)
    # vertex ranges move across the rank boundary (partition ppr - 1 / ppr) as well as inside a rank
    new = old.copy()
    new[ppr] = old[ppr] - min(37, int(old[ppr] - old[ppr - 1]))
    if ppr > 1:
        new[1] = old[1] + 11
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker_native, args=(r, world, port, n_global, n_parts, blocks, new, after, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for k in range(2):
        pp.apply(np.concatenate([blocks[r][k] for r in range(world)]))
    parts = [pp.partition(k) for k in range(n_parts)]
    flat = lambda stage: [st for r in range(world) for st in got[r][stage]]
    for k, (items, nodes) in enumerate(flat(0)):
        oi, on = parts[k].state()
        assert np.array_equal(items, oi) and np.array_equal(nodes, on), f"partition {k} after the exchanged batches"
    # the repartition rule, checked on the raw states (the bulk path itself run on a fresh emulator engine)
    from helpers import load_pkg
    from test_sim_engine import SIM_SO
    pkg = load_pkg()
    lib = pkg.load_library(SIM_SO)

    def build_bulk(size, adds):
        e = pkg.PCSR(size, lib=lib)
        e.bulk_build(adds)
        out = e.state()
        e.close()
        return out

    check_repartitioned(flat(1), flat(0), old, new, n_global, build_bulk)
    # updates after the rebuild are exact again: oracles started from the rebuilt states
    parts = [Oracle.from_state(*st) for st in flat(1)]
    glob = np.concatenate(after)
    own = np.searchsorted(new, glob[:, 0], side="right") - 1
    for k in range(n_parts):
        sub = glob[own == k].copy()
        sub[:, 0] -= np.uint32(new[k])
        parts[k].apply(sub)
        oi, on = parts[k].state()
        items, nodes = flat(2)[k]
        assert np.array_equal(items, oi) and np.array_equal(nodes, on), f"partition {k}: updates after the repartition"


# ---- world 8, one partition per rank (the config #4 layout): pppcsr_exchange_apply itself, past one rank --------------------------
# The emulator build carries the exchange over a shared-memory mailbox (tests/hostsim/sim_xchg.cpp) in place of RCCL: everything
# in capi.cc's exchange_run — bucketing, the counts step, the status step, the rows, the per-partition apply, and what every rank
# does when ONE rank fails — is the code the GPU runs.
def _worker_exchange_run(rank, world, uid, n_global, blocks, q):
    import ctypes
    from helpers import load_pkg
    from test_sim_engine import SIM_SO
    pkg = load_pkg()
    lib = pkg.load_library(SIM_SO)
    lib.ppcsr_sim_fail_alloc_after.argtypes = [ctypes.c_int]
    pp = pkg.PPPCSR(n_global, numDomain=world, partitionsPerDomain=1, lib=lib, local=(rank, 1, 0))
    _sim_tune(pp, [rank])
    pp.comm_create(uid, world, rank, 0)
    log = []
    for step, blk in enumerate(blocks[rank]):
        kind, rows = blk
        buf = np.ascontiguousarray(rows)
        ptr, n = (buf.ctypes.data if len(buf) else 0), len(buf)
        if kind == "bad_bucketing":   # a null block of non-zero length: this rank's bucketing fails before anything collective
            ptr, n = 0, 5
        if kind == "bad_layout":      # the receive buffers cannot be laid out (the next device allocation fails) after the counts step
            lib.ppcsr_sim_fail_alloc_after(1)
        try:
            pp.exchange_apply(ptr, n)
            log.append("ok")
        except pkg.PpcsrError as e:
            log.append("error: " + str(e)[:160])
        lib.ppcsr_sim_fail_alloc_after(0)
    q.put((rank, log, pp.partition(rank).state()))


def test_exchange_apply_world8_one_partition_per_rank_and_failing_ranks():
    from oracle_lib import OraclePPPCSR
    from helpers import load_pkg
    from test_sim_engine import SIM_SO, build_sim
    build_sim()
    pkg = load_pkg()
    lib = pkg.load_library(SIM_SO)
    streams = load_streams()
    world, n_global = 8, 1003
    sizes = [[900, 0, 700, 400], [0, 1200, 1, 400], [350, 350, 0, 400], [1, 0, 0, 400], [800, 10, 900, 400], [0, 0, 0, 400], [600, 600, 600, 400],
             [77, 1500, 300, 400]]  # ragged and empty blocks, four good steps
    good = [[streams.random_stream(n_global, sizes[r][k], seed=300 + 11 * k + r, p_delete=0.25) for k in range(4)] for r in range(world)]
    # step order: good, good, [rank 3's bucketing fails], good, [rank 6 cannot lay out its receive buffers], good
    blocks = []
    for r in range(world):
        b = [("good", good[r][0]), ("good", good[r][1]), ("bad_bucketing" if r == 3 else "good", streams.random_stream(n_global, 50, seed=900 + r)),
             ("good", good[r][2]), ("bad_layout" if r == 6 else "good", streams.random_stream(n_global, 60 if r == 6 else 3000, seed=950 + r)), ("good", good[r][3])]
        # (rank 6's own block is small — its send buffer is big enough already — and what the others send it is more than it has
        #  ever received: the one allocation of its step is the receive buffer's)
        blocks.append(b)
    uid = pkg.PPPCSR.comm_unique_id(lib)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_exchange_run, args=(r, world, uid, n_global, blocks, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, log, st = q.get(timeout=900)
        got[r] = (log, st)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        log = got[r][0]
        assert [x == "ok" for x in log] == [True, True, False, True, False, True], (r, log)  # EVERY rank reports both failed batches — nobody hangs
    assert "null ops" in got[3][0][2] and "peer rank failed to bucket" in got[0][0][2]
    assert "memory" in got[6][0][4] and "could not lay out" in got[1][0][4]
    # the failed batches moved nothing anywhere; the good ones arrived in global stream order per partition
    pp = OraclePPPCSR(n_global, True, world, 1)
    for k in range(4):
        pp.apply(np.concatenate([good[r][k] for r in range(world)]))
    for r in range(world):
        oi, on = pp.partition(r).state()
        items, nodes = got[r][1]
        assert np.array_equal(items, oi) and np.array_equal(nodes, on), f"partition {r}"
