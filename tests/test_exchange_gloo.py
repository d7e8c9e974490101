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
