"""GPU (MI355X): the BASELINE.json configurations at FULL size, slot by slot against the oracle.

* configs #2 / #3 (and config #5's stream shape) on the RMAT scale-20 / 10 M-edge graph: the state after the core load
  (the regime with exclusive updates, doublings and rollbacks), after 1 M inserts, after 1 M mixed updates, after 1 M
  Zipf(1.2)-source inserts — each from the same core snapshot;
* configs #4 / #5 (n = 10 000 000, 100 M-edge core, P = 8, partitionSize = 1 250 000): ONE partition at a time, fed with
  its subsequence of the global stream exactly as PPPCSR routes it (PPPCSR.cpp:46-66) — partition 0 with raw labels (the
  one that holds 44 % of the edges) and partition 3 with permuted labels (it owns the second-hottest Zipf vertex), the
  latter also under config #5's 10 M-update Zipf stream.
The partitions of a PPPCSR are independent PCSRs, so one partition against the oracle at full size is the full-size
check of that partition in the 8-GPU run; bench.py checks all resident partitions after its timed region as well.

* EVERY partition of configs #4 / #5 (raw and permuted labels) and configs #2 / #3 once more, by digest: sha256(geometry, items[],
  nodes[]) of the engine's state against tests/golden/config_digests.json, which holds the digests of the states the REAL
  reference produced (tests/golden/make_config_digests.py, run in the build container where /root/reference exists) — no
  100 M-edge CPU replay on the GPU box, and the comparison is with the reference itself, not its restatement."""
import json
import os
import time

import numpy as np
import pytest

from helpers import GOLDEN, digest, load_pkg
from oracle_lib import Oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    p = load_pkg()
    p.load_library()
    return p


def _same(eng, o, label):
    assert tuple(eng.geometry()) == tuple(o.geometry()), f"{label}: geometry {eng.geometry()} vs {o.geometry()}"
    ei, en = eng.state()
    oi, on = o.state()
    if not np.array_equal(en, on):
        bad = np.nonzero((en != on).any(1))[0]
        raise AssertionError(f"{label}: nodes[] differ at {len(bad)} vertices, first {bad[:8]}")
    if not np.array_equal(ei, oi):
        bad = np.nonzero((ei != oi).any(1))[0]
        raise AssertionError(f"{label}: edges[] differ at {len(bad)} slots, first {bad[:8]}")


def _branch(eng, core_oracle, ops, label):
    """from the core snapshot: apply `ops` to the engine and to a copy of the core oracle, compare, report the rates"""
    eng.restore()
    s0 = eng.stats()
    eng.apply(ops)
    s1 = eng.stats()
    o = core_oracle.clone()
    t0 = time.time()
    o.apply(ops)
    dt = time.time() - t0
    _same(eng, o, label)
    o.close()
    print(f"{label}: {len(ops)} updates, device {s1['last_batch_ms']:.1f} ms = {len(ops) / s1['last_batch_ms'] / 1e3:.1f} M/s, "
          f"rounds {s1['rounds'] - s0['rounds']}, exclusive {s1['exclusive_ops'] - s0['exclusive_ops']}, "
          f"rollbacks {s1['rollbacks'] - s0['rollbacks']}; oracle {dt:.1f} s = {len(ops) / dt / 1e6:.2f} M/s on one host thread")


def test_configs_2_3_and_zipf_shape_full_size(pkg, streams):
    scale, n = 20, 1 << 20
    s, d = streams.rmat_edges(scale, 10_000_000, seed=1)
    core = streams.adds(s, d)
    eng, o = pkg.PCSR(n), Oracle(n)
    eng.apply(core)
    t0 = time.time()
    o.apply(core)
    print(f"core: oracle {time.time() - t0:.1f} s; engine {eng.stats()}")
    _same(eng, o, "after the 10 M-edge core load")
    st = eng.stats()
    assert st["N"] == 1 << 24 and st["logN"] == 32
    assert st["exclusive_ops"] > 0 and st["double_calls"] >= 2  # the load is where the scheduler is stressed most
    eng.snapshot()
    s2, d2 = streams.rmat_edges(scale, 1_000_000, seed=2)
    fresh = streams.adds(s2, d2)
    _branch(eng, o, fresh, "config #2: 1 M inserts")
    _branch(eng, o, streams.mixed_existing_stream(core, fresh[:500_000], seed=3), "config #3: 1 M mixed 50/50")
    zs = streams.zipf_sources(n, 1_000_000, seed=4, alpha=1.2)
    zd = streams.uniform_ints(11, 1_000_000, n)
    _branch(eng, o, streams.adds(zs, zd), "config #5 stream shape on the config #2 graph: 1 M Zipf(1.2)-source inserts")
    o.close()
    eng.close()


# ---- configs #4 / #5 -------------------------------------------------------------------------------------------------
N4, SCALE4, CORE4, UPD4, P4 = 10_000_000, 24, 100_000_000, 10_000_000, 8


@pytest.fixture(scope="module")
def graph4(streams):
    """raw (folded) ids of the 100 M-edge core and of the 10 M-insert stream, generated once for the module"""
    t0 = time.time()
    cs, cd = streams.rmat_edges_folded(N4, SCALE4, CORE4, seed=1)
    us, ud = streams.rmat_edges_folded(N4, SCALE4, UPD4, seed=2)
    print(f"config #4 streams generated in {time.time() - t0:.1f} s")
    return cs, cd, us, ud


def _partition_subsequence(streams, s, d, part, permute):
    """PPPCSR routing (PPPCSR.cpp:20-29, 46-66): owner = src / floor(n / P) (last partition takes the remainder); the
    partition sees its updates in stream order with src made local and dest left global"""
    if permute:
        s = streams.permute_labels(s, N4)
        d = streams.permute_labels(d, N4)
    ps = N4 // P4
    own = np.minimum(s // np.uint32(ps), P4 - 1)
    m = own == part
    return streams.adds(s[m] - np.uint32(part * ps), d[m])


# (slot by slot against the oracle for ONE partition — it says where a difference is; the digests below cover all sixteen)
@pytest.mark.parametrize("labels,part", [("permuted", 3)])
def test_config4_one_partition_full_size(pkg, streams, graph4, labels, part):
    cs, cd, us, ud = graph4
    permute = labels == "permuted"
    ps = N4 // P4
    size = ps if part < P4 - 1 else N4 - part * ps
    assert size == 1_250_000
    core = _partition_subsequence(streams, cs, cd, part, permute)
    upd = _partition_subsequence(streams, us, ud, part, permute)
    print(f"config #4 partition {part} ({labels}): {len(core)} core edges ({100.0 * len(core) / CORE4:.1f} % of the graph), {len(upd)} inserts")
    eng, o = pkg.PCSR(size), Oracle(size)
    eng.apply(core)
    t0 = time.time()
    o.apply(core)
    print(f"core: oracle {time.time() - t0:.1f} s")
    _same(eng, o, f"partition {part} ({labels}) after its core subsequence")
    eng.snapshot()
    _branch(eng, o, upd, f"config #4 partition {part} ({labels}): its share of the 10 M inserts")
    if permute:
        # config #5 on the same partition: 10 M updates, src = Zipf(1.2) rank through the same permutation (rank 2 lands in
        # this partition: permute(1) = 4 435 761), dst uniform
        zs = streams.zipf_sources(N4, UPD4, seed=4, alpha=1.2)
        zd = streams.uniform_ints(11, UPD4, N4)
        zupd = _partition_subsequence(streams, zs, zd, part, True)
        hot = np.bincount(zupd[:, 0]).max()
        print(f"config #5 partition {part}: {len(zupd)} updates, {hot} of them into one vertex")
        _branch(eng, o, zupd, f"config #5 partition {part} (permuted): its share of the 10 M Zipf(1.2) updates")
    o.close()
    eng.close()


# ---- every configuration, every partition, against digests held by the real reference --------------------------------------------
def _digests():
    return json.load(open(os.path.join(GOLDEN, "config_digests.json")))["digests"]


def _dg(eng):
    return digest(*eng.state(), eng.geometry())


def test_configs_2_3_zipf_digests_of_the_reference(pkg, streams):
    want = _digests()
    n = 1 << 20
    s, d = streams.rmat_edges(20, 10_000_000, seed=1)
    core = streams.adds(s, d)
    eng = pkg.PCSR(n)
    eng.apply(core)
    assert _dg(eng) == want["config2_core"], "state after the 10 M-edge core load"
    eng.snapshot()
    s2, d2 = streams.rmat_edges(20, 1_000_000, seed=2)
    fresh = streams.adds(s2, d2)
    zipf = streams.adds(streams.zipf_sources(n, 1_000_000, seed=4, alpha=1.2), streams.uniform_ints(11, 1_000_000, n))
    for key, ops in (("config2_inserts", fresh), ("config3_mixed", streams.mixed_existing_stream(core, fresh[:500_000], seed=3)),
                     ("config5_shape_zipf", zipf)):
        eng.restore()
        eng.apply(ops)
        assert _dg(eng) == want[key], key
    eng.close()


@pytest.fixture(scope="module")
def graph4_all(streams, graph4):
    """config #4 / #5 inputs with raw and permuted labels (the permutation applied once for all partitions)"""
    cs, cd, us, ud = graph4
    zs, zd = streams.zipf_sources(N4, UPD4, seed=4, alpha=1.2), streams.uniform_ints(11, UPD4, N4)
    raw = {"cs": cs, "cd": cd, "us": us, "ud": ud, "zs": zs, "zd": zd}
    perm = {k: streams.permute_labels(v, N4) for k, v in raw.items()}
    return {"raw": raw, "permuted": perm}


def _sub(streams, s, d, part):
    ps = N4 // P4
    m = np.minimum(s // np.uint32(ps), P4 - 1) == part
    return streams.adds(s[m] - np.uint32(part * ps), d[m])


@pytest.mark.parametrize("labels", ["permuted", "raw"])
def test_config4_and_5_every_partition_digests_of_the_reference(pkg, streams, graph4_all, labels):
    want = _digests()
    g = graph4_all[labels]
    ps = N4 // P4
    t0 = time.time()
    for part in range(P4):
        size = ps if part < P4 - 1 else N4 - part * ps
        core = _sub(streams, g["cs"], g["cd"], part)
        eng = pkg.PCSR(size)
        eng.apply(core)
        assert _dg(eng) == want[f"config4_{labels}_p{part}_core"], f"partition {part} ({labels}) after its core subsequence ({len(core)} edges)"
        eng.snapshot()
        eng.apply(_sub(streams, g["us"], g["ud"], part))
        assert _dg(eng) == want[f"config4_{labels}_p{part}_inserts"], f"config #4 partition {part} ({labels}) after its share of the 10 M inserts"
        if labels == "permuted":  # config #5: the partition's share of the 10 M Zipf(1.2) updates, from the same core
            eng.restore()
            eng.apply(_sub(streams, g["zs"], g["zd"], part))
            assert _dg(eng) == want[f"config5_{labels}_p{part}_zipf"], f"config #5 partition {part} ({labels})"
        eng.close()
        print(f"partition {part} ({labels}): ok after {time.time() - t0:.0f} s", flush=True)


def test_bench_two_ranks_on_one_gpu():
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one rank per process), both ranks on the one GPU of
    this box with gloo as the process-group backend (RCCL refuses two ranks on one device): the N > 1 code path of the
    benchmark — per-rank blocks, owner exchange, per-partition apply, cross-rank parity of every resident partition — on a
    small config #4 graph.  The exchange carrier of the real multi-GPU run (native RCCL) is covered by
    test_native_rccl_exchange_single_rank and tests/test_exchange_gloo.py."""
    import json
    import os
    import subprocess
    import sys
    from helpers import ROOT
    port = 29600 + (os.getpid() % 300)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--config", "4",
           "--vertices", "200000", "--scale", "18", "--core-edges", "2000000", "--batch", "200000", "--steps", "1", "--warmup", "1",
           "--no-secondary", "--no-cpu-baseline", "--no-profile"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["parity_checked"] is True and out["value"] > 0
    assert "8 partition(s), 4 per GPU x 2 GPU(s)" in out["config"]["parallelism"]


def test_bench_config4_legs_small():
    """`bench.py --config 4` with its secondary legs (raw labels, raw labels after pppcsr_repartition to balanced ranges, the
    config #5 stream) on a small graph: every leg produces a number, the checked ones are bit-exact"""
    import json
    import os
    import subprocess
    import sys
    from helpers import ROOT
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "4", "--vertices", "200000", "--scale", "18", "--core-edges", "2000000",
           "--batch", "200000", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-profile"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["parity_checked"] is True
    for leg in ("raw_labels", "raw_labels_repartitioned", "config5_zipf"):
        assert leg in out and out[leg]["value"] > 0, (leg, {k: v for k, v in out.items() if k.endswith("_error")})
    rp = out["raw_labels_repartitioned"]
    assert rp["starts"][0] == 0 and len(rp["starts"]) == 8 and rp["repartition_s"] > 0
    assert out["config5_zipf"]["parity_checked"] is True
