#!/usr/bin/env python3
"""bench.py — edge-updates/s of the MI355X PMA engine on BASELINE.json's workloads.

One "step" = one pass of the hot path over one batch of synthetic updates already resident in HBM (every step starts
from the same core graph: the device-to-device restore of the core snapshot is part of the step and inside the timed
region).  `--config` picks the BASELINE.json configuration; the default is #2 on one GPU and #4 on several:

  #2  RMAT scale-20 core (10 M edges, a/b/c = .57/.19/.19), one partition, 1 M fresh RMAT inserts per step
  #3  same core, 1 M mixed updates per step: inserts alternating with deletes of existing core edges
  #4  n = 10 000 000 vertices (scale-24 RMAT ids folded % n), 100 M-edge core, 10 M inserts per step, P = 8 vertex-range
      partitions (partitionSize = 1 250 000, PPPCSR.cpp:20-29).  STRONG scaling: the same graph and the same stream on
      1, 2, 4 or 8 ranks, rank r holding partitions [r * 8/N, (r + 1) * 8/N) (the reference's partitions_per_domain);
      every rank holds a contiguous block of the global stream, buckets it by owner (stable) and swaps buckets with ONE
      all-to-all (RCCL over xGMI); received buckets are concatenated in source-rank order == global stream order.
      Headline: labels permuted by v -> v * 2654435761 mod n (balanced partitions); `raw_labels` beside it (44 % of the
      edges in partition 0: the skew bounds the speed-up exactly as in the reference).
  #5  the config #4 graph, 10 M updates per step whose sources follow Zipf(1.2) (hot-vertex rebalance cascades)

Launch for N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N
Prints ONE JSON line on rank 0 (DESIGN.md section 6 defines the fields).  Unless --no-check, the state of the timed
engines is compared slot by slot with the reference / oracle after the timed region (`parity_checked`).
"""
import argparse
import glob
import hashlib
import importlib.util
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _load(name, path, pkg=False):
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(
        name, path, submodule_search_locations=[os.path.dirname(path)] if pkg else None)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md; ~6.3 TB/s achievable)
T_BENCH0 = time.time()


def csrc_sha256():
    """identity of the kernel sources a PMC record belongs to (the GPU box has no .git): sha256 over csrc/'s sources"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "parallel-packed-csr_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".cc", ".hip", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def log(rank, *a):
    if rank == 0:
        print(f"[bench +{time.time() - T_BENCH0:5.1f}s]", *a, file=sys.stderr, flush=True)


class Workload:
    """the synthetic graph + update streams of one BASELINE config; element i of every stream depends on (seed, i) only,
    so any rank can regenerate any block"""

    def __init__(self, streams, cfg, n, scale, core_edges, batch, permute):
        self.st, self.cfg, self.n, self.scale, self.core_edges, self.batch, self.permute = streams, cfg, n, scale, core_edges, batch, permute
        self.folded = (1 << scale) != n

    def _labels(self, s, d):
        if self.permute:
            s = self.st.permute_labels(s, self.n)
            d = self.st.permute_labels(d, self.n)
        return s, d

    def _rmat(self, count, seed, offset):
        if self.folded:
            return self.st.rmat_edges_folded(self.n, self.scale, count, seed=seed, offset=offset)
        return self.st.rmat_edges(self.scale, count, seed=seed, offset=offset)

    def core(self, offset, count):
        s, d = self._labels(*self._rmat(count, 1, offset))
        return self.st.adds(s, d)

    def updates(self, k, offset, count, core_for_mixed=None):
        """block [offset, offset + count) of update batch k"""
        if self.cfg == 5:  # src = Zipf(1.2) rank, dst uniform, all ADD
            s = self.st.zipf_sources(self.n, count, seed=4 + 10 * k, alpha=1.2, offset=offset)
            d = self.st.uniform_ints(11 + 10 * k, count, self.n, offset=offset)
            s, d = self._labels(s, d)
            return self.st.adds(s, d)
        s, d = self._labels(*self._rmat(count, 2 + 10 * k, offset))
        ops = self.st.adds(s, d)
        if self.cfg == 3:
            ops = self.st.mixed_existing_stream(core_for_mixed, ops[:count // 2], seed=3 + 10 * k)
        return ops

    def name(self, P, world):
        lab = "permuted" if self.permute else "raw"
        if self.cfg == 2:
            w = f"config#2 {self.batch} random inserts on RMAT scale-{self.scale} / {self.core_edges}-edge core"
        elif self.cfg == 3:
            w = f"config#3 {self.batch} mixed 50/50 insert+delete on RMAT scale-{self.scale} / {self.core_edges}-edge core"
        elif self.cfg == 4:
            w = f"config#4 {self.batch} inserts on {self.n}-vertex / {self.core_edges}-edge RMAT (scale-{self.scale} ids % n)"
        else:
            w = f"config#5 {self.batch} Zipf(1.2)-source inserts on {self.n}-vertex / {self.core_edges}-edge RMAT (scale-{self.scale} ids % n)"
        if P > 1:
            w += f", {P} vertex-range partitions over {world} GPU(s), labels {lab}"
        else:
            w += f", 1 partition, labels {lab}"
        return w


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=0, choices=[0, 2, 3, 4, 5], help="BASELINE.json config (default: 2 on one GPU, 4 on several)")
    ap.add_argument("--mixed", action="store_true", help="= --config 3")
    ap.add_argument("--zipf", action="store_true", help="config #5's stream shape on the config #2 graph (one GPU stress case)")
    ap.add_argument("--scale", type=int, default=0, help="override the RMAT scale")
    ap.add_argument("--vertices", type=int, default=0, help="override n (configs 4/5)")
    ap.add_argument("--core-edges", type=int, default=0, help="override the core size (whole graph)")
    ap.add_argument("--batch", type=int, default=0, help="override the updates per step (whole job)")
    ap.add_argument("--parts", type=int, default=0, help="override the number of partitions (configs 4/5: 8)")
    ap.add_argument("--labels", choices=["permuted", "raw"], default=None)
    ap.add_argument("--distinct-batches", type=int, default=4, help="update batches generated (steps cycle through them)")
    ap.add_argument("--no-check", action="store_true", help="skip the slot-by-slot parity check after the timed region")
    ap.add_argument("--check", action="store_true", help="(parity is checked by default; kept for compatibility)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ref-cli", action="store_true", help="skip the multi-threaded runs of the reference's own CLI binary")
    ap.add_argument("--no-profile", action="store_true", help="skip the HIP-event replay that yields the roofline block")
    ap.add_argument("--no-secondary", action="store_true", help="skip scan / rebalance / consumers / raw-label / zipf legs")
    ap.add_argument("--mode", type=int, default=-1, help="0 strict prefix rounds, 1 speculative rounds (engine default)")
    ap.add_argument("--opt", action="append", default=[], help="engine option key=value (repeatable)")
    ap.add_argument("--markers", action="store_true", help="launch marker kernels around the timed region / isolated kernels (tools/roofline_profile.sh)")
    ap.add_argument("--exchange", choices=["auto", "torch", "native"], default="auto",
                    help="carrier of the owner exchange for N > 1: the engine library's own RCCL send/recv (pppcsr_exchange_apply; no torch "
                         "in the data path; default) or torch.distributed all_to_all_single (RCCL inside PyTorch).  auto = native after a "
                         "preflight that runs small ragged blocks through BOTH carriers and compares the partitions bit for bit; if the "
                         "native carrier errs or disagrees, the run says so and continues on the torch carrier")
    ap.add_argument("--backend", default="nccl", help="process-group backend; 'gloo' only for functional tests of the N > 1 path on one GPU")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    N = args.gpus
    if world != N:
        raise SystemExit(f"--gpus {N} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {N}")
    if not torch.cuda.is_available():
        raise SystemExit("no GPU: the engine is HIP-only (no CPU fallback)")
    ndev = torch.cuda.device_count()
    dev_id = local_rank % max(ndev, 1)  # (one rank per GPU in real runs; ranks share a GPU only in the gloo functional test)
    torch.cuda.set_device(dev_id)
    dev = torch.device("cuda", dev_id)
    if N > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend, rank=rank, world_size=world)

    pkg = _load("ppcsr_amd", os.path.join(ROOT, "parallel-packed-csr_amd", "__init__.py"), pkg=True)
    streams = _load("ppcsr_streams", os.path.join(ROOT, "parallel-packed-csr_amd", "streams.py"))
    exch = _load("ppcsr_exchange", os.path.join(ROOT, "parallel-packed-csr_amd", "exchange.py"))
    pkg.load_library()  # in-tree HIP build; raises if missing

    cfg = args.config or (3 if args.mixed else (2 if N == 1 else 4))
    if cfg in (2, 3):
        # one partition per GPU; with N > 1 (only on request) the per-GPU work is fixed: weak scaling
        scale = (args.scale or 20) + int(np.log2(N))
        n_global = 1 << scale
        core_edges = args.core_edges or 10_000_000 * N
        batch = args.batch or 1_000_000 * N
        P = args.parts or N
        scaling = "weak"
        permute = (args.labels or ("permuted" if P > 1 else "raw")) == "permuted"
    else:
        n_global = args.vertices or 10_000_000
        scale = args.scale or 24
        core_edges = args.core_edges or 100_000_000
        batch = args.batch or 10_000_000
        P = args.parts or 8
        scaling = "strong"
        permute = (args.labels or "permuted") == "permuted"
    assert P % N == 0, "--gpus must divide the number of partitions"
    ppr = P // N
    wl = Workload(streams, 5 if args.zipf else cfg, n_global, scale, core_edges, batch, permute)
    starts, sizes = exch.partition_layout(n_global, P)
    single = (P == 1)
    xdev = dev if args.backend == "nccl" else torch.device("cpu")  # where the exchange runs
    assert core_edges % N == 0 and batch % N == 0
    my_core, my_batch = core_edges // N, batch // N

    def to_dev(a):
        return torch.from_numpy(a.view(np.int32)).to(xdev if N > 1 else dev)

    # ---- carrier of the exchange (N > 1) ---------------------------------------------------------------------------
    carrier = {"use_native": False, "note": None}

    def new_uid():
        # (bootstrap only: 128 bytes.  Rank 0 always takes part in the broadcast, also when it could not make an id —
        #  the other ranks would wait for it for ever otherwise — and then every rank fails the same way)
        uid = [None]
        if rank == 0:
            try:
                uid = [pkg.PPPCSR.comm_unique_id()]
            except Exception as ex:  # noqa: BLE001
                log(0, f"WARNING: no RCCL unique id: {ex}")
        dist.broadcast_object_list(uid, src=0)
        if uid[0] is None:
            raise RuntimeError("rank 0 could not create an RCCL unique id")
        return uid[0]

    def torch_exchange_apply(pp_, ops_dev, n_, cap=None):
        parts, cnts = exch.exchange_parts(ops_dev, n_, P, N, dist.group.WORLD, cap=cap)
        parts = [t.to(dev) for t in parts]
        torch.cuda.current_stream().synchronize()  # the exchange ran on torch's stream; the engines apply on their own
        pp_.apply_parts_device(rank * ppr, [t.data_ptr() for t in parts], cnts)
        return parts

    def preflight_native():
        """ragged blocks (one of them a single row) through both carriers into two small PPPCSRs; True when every local partition
        agrees bit for bit on every rank.  Every step that holds a collective is followed by an agreement (all-reduce MIN of "I am
        fine"): a rank that failed locally never leaves its peers alone inside a collective of the NEXT step — all ranks stop
        together and take the torch carrier."""
        n_small = 512 * P
        state = {"ok": 1, "why": ""}
        objs = {"a": None, "b": None}

        def agree():
            flag = torch.tensor([state["ok"]], dtype=torch.int32, device=xdev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return bool(flag.item())

        def step(fn):
            """run fn unless this rank has already failed; returns True when EVERY rank is still fine"""
            if state["ok"]:
                try:
                    fn()
                except Exception as ex:  # noqa: BLE001 — any failure of the native carrier selects the other one
                    state["ok"], state["why"] = 0, f"{type(ex).__name__}: {ex}"
            return agree()

        def create():
            objs["a"] = pkg.PPPCSR(n_small, numDomain=N, partitionsPerDomain=ppr, local=(rank * ppr, ppr, dev_id))
            objs["b"] = pkg.PPPCSR(n_small, numDomain=N, partitionsPerDomain=ppr, local=(rank * ppr, ppr, dev_id))

        try:
            good = step(create)
            uid = [None]
            if good:  # (the id itself: a broadcast every rank takes part in, whatever rank 0 managed)
                try:
                    uid[0] = new_uid()
                except Exception as ex:  # noqa: BLE001
                    state["ok"], state["why"] = 0, f"{type(ex).__name__}: {ex}"
                good = agree()
            if good:
                good = step(lambda: objs["a"].comm_create(uid[0], N, rank, dev_id))  # (ncclCommInitRank: collective)
            for k in range(3):
                if not good:
                    break
                m = [4096, 1 if (rank + k) % 2 else 3000, 257][k]
                blk = streams.random_stream(n_small, m, seed=1000 + 17 * k + rank, p_delete=0.3)
                t = to_dev(blk)
                torch.cuda.synchronize()
                # (pppcsr_exchange_apply reports a failing rank on every rank — its steps stay collective — so the agreement after
                #  it is about what it returned; the torch carrier's all_to_all follows only when all ranks got through)
                good = step(lambda: objs["a"].exchange_apply(t.data_ptr(), m))
                if good:
                    good = step(lambda: torch_exchange_apply(objs["b"], t, n_small, cap=4096))

            def compare():
                for q in range(ppr):
                    ea, eb = objs["a"].partition(rank * ppr + q), objs["b"].partition(rank * ppr + q)
                    sa, sb = ea.state(), eb.state()
                    if ea.geometry() != eb.geometry() or not (np.array_equal(sa[0], sb[0]) and np.array_equal(sa[1], sb[1])):
                        raise RuntimeError(f"partition {rank * ppr + q} differs between the carriers")

            if good:
                good = step(compare)
        finally:
            for o in objs.values():
                if o is not None:
                    try:
                        o.close()
                    except Exception:  # noqa: BLE001
                        pass
        return good, state["why"]

    if N > 1 and args.exchange != "torch":
        if args.backend != "nccl":
            carrier["note"] = "torch carrier: the native exchange needs one GPU per rank (RCCL), this is a gloo functional run"
            if args.exchange == "native":
                raise SystemExit("--exchange native needs --backend nccl")
        elif args.exchange == "native":
            carrier["use_native"] = True
        else:
            good, why = preflight_native()
            carrier["use_native"] = good
            if not good:
                carrier["note"] = "native carrier failed its preflight on some rank, torch carrier used" + (f" (this rank: {why})" if why else "")
                log(0 if why else rank, "WARNING: " + carrier["note"])
            else:
                log(rank, "exchange preflight: native RCCL carrier == torch carrier on every rank, bit for bit")

    # ---- engines -------------------------------------------------------------------------------------------------
    def make_engines():
        if single:
            e = pkg.PCSR(n_global, device=dev_id)
            es = [e]
            pp = None
        else:
            pp = pkg.PPPCSR(n_global, numDomain=N, partitionsPerDomain=ppr, local=(rank * ppr, ppr, dev_id))
            es = [pp.partition(rank * ppr + q) for q in range(ppr)]
            if N > 1 and carrier["use_native"]:
                pp.comm_create(new_uid(), N, rank, dev_id)
        apply_options(es)
        return pp, es

    def apply_options(es):
        for e in es:
            if args.mode >= 0:
                e.set_option("mode", args.mode)
            for kv in args.opt:
                k, v = kv.split("=")
                e.set_option(k, int(v))

    def sum_stats(es):
        tot = {}
        for e in es:
            for k, v in e.stats().items():
                if k in ("N", "n", "logN", "H", "last_batch_ms", "last_batch_h2d_ms"):
                    tot.setdefault(k, []).append(v)
                else:
                    tot[k] = tot.get(k, 0) + v
        return tot

    def run_step(pp, es, ops_dev):
        """route (N > 1 or several partitions) + apply in stream order on this rank's partition(s)"""
        if single:
            if ops_dev.shape[0]:
                es[0].apply_device(ops_dev.data_ptr(), ops_dev.shape[0])
            return
        if N == 1:
            pp.apply_device(ops_dev.data_ptr(), ops_dev.shape[0])  # device bucketing + all partitions concurrently
            return
        if carrier["use_native"]:
            pp.exchange_apply(ops_dev.data_ptr(), ops_dev.shape[0])
            return
        torch_exchange_apply(pp, ops_dev, n_global)

    def run_workload(wl_, label, steps, warmup, want_profile, after_core=None):
        """core load (untimed), snapshot, warm-up, timed steps [, profiled replay]; returns a result dict.
        after_core(pp, es) -> es: hook between the core load and the snapshot (the repartitioning leg)"""
        t0 = time.time()
        core_blk = wl_.core(rank * my_core, my_core)
        nb = max(1, min(args.distinct_batches, warmup + steps))
        upd = [wl_.updates(k, rank * my_batch, my_batch, core_for_mixed=core_blk) for k in range(nb)]
        log(rank, f"{label}: generated core {len(core_blk)} + {nb} x {len(upd[0])} updates per rank in {time.time() - t0:.1f}s "
                  f"(n={wl_.n}, labels={'permuted' if wl_.permute else 'raw'})")
        pp, es = make_engines()
        t0 = time.time()
        core_dev = to_dev(core_blk)
        torch.cuda.synchronize()  # (the engines run on their own non-blocking streams)
        run_step(pp, es, core_dev)
        torch.cuda.synchronize()
        del core_dev
        st = sum_stats(es)
        log(rank, f"{label}: core loaded in {time.time() - t0:.1f}s: N={st['N']} logN={st['logN']} rounds={st['rounds']} "
                  f"exclusive={st['exclusive_ops']} doubles={st['double_calls']} rollbacks={st['rollbacks']}")
        if after_core is not None:
            es = after_core(pp, es)
        for e in es:
            e.snapshot()
        upd_dev = [to_dev(u) for u in upd]
        torch.cuda.synchronize()

        def step(k):
            for e in es:
                e.restore()
            run_step(pp, es, upd_dev[k % nb])

        for k in range(warmup):
            step(k)
        torch.cuda.synchronize()
        s0 = sum_stats(es)
        if N > 1:
            dist.barrier()
        torch.cuda.synchronize()
        mark = args.markers and label.startswith("config#") and single
        if mark:
            es[0].set_option("marker", 0)
        t_start = time.perf_counter()
        for k in range(warmup, warmup + steps):
            step(k)
        if mark:
            es[0].set_option("marker", 1)
        torch.cuda.synchronize()
        if N > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t_start
        s1 = sum_stats(es)
        if N > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=xdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        total_updates = len(upd[0]) * N * steps
        keys = ("rounds", "committed", "planned", "exclusive_ops", "rollbacks", "round_syncs", "redistribute_slots",
                "redistribute_calls", "ops_applied", "double_calls", "wasted_rounds")
        dstat = {k: s1[k] - s0[k] for k in keys if k in s1}
        dstat["updates_per_round"] = dstat["committed"] / max(dstat["rounds"], 1)
        dstat["replan_factor"] = dstat["planned"] / max(dstat["committed"], 1)
        dstat["device_ms_last_batch"] = max(s1["last_batch_ms"])
        res = {"value": total_updates / elapsed, "ms_per_step": elapsed / steps * 1e3, "engine": dstat,
               "N_slots": [int(x) for x in s1["N"]], "logN": int(s1["logN"][0]), "last_k": (warmup + steps - 1) % nb,
               "core_blk": core_blk, "upd": upd, "es": es, "pp": pp, "step": step}
        if single and N == 1:
            # end to end (the reference's own clock covers its whole apply phase: thread_pool.cpp:79,110): the
            # op array starts in HOST memory — ppcsr_apply_batch uploads it (12 B per update over PCIe) and applies it; wall clock
            # around restore + upload + apply, same steps as the device-only figure above.  Never `value`.
            t_e = time.perf_counter()
            h2d = 0.0
            for k in range(warmup, warmup + steps):
                es[0].restore()
                es[0].apply(upd[k % nb])
                h2d += es[0].stats()["last_batch_h2d_ms"]
            torch.cuda.synchronize()
            t_e = time.perf_counter() - t_e
            res["end_to_end"] = {"value": total_updates / t_e, "unit": "edge-updates/s", "ms_per_step": t_e / steps * 1e3,
                                 "h2d_ms_per_step": h2d / steps, "op_bytes_per_step": 12 * len(upd[0]),
                                 "note": "host-resident op array: upload (PCIe) + apply, wall clock incl. the snapshot restore; "
                                         "the headline `value` is device-only (ops already in HBM)"}
        # ---- roofline of the dominant round kernel: HIP events on the engines' own streams around every round kernel.
        # Recording ~5 events per round costs 25-35 % of throughput, so `value` comes from the un-instrumented timed
        # region and the SAME steps are replayed here with the events on (same inputs, same state).
        if want_profile:
            for e in es:
                e.set_option("profile", 1)
            p0 = sum_stats(es)
            for k in range(warmup, warmup + steps):
                step(k)
            torch.cuda.synchronize()
            p1 = sum_stats(es)
            for e in es:
                e.set_option("profile", 0)
            launches = p1["prof_launches"]
            kern = {"plan": p1["prof_plan_ms"], "check": p1["prof_check_ms"], "apply": p1["prof_apply_ms"], "compact": p1["prof_compact_ms"]}
            spec = args.mode != 0
            names = {"plan": "o_plan" if spec else "k_plan", "check": "o_check" if spec else "k_check",
                     "apply": "o_apply" if spec else "k_apply", "compact": "o_big"}
            dom = max(kern, key=kern.get)
            d = {k: p1[k] - p0[k] for k in ("redistribute_slots", "ops_applied")}
            # algorithmic bytes: 12 B op record + 24 B per slot of every redistribute() the reference makes
            alg_bytes = 12.0 * d["ops_applied"] + 24.0 * d["redistribute_slots"]
            avg_ms = kern[dom] / max(launches, 1)
            achieved = (alg_bytes / max(launches, 1)) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
            traffic, tsrc = None, None
            if cfg == 2 and N == 1:
                # HBM bytes per launch of the timed rounds, from the committed rocprofv3 --pmc passes of THIS command
                # (tools/roofline_profile.sh).  Only a record taken on exactly these kernel sources counts: the record
                # carries the sha256 of csrc/ it was measured on; anything else reports null and says which record is stale.
                here = csrc_sha256()
                for tpath in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_roofline.json")), reverse=True):
                    try:
                        pj = json.load(open(tpath))
                        rel = os.path.relpath(tpath, ROOT)
                        if pj.get("csrc_sha256") == here:
                            traffic = pj["timed_rounds"][names[dom]]["hbm_bytes_per_launch"]
                            tsrc = {"file": rel, "commit": pj.get("commit"), "csrc_sha256": here, "command": pj.get("command")}
                            break
                        tsrc = tsrc or {"stale": rel, "reason": "record was measured on other kernel sources",
                                        "record_csrc_sha256": pj.get("csrc_sha256"), "csrc_sha256": here}
                    except Exception:
                        continue
            res["roofline"] = {"bound": "hbm", "kernel": names[dom], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": tsrc,
                               "csrc_sha256": csrc_sha256(),
                               "launches": int(launches), "avg_launch_us": avg_ms * 1e3,
                               "alg_bytes_per_launch": alg_bytes / max(launches, 1),
                               "alg_bytes_per_update": alg_bytes / max(d["ops_applied"], 1),
                               "kernel_ms": {names[k]: round(v, 3) for k, v in kern.items() if v > 0},
                               "measured_on": "profiled replay of the timed steps (HIP events on the engine stream)"
                                              + ("" if len(es) == 1 else f", summed over this rank's {len(es)} partitions")}
            # leave the engines in the state of the last timed step for the parity check
        return res

    def expected_subsequences(wl_, kinds):
        """this rank's partitions' subsequences of the GLOBAL stream, regenerated from the counters block by block and
        routed with numpy (PPPCSR.cpp:46-66) — independent of the device bucketing and of the exchange.
        kinds: list of ('core',) / ('upd', k)"""
        ps = n_global // P
        out = [[] for _ in range(ppr)]
        for kind in kinds:
            for r in range(N):
                if kind[0] == "core":
                    g = wl_.core(r * my_core, my_core)
                else:
                    cb = wl_.core(r * my_core, my_core) if wl_.cfg == 3 else None
                    g = wl_.updates(kind[1], r * my_batch, my_batch, core_for_mixed=cb)
                own = np.minimum(g[:, 0].astype(np.int64) // ps, P - 1) if P > 1 else np.zeros(len(g), np.int64)
                for q in range(ppr):
                    part = rank * ppr + q
                    sub = g[own == part].copy()
                    sub[:, 0] -= np.uint32(starts[part])
                    out[q].append(sub)
        return [np.concatenate(x) for x in out]

    def parity_check(wl_, res):
        """every resident partition against the reference (N == 1, one partition, oracle/_ref built) or the oracle, fed with
        the partition's subsequence of core + the last timed batch; host threads (ctypes releases the GIL)"""
        from oracle_lib import Oracle, RefPCSR, have_ref
        if N == 1 and single:
            seqs = [np.concatenate([res["core_blk"], res["upd"][res["last_k"]]])]
        else:
            seqs = expected_subsequences(wl_, [("core",), ("upd", res["last_k"])])
        use_ref = have_ref() and single
        oks = [False] * ppr
        secs = [0.0] * ppr

        def one(q):
            part = rank * ppr + q
            o = (RefPCSR if use_ref else Oracle)(int(sizes[part]))
            t0 = time.time()
            o.apply(seqs[q])
            secs[q] = time.time() - t0
            e = res["es"][q]
            ei, en = e.state()
            oi, on = o.state()
            oks[q] = tuple(e.geometry()) == tuple(o.geometry()) and np.array_equal(ei, oi) and np.array_equal(en, on)
            o.close()
        th = [threading.Thread(target=one, args=(q,)) for q in range(ppr)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        ok = all(oks)
        if N > 1:
            t = torch.tensor([1 if ok else 0], dtype=torch.int64, device=xdev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            ok = bool(t.item())
        return ok, ("reference (oracle/_ref)" if use_ref else "oracle"), max(secs), sum(len(s) for s in seqs)

    # ================================================ headline ====================================================
    res = run_workload(wl, f"config#{wl.cfg}", args.steps, args.warmup, not args.no_profile)
    extra = {}
    if not args.no_check:
        t0 = time.time()
        ok, against, osecs, nops = parity_check(wl, res)
        log(rank, f"parity vs {against}: {'bit-exact' if ok else 'MISMATCH'} ({nops} updates replayed per rank, {osecs:.1f}s, total {time.time() - t0:.1f}s)")
        extra["parity_checked"] = ok
        extra["parity"] = {"against": against, "what": "N/logN/H + every slot of edges[] + every nodes[] triple of every partition, after core + the last timed batch",
                           "cpu_updates_per_s_one_thread": nops / max(osecs, 1e-9) / max(ppr, 1)}
        if not ok:
            raise SystemExit("PARITY FAILURE: engine state differs from the " + against)

    # ---- CPU baseline beside it (rank 0, N == 1): the reference's own CLI binary with its thread pools on this box's cores
    cpu = None
    if rank == 0 and N == 1 and not args.no_cpu_baseline:
        from oracle_lib import Oracle, RefPCSR, have_ref
        try:
            cores = len(os.sched_getaffinity(0))
        except Exception:
            cores = os.cpu_count() or 1
        kindc = "reference" if have_ref() else "port"
        Cls = RefPCSR if have_ref() else Oracle
        part0 = 0
        # bounded sample: ONE partition's subsequence (the reference's partitions are independent PCSRs), sequential order
        if single:
            seq_core, seq_upd = res["core_blk"], res["upd"][0]
        else:
            sq = expected_subsequences(wl, [("core",)])[part0], expected_subsequences(wl, [("upd", 0)])[part0]
            seq_core, seq_upd = sq
        c = Cls(int(sizes[part0]))
        tl = time.time()
        c.apply(seq_core)
        tl = time.time() - tl
        tc = time.time()
        c.apply(seq_upd)
        tc = time.time() - tc
        c.close()
        one_thread = len(seq_upd) / tc
        cpu = {"value": one_thread, "unit": "edge-updates/s", "cores": 1, "kind": kindc,
               "sample": (f"{'partition 0 of ' + str(P) + ': ' if not single else ''}{len(seq_core)}-edge core ({tl:.1f}s load, untimed) + "
                          f"one batch of {len(seq_upd)} updates in stream order on one host thread ({tc:.2f}s) — the sequential order is "
                          "the parity semantics"),
               "cores_visible": cores}
        if cfg in (4, 5) or args.zipf:
            # configs #4 / #5: the reference's own pools (-pppcsrnuma -partitions_per_domain=8, thread sweep) on the GPU box's host
            # cores.  One run loads the 100 M-edge core for 12 s and needs 1.6 GB of text files, so the sweep is a protocol of
            # its own (tools/cpu_baseline_protocol.py ... 4|5, same generator and seeds as this workload) whose committed
            # record is attached here; the live sample above stays the bounded one.
            ppath = os.path.join(ROOT, "profiles", f"r03_cpu_baseline_config{5 if (cfg == 5 or args.zipf) else 4}.json")
            if os.path.exists(ppath):
                try:
                    pj = json.load(open(ppath))
                    best = pj.get("best")
                    cpu["reference_pools"] = {"file": os.path.relpath(ppath, ROOT), "workload": pj.get("workload"), "best": best,
                                              "cores_available": pj.get("cores_available_to_this_process"), "cpu_model": pj.get("cpu_model"),
                                              "runs": {k: {"threads": v["threads"], "updates_per_s_mean": v["updates_per_s_mean"],
                                                           "updates_per_s_std": v["updates_per_s_std"], "repetitions": v["repetitions"]}
                                                       for k, v in pj.get("runs", {}).items()}}
                    # The live measurement stays cpu_baseline.value.  The recorded thread sweep is attached beside it and says whether
                    # it was taken on a host like this one (CPU model and CPUs online): a record from another machine is marked stale.
                    def _cpu_model():
                        try:
                            for ln in open("/proc/cpuinfo"):
                                if ln.startswith("model name"):
                                    return ln.split(":", 1)[1].strip()
                        except Exception:  # noqa: BLE001
                            pass
                        return None
                    same = (pj.get("cpu_model") == _cpu_model()) and (pj.get("cores_available_to_this_process") == cores)
                    cpu["reference_pools"]["same_host_class"] = bool(same)
                    if not same:
                        cpu["reference_pools"]["stale"] = (f"recorded on '{pj.get('cpu_model')}' with {pj.get('cores_available_to_this_process')} CPUs, "
                                                           f"this host is '{_cpu_model()}' with {cores}")
                    if best:
                        cpu["reference_pools"]["best_updates_per_s"] = pj["runs"][best]["updates_per_s_mean"]
                        cpu["reference_pools"]["best_threads"] = pj["runs"][best]["threads"]
                except Exception as e:
                    cpu["reference_pools_error"] = str(e)
        ref_cli = os.path.join(ROOT, "oracle", "_ref", "ref_cli")
        if os.path.exists(ref_cli) and not args.no_ref_cli and cfg == 2:
            # the north_star's comparison: the reference's -pppcsrnuma path on the same box's host cores.  Text edge lists in
            # /tmp; phase-2 time = the SECOND "Elapsed wall clock time" line (benchmark-strong-scaling.sh:114-126); 3
            # repetitions here, the full protocol (thread sweep, 5 repetitions) is tools/cpu_baseline_protocol.py ->
            # profiles/r02_cpu_baseline.json
            try:
                import subprocess
                import pandas as pd
                cf, uf = "/tmp/ppcsr_bench_core.txt", "/tmp/ppcsr_bench_upd.txt"
                pd.DataFrame(res["core_blk"][:, :2]).to_csv(cf, sep=" ", header=False, index=False)
                pd.DataFrame(res["upd"][0][:, :2]).to_csv(uf, sep=" ", header=False, index=False)
                share = min(cores, 16)  # the box's CPU share for one GPU
                runs = {}
                for label, thr, flags in ((f"pppcsrnuma_t{share}", share, ["-pppcsrnuma", "-partitions_per_domain=8"]),
                                          (f"ppcsr_t{share}", share, ["-ppcsr"])):
                    vals = []
                    for _ in range(5):  # (5 repetitions)
                        r = subprocess.run([ref_cli, f"-threads={thr}", f"-size={len(res['upd'][0])}", "-insert"] + flags +
                                           [f"-core_graph={cf}", f"-update_file={uf}"], capture_output=True, text=True, timeout=600)
                        el = [int(l.split(":")[1]) for l in r.stdout.splitlines() if l.startswith("Elapsed wall clock time")]
                        if len(el) >= 2 and el[1] > 0:
                            vals.append(len(res["upd"][0]) / (el[1] * 1e-3))
                    if vals:
                        runs[label] = {"updates_per_s_mean": float(np.mean(vals)), "updates_per_s_std": float(np.std(vals, ddof=1)) if len(vals) > 1 else 0.0,
                                       "repetitions": len(vals), "threads": thr}
                os.remove(cf)
                os.remove(uf)
                if runs:
                    best = max(runs, key=lambda k: runs[k]["updates_per_s_mean"])
                    cpu["multi_thread"] = {"best": best, "runs": runs,
                                           "note": "unmodified reference binary (oracle/_ref/ref_cli); multi-threaded runs are not "
                                                   "deterministic in layout (SURVEY.md section 8c); full sweep: profiles/r02_cpu_baseline.json"}
                    # the reported baseline is the reference's best figure on this box's cores
                    cpu["value"] = runs[best]["updates_per_s_mean"]
                    cpu["cores"] = runs[best]["threads"]
                    cpu["one_thread_sequential"] = one_thread
                    cpu["sample"] = (f"reference CLI {best.split('_')[0]} -threads={runs[best]['threads']}, same {len(res['core_blk'])}-edge core (phase 1, "
                                     f"untimed) + the first batch of {len(res['upd'][0])} updates (phase 2), mean of {runs[best]['repetitions']} runs")
            except Exception as e:
                cpu["multi_thread_error"] = str(e)

    # ================================================ secondary legs ===============================================
    es = res["es"]
    if rank == 0 and not args.no_secondary:
        try:
            eng = es[0]
            ms, tot = eng.bench_scan_all()
            if args.markers:
                eng.set_option("marker", 6)
            ms, tot = eng.bench_scan_all()
            if args.markers:
                eng.set_option("marker", 7)
            stt = eng.stats()
            scan_bytes = 12.0 * stt["N"] + 12.0 * stt["n"] + 4.0 * tot
            extra["neighbour_scan"] = {"edges_per_s": tot / (ms * 1e-3), "ms": ms, "edges": int(tot),
                                       "alg_GBps": scan_bytes / (ms * 1e-3) / 1e9,
                                       "frac_of_peak": scan_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                       "partition": 0 if not single else None}
            for mk, (label, w) in enumerate((("window_rebalance", int(stt["N"])), ("window_rebalance_half", int(stt["N"]) // 2))):
                if args.markers:
                    eng.bench_rebalance(w, 1)  # (sizes the scratch array outside the marked section)
                    eng.set_option("marker", 2 + 2 * mk)
                rms = eng.bench_rebalance(w, 5)
                if args.markers:
                    eng.set_option("marker", 3 + 2 * mk)
                extra[label] = {"window_slots": w, "ms_per_call": rms, "alg_GBps": 24.0 * w / (rms * 1e-3) / 1e9,
                                "frac_of_peak": 24.0 * w / (rms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                "note": "device time (HIP events) of tile sums + position table + fused scatter/fill"}
            # the other half-array window, [N/2, N): it lies in ONE binade of the position chain (a window that starts at slot 0
            # crosses one per doubling, and one thread builds that table: ~9 us of the launch), and the quarter window in place
            for label, w, upper in (("window_rebalance_half_upper", int(stt["N"]) // 2, 1), ("window_rebalance_quarter", int(stt["N"]) // 4, 1)):
                eng.set_option("rb_bench_upper", upper)
                eng.bench_rebalance(w, 1)
                rms = eng.bench_rebalance(w, 5)
                eng.set_option("rb_bench_upper", 0)
                extra[label] = {"window_slots": w, "window_start": int(stt["N"]) - w, "ms_per_call": rms, "alg_GBps": 24.0 * w / (rms * 1e-3) / 1e9,
                                "frac_of_peak": 24.0 * w / (rms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                "note": "in place (tile sums + scan / position table / tile order + ticketed in-place pass)"}
            if single:
                # graph-algorithm consumers on the device: BFS from vertex 0, one PageRank push
                nv = int(stt["n"])
                lv, bms = eng.bfs(0, with_ms=True)
                lv, bms = eng.bfs(0, with_ms=True)
                reached = int((lv != 0xFFFFFFFF).sum())
                pr, pms = eng.pagerank(np.ones(nv, np.float32), with_ms=True)
                pr, pms = eng.pagerank(np.ones(nv, np.float32), with_ms=True)
                extra["consumers"] = {"bfs_ms": bms, "bfs_levels": int(lv[lv != 0xFFFFFFFF].max()), "bfs_reached": reached,
                                      "bfs_edges_per_s": tot / (bms * 1e-3), "pagerank_ms": pms,
                                      "pagerank_edges_per_s": tot / (pms * 1e-3),
                                      "note": "device time; pagerank = bulk scan + stable radix sort by dest (rocPRIM, a library op) + "
                                              "in-order segment sums (bit-identical to the reference's fp32 loop; the adds into one "
                                              "destination are sequential by definition)"}
                # non-parity bulk build of the same core graph on a fresh engine
                eb = pkg.PCSR(n_global, device=dev_id)
                tb0 = time.perf_counter()
                bb_ms = eb.bulk_build(res["core_blk"], with_ms=True)
                tb1 = time.perf_counter()
                extra["bulk_build"] = {"edges": int(len(res["core_blk"])), "device_ms": bb_ms, "wall_ms_incl_h2d": (tb1 - tb0) * 1e3,
                                       "edges_per_s_device": len(res["core_blk"]) / (bb_ms * 1e-3), "N_slots": int(eb.geometry()[0]),
                                       "note": "NOT layout-identical to the one-by-one build (history dependent); same edge "
                                               "set, values, num_neighbors and invariants"}
                # double_list / half_list alone (PCSR.cpp:251-320) on that engine: N -> 2N -> N, three times; algorithmic bytes per
                # 12 (N/2) + 12 N = 18 N' for a doubling INTO N' slots, 12 (2N) + 12 N = 36 N' for a halving into N'
                n0 = int(eb.geometry()[0])
                dms, hms = eb.bench_resize(3)
                extra["double_list"] = {"N_slots_from": n0, "N_slots_to": 2 * n0, "ms_per_call": dms, "alg_bytes": 18.0 * 2 * n0,
                                        "alg_GBps": 18.0 * 2 * n0 / (dms * 1e-3) / 1e9, "frac_of_peak": 18.0 * 2 * n0 / (dms * 1e-3) / 1e9 / HBM_PEAK_GBS}
                extra["half_list"] = {"N_slots_from": 2 * n0, "N_slots_to": n0, "ms_per_call": hms, "alg_bytes": 36.0 * n0,
                                      "alg_GBps": 36.0 * n0 / (hms * 1e-3) / 1e9, "frac_of_peak": 36.0 * n0 / (hms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                      "note": "device time of tile sums + position table + ONE fused pass from the old array into the fresh one "
                                              "(the reference compacts, reallocates and redistributes: 36 N bytes; this pass moves 12 (2N) + 12 N)"}
                eb.close()
        except Exception as e:  # never let a secondary measurement kill the headline
            extra["secondary_error"] = str(e)
    # release the headline engines before further workloads are built
    headline = {k: res[k] for k in ("value", "ms_per_step", "engine", "N_slots", "logN")}
    end_to_end = res.get("end_to_end")
    roofline = res.get("roofline")
    for e in es:
        e.close() if single else None
    if res["pp"] is not None:
        res["pp"].close()
    res = None

    def side_leg(key, wl_, steps, warmup, check, after_core=None, more=None):
        try:
            r = run_workload(wl_, key, steps, warmup, False, after_core=after_core)
            out = {"workload": wl_.name(P, N), "value": r["value"], "ms_per_step": r["ms_per_step"], "steps": steps, "engine": r["engine"]}
            if r.get("end_to_end"):
                out["end_to_end"] = r["end_to_end"]
            if more:
                out.update(more)
            if check and not args.no_check:
                ok, against, osecs, nops = parity_check(wl_, r)
                out["parity_checked"] = ok
                if not ok:
                    raise SystemExit(f"PARITY FAILURE in {key}")
            for e in r["es"]:
                e.close() if single else None
            if r["pp"] is not None:
                r["pp"].close()
            extra[key] = out
        except SystemExit:
            raise
        except Exception as e:
            extra[key + "_error"] = str(e)

    if not args.no_secondary and cfg == 4 and not args.zipf:
        # the same graph with raw labels (the faithful input: partition 0 holds 44 % of the edges), and config #5's stream
        if permute:
            side_leg("raw_labels", Workload(streams, 4, n_global, scale, core_edges, batch, False), max(1, min(args.steps, 2)), 1, False)
            if N == 1:
                # the same raw-label graph after pppcsr_repartition to balanced vertex ranges (no reference
                # behaviour to compare with — the repartitioning tests hold the rule): what balancing buys on the skewed input
                info = {}

                def rebalance_parts(pp_, es_):
                    before = [int(e.stats()["N"]) for e in es_]
                    st_new = pp_.balanced_starts()
                    t_r = time.perf_counter()
                    pp_.repartition(st_new)
                    t_r = time.perf_counter() - t_r
                    es2 = [pp_.partition(q) for q in range(P)]
                    apply_options(es2)
                    info.update({"starts": [int(x) for x in st_new], "repartition_s": t_r, "N_slots_before": before,
                                 "N_slots_after": [int(e.stats()["N"]) for e in es2],
                                 "note": "raw labels, vertex ranges moved to equal (num_neighbors + 1) weight by pppcsr_repartition after the core "
                                         "load; compare with raw_labels (uniform ranges)"})
                    return es2

                side_leg("raw_labels_repartitioned", Workload(streams, 4, n_global, scale, core_edges, batch, False), max(1, min(args.steps, 2)), 1,
                         False, after_core=rebalance_parts, more=info)
        side_leg("config5_zipf", Workload(streams, 5, n_global, scale, core_edges, batch, permute), 1, 1, True)
    if not args.no_secondary and cfg == 2 and N == 1 and not args.zipf:
        # config #3 and the config #5 stream shape on this one-GPU graph
        side_leg("config3_mixed", Workload(streams, 3, n_global, scale, core_edges, batch, permute), max(1, min(args.steps, 3)), 1, True)
        side_leg("config5_shape_zipf", Workload(streams, 5, n_global, scale, core_edges, batch, permute), max(1, min(args.steps, 2)), 1, True)

    if rank == 0:
        out = {
            "metric": "edge-updates/sec", "value": headline["value"], "unit": "edge-updates/s", "n_gpus": N, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": headline["ms_per_step"], "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": wl.name(P, N), "vertices": n_global, "core_edges": core_edges, "updates_per_step": batch,
                       "parallelism": f"{P} partition(s), {ppr} per GPU x {N} GPU(s)"
                                      + ("" if N == 1 else (", native RCCL send/recv (pppcsr_exchange_apply)" if carrier["use_native"] else
                                                             f", {'RCCL' if args.backend == 'nccl' else args.backend} all-to-all (torch.distributed)")),
                       "N_slots": headline["N_slots"], "logN": headline["logN"],
                       "distinct_update_batches": max(1, min(args.distinct_batches, args.warmup + args.steps)),
                       "semantics": "sequential stream order (bit-exact vs reference -threads=1)",
                       **({"exchange_note": carrier["note"]} if carrier["note"] else {})},
            "roofline": roofline, "cpu_baseline": cpu, "engine": headline["engine"], "end_to_end": end_to_end, **extra,
        }
        print(json.dumps(out), flush=True)
    if N > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
