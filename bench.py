#!/usr/bin/env python3
"""bench.py — edge-updates/s of the MI355X PMA engine on BASELINE.json's workloads.

One "step" = one pass of the hot path over one batch of synthetic updates already resident in HBM:
  N = 1 : config #2 — RMAT scale-20 core graph (10 M edges, a/b/c = .57/.19/.19), single partition,
          each step applies a fresh batch of 1 M RMAT inserts (or --mixed: config #3, 50/50 insert +
          delete of existing core edges) in stream order.
  N > 1 : configs #4/#5 shape, weak scaling — N * 2^20 vertices, N * 10 M core edges, N * 1 M updates per
          step; every rank owns one vertex-range partition (PPPCSR.cpp:13-34 rule), holds a contiguous
          block of the global stream, buckets it by owner (stable) and exchanges buckets with one RCCL
          all-to-all; the receiver concatenates in source-rank order == global stream order.
Launch for N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N
Prints ONE JSON line on rank 0 (see DESIGN.md §6 for the field definitions).
"""
import argparse
import ctypes
import importlib.util
import json
import os
import sys
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


def log(rank, *a):
    if rank == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def gen_block(streams, kind, scale, count, seed, offset, n_global, permute, core=None, mixed_seed=0):
    """counter-based block [offset, offset+count) of the global stream `seed`"""
    if kind == "zipf":  # config #5: hot-vertex stream, src = Zipf(1.2) rank, dst uniform, all ADD
        s = streams.zipf_sources(n_global, count, seed=seed, alpha=1.2, offset=offset)
        d = streams.uniform_ints(seed + 7, count, n_global, offset=offset)
    else:
        s, d = streams.rmat_edges(scale, count, seed=seed, offset=offset)
    if permute:
        s = streams.permute_labels(s, n_global)
        d = streams.permute_labels(d, n_global)
    ops = streams.adds(s, d)
    if kind == "mixed":
        half = count // 2
        ops = streams.mixed_existing_stream(core, ops[:half], seed=mixed_seed)
    return ops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scale", type=int, default=20, help="RMAT scale per GPU")
    ap.add_argument("--core-edges", type=int, default=10_000_000, help="core edges per GPU")
    ap.add_argument("--batch", type=int, default=1_000_000, help="updates per GPU per step")
    ap.add_argument("--mixed", action="store_true", help="config #3: alternate insert / delete-existing")
    ap.add_argument("--zipf", action="store_true", help="config #5: updates with Zipf(1.2) sources (hot-vertex rebalance cascades)")
    ap.add_argument("--labels", choices=["permuted", "raw"], default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ref-cli", action="store_true", help="skip the multi-threaded run of the reference's own CLI binary")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket round kernels with HIP events")
    ap.add_argument("--mode", type=int, default=-1, help="0 strict prefix rounds, 1 speculative rounds (engine default)")
    ap.add_argument("--opt-horizon", type=int, default=0)
    ap.add_argument("--epoch-ops", type=int, default=0)
    ap.add_argument("--region-slots", type=int, default=0)
    ap.add_argument("--max-horizon", type=int, default=0)
    ap.add_argument("--rounds-per-sync", type=int, default=0)
    ap.add_argument("--check", action="store_true", help="verify the final state against the oracle (slow)")
    ap.add_argument("--backend", default="nccl", help="process-group backend; 'gloo' only for functional tests of the N > 1 path on one GPU")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    P = args.gpus
    if world != P:
        raise SystemExit(f"--gpus {P} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {P}")
    if not torch.cuda.is_available():
        raise SystemExit("no GPU: the engine is HIP-only (no CPU fallback)")
    ndev = torch.cuda.device_count()
    dev_id = local_rank % max(ndev, 1)  # (one rank per GPU in real runs; ranks share a GPU only in the gloo functional test)
    torch.cuda.set_device(dev_id)
    dev = torch.device("cuda", dev_id)
    if P > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend, rank=rank, world_size=world)

    pkg = _load("ppcsr_amd", os.path.join(ROOT, "parallel-packed-csr_amd", "__init__.py"), pkg=True)
    streams = _load("ppcsr_streams", os.path.join(ROOT, "parallel-packed-csr_amd", "streams.py"))
    exch = _load("ppcsr_exchange", os.path.join(ROOT, "parallel-packed-csr_amd", "exchange.py"))
    pkg.load_library()  # in-tree HIP build; raises if missing

    gscale = args.scale + int(np.log2(P))
    assert (1 << (gscale - args.scale)) == P, "--gpus must be a power of two"
    n_global = 1 << gscale
    permute = (args.labels or ("permuted" if P > 1 else "raw")) == "permuted"
    starts, sizes = exch.partition_layout(n_global, P)
    my_n = int(sizes[rank])
    kind = "mixed" if args.mixed else ("zipf" if args.zipf else "insert")

    t0 = time.time()
    core_blk = gen_block(streams, "insert", gscale, args.core_edges, 1, rank * args.core_edges, n_global, permute)
    nsteps = args.warmup + args.steps + (0 if args.no_profile else 0)
    upd = []
    for k in range(nsteps):
        upd.append(gen_block(streams, kind, gscale, args.batch, 2 + 10 * k, rank * args.batch, n_global, permute,
                             core=core_blk, mixed_seed=3 + 10 * k))
    log(rank, f"generated core {len(core_blk)} + {nsteps} x {args.batch} updates per rank in {time.time() - t0:.1f}s "
              f"(n_global={n_global}, labels={'permuted' if permute else 'raw'})")

    eng = pkg.PCSR(my_n, device=dev_id)
    if args.mode >= 0:
        eng.set_option("mode", args.mode)
    if args.opt_horizon:
        eng.set_option("opt_horizon", args.opt_horizon)
    if args.epoch_ops:
        eng.set_option("epoch_ops", args.epoch_ops)
    if args.region_slots:
        eng.set_option("region_slots", args.region_slots)
    if args.max_horizon:
        eng.set_option("max_horizon", args.max_horizon)
    if args.rounds_per_sync:
        eng.set_option("rounds_per_sync", args.rounds_per_sync)

    xdev = dev if args.backend == "nccl" else torch.device("cpu")  # where the exchange runs

    def to_dev(a):
        return torch.from_numpy(a.view(np.int32)).to(xdev if P > 1 else dev)

    keep_alive = []  # device tensors handed to the engine must outlive the (asynchronous) apply

    def run_step(ops_dev):
        """bucket by owner + all-to-all (N > 1), then apply in stream order on this rank's partition"""
        if P > 1:
            mine = exch.exchange_ops(ops_dev, n_global, P, dist.group.WORLD).to(dev)
            # the exchange ran on torch's stream; the engine applies on its own HIP stream
            torch.cuda.current_stream().synchronize()
            keep_alive.append(mine)
        else:
            mine = ops_dev
        if mine.shape[0]:
            eng.apply_device(mine.data_ptr(), mine.shape[0])
        return mine.shape[0]

    # ---- core load (untimed) ----
    t0 = time.time()
    core_dev = to_dev(core_blk)
    run_step(core_dev)
    torch.cuda.synchronize()
    del core_dev
    st = eng.stats()
    log(rank, f"core loaded in {time.time() - t0:.1f}s: N={st['N']} logN={st['logN']} rounds={st['rounds']} "
              f"exclusive={st['exclusive_ops']} doubles={st['double_calls']} rollbacks={st['rollbacks']}")

    # every step starts from the SAME core graph (config #2/#3 exactly): the device-to-device restore of the
    # core snapshot is part of the step and inside the timed region (2 x 12 B/slot of HBM traffic, ~0.1 ms)
    eng.snapshot()
    upd_dev = [to_dev(u) for u in upd]
    for k in range(args.warmup):
        eng.restore()
        run_step(upd_dev[k])
    torch.cuda.synchronize()

    s0 = eng.stats()
    if P > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    applied = 0
    for k in range(args.warmup, args.warmup + args.steps):
        eng.restore()
        applied += run_step(upd_dev[k])
    torch.cuda.synchronize()
    if P > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    s1 = eng.stats()
    if P > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=xdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    total_updates = args.batch * P * args.steps
    value = total_updates / elapsed
    dstat = {k: s1[k] - s0[k] for k in ("rounds", "committed", "planned", "exclusive_ops", "rollbacks", "round_syncs",
                                         "redistribute_slots", "redistribute_calls", "ops_applied", "double_calls")}
    dstat["updates_per_round"] = dstat["committed"] / max(dstat["rounds"], 1)
    dstat["device_ms_last_batch"] = s1["last_batch_ms"]

    # ---- roofline of the dominant round kernel ------------------------------------------------------------------
    # HIP events recorded on the engine's own stream around every round kernel.  Recording ~5 events per round costs
    # 25-35 % of throughput, so `value` above comes from the un-instrumented timed region and the SAME K steps are
    # replayed here with the events on (same inputs, same state: every step restarts from the core snapshot).
    roofline = None
    if not args.no_profile:
        eng.set_option("profile", 1)
        p0 = eng.stats()
        for k in range(args.warmup, args.warmup + args.steps):
            eng.restore()
            run_step(upd_dev[k])
        torch.cuda.synchronize()
        p1 = eng.stats()
        eng.set_option("profile", 0)
        launches = p1["prof_launches"]
        kern = {"plan": p1["prof_plan_ms"], "check": p1["prof_check_ms"], "apply": p1["prof_apply_ms"],
                "compact": p1["prof_compact_ms"]}
        spec = args.mode != 0  # engine default: speculative rounds (o_* kernels; the compaction is folded into o_apply)
        names = {"plan": "o_plan" if spec else "k_plan", "check": "o_check" if spec else "k_check",
                 "apply": "o_apply" if spec else "k_apply", "compact": "o_compact"}
        dom = max(kern, key=kern.get)
        d = {k: p1[k] - p0[k] for k in ("redistribute_slots", "ops_applied", "committed", "rounds", "planned")}
        # algorithmic bytes: 12 B op record + 24 B per slot of every redistribute() the reference makes
        alg_bytes = 12.0 * d["ops_applied"] + 24.0 * d["redistribute_slots"]
        avg_ms = kern[dom] / max(launches, 1)
        achieved = (alg_bytes / max(launches, 1)) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(tpath):  # HBM bytes per launch from rocprofv3 --pmc passes of this same command (offline)
            try:
                traffic = json.load(open(tpath)).get(names[dom], {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": names[dom], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                    "launches": int(launches), "avg_launch_us": avg_ms * 1e3,
                    "alg_bytes_per_launch": alg_bytes / max(launches, 1),
                    "alg_bytes_per_update": alg_bytes / max(d["ops_applied"], 1),
                    "kernel_ms": {names[k]: round(v, 3) for k, v in kern.items() if v > 0},
                    "measured_on": "profiled replay of the timed steps (HIP events on the engine stream)"}

    # ---- CPU baseline beside it (rank 0, N == 1): the reference (oracle/_ref) or the oracle port, 1 thread ----
    cpu = None
    extra = {}
    if rank == 0 and P == 1 and not args.no_cpu_baseline:
        from oracle_lib import Oracle, RefPCSR, have_ref
        kindc = "reference" if have_ref() else "port"
        Cls = RefPCSR if have_ref() else Oracle
        c = Cls(my_n)
        tl = time.time()
        c.apply(core_blk)
        tl = time.time() - tl
        tc = time.time()
        c.apply(upd[args.warmup])
        tc = time.time() - tc
        cpu = {"value": len(upd[args.warmup]) / tc, "unit": "edge-updates/s", "cores": 1, "kind": kindc,
               "sample": f"same {args.core_edges}-edge core ({tl:.1f}s load, untimed) + the first timed batch of "
                         f"{len(upd[args.warmup])} updates in stream order on one host thread ({tc:.2f}s)"}
        if args.check:
            c2 = Oracle(my_n)
            c2.apply(core_blk)
            c2.apply(upd[args.warmup + args.steps - 1])  # the engine holds core + the last timed batch
            ei, en = eng.state()
            oi, on = c2.state()
            assert eng.geometry() == c2.geometry() and np.array_equal(ei, oi) and np.array_equal(en, on), "PARITY FAILURE"
            extra["parity_checked"] = True
        c.close()
        # the reference's own binary with its thread pools (the north_star's "-pppcsrnuma CPU path on the same box"):
        # text edge lists in /tmp, phase-2 time = the SECOND "Elapsed wall clock time" line (reference bench protocol)
        ref_cli = os.path.join(ROOT, "oracle", "_ref", "ref_cli")
        if os.path.exists(ref_cli) and not args.no_ref_cli and not args.mixed:
            try:
                import subprocess
                import pandas as pd
                cores = os.cpu_count() or 1
                try:
                    cores = len(os.sched_getaffinity(0))
                except Exception:
                    pass
                cf, uf = "/tmp/ppcsr_bench_core.txt", "/tmp/ppcsr_bench_upd.txt"
                pd.DataFrame(core_blk[:, :2]).to_csv(cf, sep=" ", header=False, index=False)
                pd.DataFrame(upd[args.warmup][:, :2]).to_csv(uf, sep=" ", header=False, index=False)
                runs = {}
                share = min(cores, 16)  # the box's CPU share for one GPU
                for label, thr, flags in (("ppcsr_t8", 8, ["-ppcsr"]), (f"ppcsr_t{share}", share, ["-ppcsr"]),
                                          (f"pppcsrnuma_t{share}", share, ["-pppcsrnuma", "-partitions_per_domain=8"])):
                    r = subprocess.run([ref_cli, f"-threads={thr}", f"-size={args.batch}", "-insert"] + flags +
                                       [f"-core_graph={cf}", f"-update_file={uf}"], capture_output=True, text=True, timeout=600)
                    el = [int(l.split(":")[1]) for l in r.stdout.splitlines() if l.startswith("Elapsed wall clock time")]
                    if len(el) >= 2 and el[1] > 0:
                        runs[label] = {"updates_per_s": args.batch / (el[1] * 1e-3), "ms": el[1], "core_load_ms": el[0], "threads": thr}
                    else:
                        runs[label] = {"failed": r.returncode, "stderr": r.stderr[-200:], "stdout_tail": r.stdout[-200:]}
                extra["cpu_reference_cli"] = {"cores_visible": cores, "runs": runs,
                                              "note": "unmodified reference binary (oracle/_ref/ref_cli); multi-threaded runs are "
                                                      "not deterministic in layout (SURVEY.md §8c)"}
                os.remove(cf)
                os.remove(uf)
            except Exception as e:
                extra["cpu_reference_cli_error"] = str(e)

    if P > 1 and args.check:
        # tier-A parity per partition: this rank's partition must equal the oracle fed with the partition's subsequence
        # of the GLOBAL stream (block r of every batch is regenerated from its counters)
        from oracle_lib import Oracle
        o = Oracle(my_n)

        def mine_of(kind_, count, seed, mixed_seed=0, corefor=None):
            parts = []
            for r in range(P):
                cb = None
                if kind_ == "mixed":
                    cb = gen_block(streams, "insert", gscale, args.core_edges, 1, r * args.core_edges, n_global, permute)
                parts.append(gen_block(streams, kind_, gscale, count, seed, r * count, n_global, permute, core=cb, mixed_seed=mixed_seed))
            g = np.concatenate(parts)
            ps = n_global // P
            own = np.minimum(g[:, 0].astype(np.int64) // ps, P - 1)
            sub = g[own == rank].copy()
            sub[:, 0] -= np.uint32(starts[rank])
            return sub
        o.apply(mine_of("insert", args.core_edges, 1))
        k = args.warmup + args.steps - 1
        o.apply(mine_of(kind, args.batch, 2 + 10 * k, mixed_seed=3 + 10 * k))
        ei, en = eng.state()
        oi, on = o.state()
        okp = eng.geometry() == o.geometry() and np.array_equal(ei, oi) and np.array_equal(en, on)
        t = torch.tensor([1 if okp else 0], dtype=torch.int64, device=xdev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        extra["parity_checked_all_partitions"] = bool(t.item())
        assert t.item() == 1, "PARITY FAILURE on some partition"
    # ---- secondary kernels: bulk neighbour scan + whole-window rebalance (HBM-roofline kernels) ----
    if rank == 0:
        try:
            ms, tot = eng.bench_scan_all()
            ms, tot = eng.bench_scan_all()
            stt = eng.stats()
            scan_bytes = 12.0 * stt["N"] + 12.0 * stt["n"] + 4.0 * tot
            extra["neighbour_scan"] = {"edges_per_s": tot / (ms * 1e-3), "ms": ms, "edges": int(tot),
                                       "alg_GBps": scan_bytes / (ms * 1e-3) / 1e9,
                                       "frac_of_peak": scan_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
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
                                  "note": "device time; pagerank = bulk scan + stable radix sort by dest + in-order "
                                          "segment sums (bit-identical to the reference's fp32 loop)"}
            # non-parity bulk build of the same core graph on a fresh engine beside the parity load above
            if P == 1:
                eb = pkg.PCSR(my_n, device=dev_id)
                tb0 = time.perf_counter()
                bb_ms = eb.bulk_build(core_blk, with_ms=True)
                tb1 = time.perf_counter()
                extra["bulk_build"] = {"edges": int(len(core_blk)), "device_ms": bb_ms, "wall_ms_incl_h2d": (tb1 - tb0) * 1e3,
                                       "edges_per_s_device": len(core_blk) / (bb_ms * 1e-3), "N_slots": int(eb.geometry()[0]),
                                       "note": "NOT layout-identical to the one-by-one build (history dependent); same edge "
                                               "set, values, num_neighbors and invariants"}
                eb.close()
            for label, w in (("window_rebalance", int(stt["N"])), ("window_rebalance_half", int(stt["N"]) // 2)):
                rms = eng.bench_rebalance(w, 5)
                extra[label] = {"window_slots": w, "ms_per_call": rms, "alg_GBps": 24.0 * w / (rms * 1e-3) / 1e9,
                                "frac_of_peak": 24.0 * w / (rms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                "note": "device time (HIP events) of rank scan + position table + fused scatter/fill"
                                        + ("" if w == int(stt["N"]) else " + copy-back")}
        except Exception as e:  # never let a secondary measurement kill the headline
            extra["secondary_error"] = str(e)

    if rank == 0:
        out = {
            "metric": "edge-updates/sec", "value": value, "unit": "edge-updates/s", "n_gpus": P, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": (f"config#3 {args.batch} mixed 50/50 insert+delete" if args.mixed else
                                    (f"config#5 {args.batch} Zipf(1.2)-source inserts" if args.zipf else f"config#2 {args.batch} random inserts"))
                       + f" on RMAT scale-{args.scale} / {args.core_edges}-edge core per GPU"
                       + (f", {P} vertex-range partitions, labels {'permuted' if permute else 'raw'}, {'RCCL' if args.backend == 'nccl' else args.backend} all-to-all" if P > 1 else ", 1 partition"),
                       "vertices": n_global, "core_edges": args.core_edges * P, "updates_per_step": args.batch * P,
                       "parallelism": f"partition-per-gpu x{P}", "N_slots": int(s1["N"]), "logN": int(s1["logN"]),
                       "semantics": "sequential stream order (bit-exact vs reference -threads=1)"},
            "roofline": roofline, "cpu_baseline": cpu, "engine": dstat, **extra,
        }
        print(json.dumps(out), flush=True)
    if P > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
