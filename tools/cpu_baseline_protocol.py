#!/usr/bin/env python3
"""The reference's own benchmark protocol on this box's host cores (src/benchmarking/benchmark-strong-scaling.sh:105-156):
the unmodified CLI binary (oracle/_ref/ref_cli), `-pppcsrnuma -partitions_per_domain=8` and `-ppcsr`, a thread sweep, REPS
repetitions each, mean and sample standard deviation of the SECOND `Elapsed wall clock time` line (the update phase; phase 1
loads the core graph).  Workload: exactly bench.py's stream for the chosen config (same generator, same seeds, same labels) —
config #2 (RMAT scale-20 core of 10 M edges, 1 M inserts), #4 (10 M vertices / 100 M edges, 10 M inserts) or #5 (#4's graph,
10 M Zipf(1.2)-source inserts).  Writes a JSON summary after every run (a call cut short keeps what it measured).
usage: python tools/cpu_baseline_protocol.py [out.json] [reps] [threads,comma,separated] [config] [modes,comma,separated] [labels]"""
import json
import os
import subprocess
import sys
import time

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import load_streams  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r03_cpu_baseline.json")
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
try:
    cores = len(os.sched_getaffinity(0))
except Exception:
    cores = os.cpu_count() or 1
threads = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [t for t in (1, 2, 4, 8, 16, 32, 64) if t <= max(cores, 16)]
cli = os.path.join(ROOT, "oracle", "_ref", "ref_cli")
if not os.path.exists(cli):
    raise SystemExit("oracle/_ref/ref_cli is not built (the reference tree is compiled in the build container)")
st = load_streams()
cfg = int(sys.argv[4]) if len(sys.argv) > 4 else 2
modes = sys.argv[5].split(",") if len(sys.argv) > 5 else ["pppcsrnuma", "ppcsr"]
labels = sys.argv[6] if len(sys.argv) > 6 else ("raw" if cfg == 2 else "permuted")  # bench.py's defaults
size_cap = int(sys.argv[7]) if len(sys.argv) > 7 else 0  # only the first `size_cap` updates of the batch (-size=): a bounded sample
run_timeout = int(sys.argv[8]) if len(sys.argv) > 8 else 900
import threading  # noqa: E402


def _heartbeat():  # (the GPU box's watchdog kills a command that prints nothing for 7 minutes; one reference run can take longer)
    t0 = time.time()
    while True:
        time.sleep(60)
        print(f"  ... running ({time.time() - t0:.0f}s)", flush=True)


threading.Thread(target=_heartbeat, daemon=True).start()
import importlib.util  # noqa: E402
spec = importlib.util.spec_from_file_location("ppcsr_bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
if cfg == 2:
    wl = bench.Workload(st, 2, 1 << 20, 20, 10_000_000, 1_000_000, labels == "permuted")
else:
    wl = bench.Workload(st, cfg, 10_000_000, 24, 100_000_000, 10_000_000, labels == "permuted")
t_gen = time.time()
print("generating the core ...", flush=True)  # (a silent stretch of minutes looks like a hang to the GPU box's watchdog)
core = wl.core(0, wl.core_edges)
print(f"core generated ({time.time() - t_gen:.0f}s); generating the updates ...", flush=True)
upd = wl.updates(0, 0, wl.batch, core_for_mixed=core)
n_upd = min(len(upd), size_cap) if size_cap else len(upd)
print(f"updates generated ({time.time() - t_gen:.0f}s); writing the text files ...", flush=True)
cf, uf = "/tmp/ppcsr_proto_core.txt", "/tmp/ppcsr_proto_upd.txt"
pd.DataFrame(core[:, :2]).to_csv(cf, sep=" ", header=False, index=False)
pd.DataFrame(upd[:, :2]).to_csv(uf, sep=" ", header=False, index=False)
print(f"workload written in {time.time() - t_gen:.0f}s: {wl.name(8 if cfg != 2 else 1, 1)}", flush=True)
del core, upd
cpu_model = ""
try:
    cpu_model = [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
except Exception:
    pass
numa = len([x for x in os.listdir("/sys/devices/system/node") if x.startswith("node")]) if os.path.isdir("/sys/devices/system/node") else None
res = {"workload": f"{wl.name(8 if cfg != 2 else 1, 1)}: core = phase 1 (untimed), the first update batch = phase 2 (timed)", "config": cfg, "labels": labels,
       "binary": "oracle/_ref/ref_cli (unmodified reference)",
       "cores_available_to_this_process": cores, "cpus_online": os.cpu_count(), "cpu_model": cpu_model, "numa_nodes": numa, "repetitions": reps, "updates_timed": n_upd, "runs": {}}
t00 = time.time()
for mode, flags in (("pppcsrnuma", ["-pppcsrnuma", "-partitions_per_domain=8"]), ("ppcsr", ["-ppcsr"])):
    if mode not in modes:
        continue
    for t in threads:
        vals, loads = [], []
        r_n = reps if t > 1 else min(reps, 3)  # (a one-thread run takes ~20 s)
        for _ in range(r_n):
            try:
                r = subprocess.run([cli, f"-threads={t}", f"-size={n_upd}", "-insert"] + flags + [f"-core_graph={cf}", f"-update_file={uf}"],
                                   capture_output=True, text=True, timeout=run_timeout)
                el = [int(l.split(":")[1]) for l in r.stdout.splitlines() if l.startswith("Elapsed wall clock time")]
            except subprocess.TimeoutExpired:
                res["runs"][f"{mode}_t{t}"] = {"mode": mode, "threads": t, "timed_out_after_s": run_timeout, "updates": n_upd,
                                              "note": "the reference did not finish load + update phase within the limit"}
                print(f"  {mode} threads={t}: no result within {run_timeout}s", flush=True)
                json.dump(res, open(out, "w"), indent=1)
                break
            if len(el) >= 2 and el[1] > 0:
                vals.append(n_upd / (el[1] * 1e-3))
                loads.append(el[0])
            print(f"  {mode} threads={t} run {len(vals)}: phase 1 {el[0] if el else '?'} ms, phase 2 {el[1] if len(el) > 1 else '?'} ms "
                  f"({time.time() - t00:.0f}s)", flush=True)
        if vals:
            res["runs"][f"{mode}_t{t}"] = {"mode": mode, "threads": t, "repetitions": len(vals), "updates_per_s_mean": float(np.mean(vals)),
                                          "updates_per_s_std": float(np.std(vals, ddof=1)) if len(vals) > 1 else 0.0,
                                          "phase2_ms": [round(n_upd * 1e3 / v) for v in vals], "phase1_ms_mean": float(np.mean(loads))}
            print(f"{mode} threads={t}: {np.mean(vals) / 1e6:.2f} +- {(np.std(vals, ddof=1) if len(vals) > 1 else 0) / 1e6:.2f} M updates/s ({len(vals)} runs, {time.time() - t00:.0f}s)", flush=True)
        json.dump(res, open(out, "w"), indent=1)
done = {k: v for k, v in res["runs"].items() if "updates_per_s_mean" in v}
best = max(done, key=lambda k: done[k]["updates_per_s_mean"]) if done else None
res["best"] = best
json.dump(res, open(out, "w"), indent=1)
os.remove(cf)
os.remove(uf)
print("best:", best, res["runs"][best] if best else None)
