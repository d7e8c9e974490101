#!/usr/bin/env python3
"""Cut marked sections out of a rocprofv3 kernel trace and two --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py --markers`
and write per-kernel {launches, avg_us, fetch_bytes (x2, gfx950), write_bytes} for each section, plus the roofline figures that
follow from them (algorithmic bytes / average duration / 8 TB/s, HBM traffic / algorithmic bytes).
usage: roofline_summary.py <kernel_trace.csv> <fetch counter_collection.csv> <write counter_collection.csv> <bench.json> <out.json> [command]
Units per /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE count KiB; on gfx950
FETCH_SIZE reports half of the bytes of a wide coalesced read (doubled here; our 12-B-per-lane accesses are narrower than the
calibrated 16-B case, so absolutes are approximate, ratios exact).  Counters and traces come from separate runs."""
import csv
import json
import os
import subprocess
import sys
from collections import OrderedDict, defaultdict

SECTIONS = OrderedDict([("timed_rounds", (0, 1)), ("full_rebalance", (2, 3)), ("half_rebalance", (4, 5)), ("scan_all", (6, 7))])
HBM_PEAK = 8000.0  # GB/s


def short(name):
    n = name.split("(")[0]
    return n.replace("ppcsr::", "").replace("void ", "").strip()


def read_rows(path, want_counter=None):
    rows = []
    for r in csv.DictReader(open(path)):
        if want_counter and r.get("Counter_Name") != want_counter:
            continue
        rows.append(r)
    key = "Start_Timestamp" if rows and "Start_Timestamp" in rows[0] and not want_counter else "Dispatch_Id"
    rows.sort(key=lambda r: int(r[key]))
    return rows


def cut(rows):
    """{section: [rows between its two markers]} in dispatch order"""
    out = {}
    for sec, (a, b) in SECTIONS.items():
        ia = [i for i, r in enumerate(rows) if f"k_mark_{a}" in r["Kernel_Name"]]
        ib = [i for i, r in enumerate(rows) if f"k_mark_{b}" in r["Kernel_Name"]]
        if not ia or not ib:
            continue
        lo, hi = ia[0], [i for i in ib if i > ia[0]][0]
        out[sec] = [r for r in rows[lo + 1:hi] if "k_mark_" not in r["Kernel_Name"]]
    return out


def main():
    trace, fetch, write, bench, outp = sys.argv[1:6]
    cmd = sys.argv[6] if len(sys.argv) > 6 else None
    tsec = cut(read_rows(trace))
    fsec = cut(read_rows(fetch, "FETCH_SIZE"))
    wsec = cut(read_rows(write, "WRITE_SIZE"))
    res = OrderedDict()
    for sec in SECTIONS:
        per = OrderedDict()
        dur, fb, wb = defaultdict(list), defaultdict(list), defaultdict(list)
        for r in tsec.get(sec, []):
            dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for r in fsec.get(sec, []):
            fb[short(r["Kernel_Name"])].append(2.0 * float(r["Counter_Value"]) * 1024.0)
        for r in wsec.get(sec, []):
            wb[short(r["Kernel_Name"])].append(float(r["Counter_Value"]) * 1024.0)
        for k in sorted(dur, key=lambda k: -sum(dur[k])):
            d = sorted(dur[k])
            per[k] = {"launches": len(d), "avg_us": sum(d) / len(d) / 1e3, "p50_us": d[len(d) // 2] / 1e3, "max_us": d[-1] / 1e3,
                      "total_ms": sum(d) / 1e6,
                      "fetch_bytes_per_launch": (sum(fb[k]) / len(fb[k])) if fb.get(k) else None,
                      "write_bytes_per_launch": (sum(wb[k]) / len(wb[k])) if wb.get(k) else None}
            if per[k]["fetch_bytes_per_launch"] is not None and per[k]["write_bytes_per_launch"] is not None:
                per[k]["hbm_bytes_per_launch"] = per[k]["fetch_bytes_per_launch"] + per[k]["write_bytes_per_launch"]
        rows = tsec.get(sec, [])
        span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e6 if rows else None
        res[sec] = {"kernels": per, "span_ms": span}
    bj = json.loads(open(bench).read().strip().splitlines()[-1])
    derived = OrderedDict()
    # (i) timed rounds: the dominant kernel against the algorithmic bytes of the updates it planned / applied
    tr = res["timed_rounds"]["kernels"]
    rk = [k for k in ("o_plan", "o_check", "o_check_w", "o_apply", "o_big", "o_settle", "o_compact") if k in tr]
    if rk:
        alg_per_update = bj["roofline"]["alg_bytes_per_update"] if bj.get("roofline") else None
        updates = bj["config"]["updates_per_step"] * bj["steps"]
        dom = max(rk, key=lambda k: tr[k]["total_ms"])
        launches = tr[dom]["launches"]
        alg_launch = alg_per_update * updates / launches if alg_per_update else None
        # per round: every kernel weighted by how often it ran per o_plan launch (o_settle: once per chunk; o_big: only while a stream queues windows)
        nplan = max(tr["o_plan"]["launches"], 1) if "o_plan" in tr else launches
        round_us = sum(tr[k]["total_ms"] * 1e3 for k in rk) / nplan
        hbm_round = sum((tr[k].get("hbm_bytes_per_launch") or 0.0) * tr[k]["launches"] for k in rk) / nplan
        derived["timed_rounds"] = {"dominant_kernel": dom, "launches": launches, "avg_us": tr[dom]["avg_us"], "alg_bytes_per_launch": alg_launch,
                                   "achieved_GBps": alg_launch / (tr[dom]["avg_us"] * 1e-6) / 1e9 if alg_launch else None,
                                   "frac_of_8TBps": alg_launch / (tr[dom]["avg_us"] * 1e-6) / 1e9 / HBM_PEAK if alg_launch else None,
                                   "round_us_sum_of_kernel_avgs": round_us, "hbm_bytes_per_round": hbm_round,
                                   "traffic_over_algorithmic": hbm_round / alg_launch if alg_launch else None,
                                   "updates": updates, "alg_bytes_per_update": alg_per_update,
                                   "updates_per_s_bench": bj["value"], "ms_per_step_bench": bj["ms_per_step"]}
    N = bj["config"]["N_slots"][0]
    for sec, w in (("full_rebalance", N), ("half_rebalance", N // 2)):
        ks = res[sec]["kernels"]
        main = "k_rb_scatter" if "k_rb_scatter" in ks else next((k for k in ks if k.startswith("k_rb_inplace")), None)
        if main:
            calls = ks[main]["launches"]
            us = sum(v["total_ms"] for v in ks.values()) * 1e3 / calls
            hbm = sum((v.get("hbm_bytes_per_launch") or 0.0) * v["launches"] for v in ks.values()) / calls
            alg = 24.0 * w
            derived[sec] = {"window_slots": w, "calls": calls, "kernel_us_per_call": us, "alg_bytes": alg, "achieved_GBps": alg / (us * 1e-6) / 1e9,
                            "frac_of_8TBps": alg / (us * 1e-6) / 1e9 / HBM_PEAK, "hbm_bytes_per_call": hbm, "traffic_over_algorithmic": hbm / alg,
                            "kernels": {k: round(v["avg_us"], 2) for k, v in ks.items()}}
    ks = res["scan_all"]["kernels"]
    if "k_scan_write" in ks and bj.get("neighbour_scan"):
        calls = ks["k_scan_write"]["launches"]
        us = sum(v["total_ms"] for v in ks.values()) * 1e3 / calls
        hbm = sum((v.get("hbm_bytes_per_launch") or 0.0) * v["launches"] for v in ks.values()) / calls
        E = bj["neighbour_scan"]["edges"]
        alg = 12.0 * N + 12.0 * bj["config"]["vertices"] + 4.0 * E
        derived["scan_all"] = {"edges": E, "calls": calls, "kernel_us_per_call": us, "alg_bytes": alg, "achieved_GBps": alg / (us * 1e-6) / 1e9,
                               "frac_of_8TBps": alg / (us * 1e-6) / 1e9 / HBM_PEAK, "hbm_bytes_per_call": hbm, "traffic_over_algorithmic": hbm / alg}
    commit = None
    try:
        commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=os.path.dirname(os.path.abspath(__file__))).stdout.strip() or None
    except Exception:
        pass
    out = OrderedDict([("command", cmd), ("commit", commit or os.environ.get("PPCSR_COMMIT")),
                       ("csrc_sha256", (bj.get("roofline") or {}).get("csrc_sha256")), ("workload", bj["config"]["workload"]),
                       ("units", "durations from the --kernel-trace pass; FETCH_SIZE (x2, gfx950) and WRITE_SIZE from separate --pmc passes, KiB -> bytes"),
                       ("derived", derived), ("sections", res)])
    # bench.py reads timed_rounds[kernel].hbm_bytes_per_launch
    out["timed_rounds"] = res["timed_rounds"]["kernels"]
    json.dump(out, open(outp, "w"), indent=1)
    print(json.dumps(derived, indent=1))


if __name__ == "__main__":
    main()
