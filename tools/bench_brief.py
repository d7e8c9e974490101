#!/usr/bin/env python3
"""One line per leg of a bench.py JSON: value, ms, rounds, and the per-round kernel times of the instrumented replay."""
import json
import sys

for f in sys.argv[1:]:
    j = json.loads(open(f).read().strip().split("\n")[-1])
    r = j.get("roofline", {})
    km, n = r.get("kernel_ms", {}), max(r.get("launches", 1), 1)
    print(f, "value", round(j["value"] / 1e6, 1), "ms", round(j["ms_per_step"], 3), {a: round(b / n * 1e3, 1) for a, b in km.items()},
          "rounds", j.get("engine", {}).get("rounds"), "parity", j.get("parity_checked"))
    for k in ("config3_mixed", "config5_shape_zipf", "end_to_end"):
        if k in j:
            e = j[k].get("engine", {})
            print("   ", k, round(j[k]["value"] / 1e6, 1), "ms", round(j[k]["ms_per_step"], 3), "rounds", e.get("rounds"), "syncs", e.get("round_syncs"), "rollbacks", e.get("rollbacks"))
    for k in ("window_rebalance", "window_rebalance_half", "window_rebalance_half_upper", "window_rebalance_quarter", "double_list", "half_list", "neighbour_scan"):
        if k in j:
            print("   ", k, round(j[k].get("frac_of_peak", 0) * 100, 1), "%", round(j[k].get("ms_per_call", j[k].get("ms", 0)) * 1e3, 1), "us")
