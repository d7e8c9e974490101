#!/usr/bin/env python3
"""Per-kernel durations of the LAST `rounds` speculative rounds in a rocprofv3 kernel trace (rocpd sqlite output), i.e.
the timed region of `bench.py --no-profile` (the core load and warm-up come first).  Also prints the gaps between
consecutive kernels of those rounds.  usage: trace_summary.py <results.db | kernel_trace.csv> <rounds> [out.json] [rounds_to_skip_at_the_end]"""
import json
import sqlite3
import sys

db, rounds = sys.argv[1], int(sys.argv[2])
if db.endswith(".csv"):  # rocprofv3 --output-format csv: <prefix>_kernel_trace.csv
    import csv
    rows = []
    for r in csv.DictReader(open(db)):
        nm = r["Kernel_Name"]
        if any(k in nm for k in ("o_plan", "o_check", "o_apply", "o_compact")):
            rows.append((nm, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    rows.sort(key=lambda x: x[1])
else:  # rocpd sqlite output (the default format)
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end from kernels where name like '%o_plan%' or name like '%o_check%' or "
                     "name like '%o_apply%' or name like '%o_compact%' order by start").fetchall()
plans = [i for i, r in enumerate(rows) if "o_plan" in r[0]]
skip = int(sys.argv[4]) if len(sys.argv) > 4 else 0  # rounds to leave out at the end (e.g. bench.py's profiled replay)
first = plans[-(rounds + skip)]
last = plans[-skip] if skip else len(rows)
sel = rows[first:last]
out = {}
for key in ("o_plan", "o_check", "o_apply", "o_compact"):
    d = [r[2] - r[1] for r in sel if key in r[0]]
    d.sort()
    out[key] = {"launches": len(d), "avg_us": sum(d) / len(d) / 1e3, "p50_us": d[len(d) // 2] / 1e3,
                "p95_us": d[int(len(d) * 0.95)] / 1e3, "max_us": d[-1] / 1e3}
gaps = [sel[i + 1][1] - sel[i][2] for i in range(len(sel) - 1)]
gaps.sort()
span = sel[-1][2] - sel[0][1]
out["gaps"] = {"avg_us": sum(gaps) / len(gaps) / 1e3, "p50_us": gaps[len(gaps) // 2] / 1e3, "p95_us": gaps[int(len(gaps) * 0.95)] / 1e3}
out["span_ms"] = span / 1e6
out["rounds"] = rounds
out["us_per_round"] = span / 1e3 / rounds
print(json.dumps(out, indent=1))
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
