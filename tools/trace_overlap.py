#!/usr/bin/env python3
"""rocprofv3 kernel-trace CSV -> for the LAST 35 % of the run (the timed steps): wall span, union of kernel intervals (GPU busy),
sum of kernel durations (overlap = sum / union), per-kernel totals, and the same per stream/queue."""
import csv
import sys
from collections import defaultdict

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], r.get("Queue_Id", r.get("Stream_Id", "?"))))
rows.sort()
t0, t1 = rows[0][0], rows[-1][1]
cut = t1 - int(0.35 * (t1 - t0))
sel = [x for x in rows if x[0] >= cut]
span = sel[-1][1] - sel[0][0]
busy, cur_s, cur_e = 0, None, None
for s, e, _, _ in sel:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, _, _ in sel)
print(f"window {span / 1e6:.1f} ms, GPU busy (union) {busy / 1e6:.1f} ms, sum of kernel durations {tot / 1e6:.1f} ms, overlap factor {tot / busy:.2f}, kernels {len(sel)}")
byk = defaultdict(lambda: [0, 0])
for s, e, k, _ in sel:
    byk[k][0] += e - s
    byk[k][1] += 1
for k, (d, c) in sorted(byk.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"  {k[:48]:48s} {c:6d} x avg {d / c / 1e3:8.1f} us = {d / 1e6:8.2f} ms")
byq = defaultdict(int)
for s, e, _, q in sel:
    byq[q] += e - s
print("per queue ms:", {q: round(v / 1e6, 1) for q, v in sorted(byq.items())})
