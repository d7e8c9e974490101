#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 --pmc counter_collection CSVs.  usage: pmc_summary.py <dir-or-csv>... (prints a table, writes JSON to stdout end)"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

tot = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for arg in sys.argv[1:]:
    files = [arg] if arg.endswith(".csv") else glob.glob(os.path.join(arg, "**", "*counter_collection.csv"), recursive=True)
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("ppcsr::", "")
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
out = {}
for k in sorted(tot):
    out[k] = {c: tot[k][c] / max(cnt[k][c], 1) for c in sorted(tot[k])}
    out[k]["launches"] = max(cnt[k].values())
keep = [k for k in out if k.startswith(("o_", "k_rb", "k_scan"))]
for k in keep:
    print(k, json.dumps({a: round(b, 1) for a, b in out[k].items()}))
