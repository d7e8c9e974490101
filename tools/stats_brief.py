#!/usr/bin/env python3
"""rocprofv3 --stats kernel_stats.csv -> short table (name up to '(', calls, average us).  usage: stats_brief.py <dir-or-csv> [n]"""
import csv, glob, os, sys
arg = sys.argv[1]
f = arg if arg.endswith(".csv") else glob.glob(os.path.join(arg, "**", "*kernel_stats.csv"), recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for i, r in enumerate(csv.DictReader(open(f))):
    if i >= n:
        break
    print(f"{r['Name'].split('(')[0].replace('ppcsr::', '')[:60]:60s} calls {int(r['Calls']):7d}  avg {float(r['AverageNs']) / 1e3:9.2f} us  {float(r['Percentage']):5.1f} %")
