#!/usr/bin/env python3
"""Turn rocprofv3 --pmc counter_collection CSVs (one pass per counter) into HBM bytes per launch per kernel.
usage: parse_pmc.py <fetch_csv> <write_csv> <out_json>
Corrections per /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE reports 1/2 of the bytes of a wide coalesced read, so it is doubled (our 12 B/lane accesses are
narrower than the calibrated 16 B/lane case: treat the absolute as approximate, ratios are exact)."""
import csv
import json
import sys
from collections import defaultdict


def load(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("ppcsr::", "")
        tot[k] += float(r["Counter_Value"])
        cnt[k] += 1
    return tot, cnt


def main():
    fetch, fcnt = load(sys.argv[1], "FETCH_SIZE")
    write, wcnt = load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        n = max(fcnt.get(k, 0), wcnt.get(k, 0), 1)
        fb = 2.0 * fetch.get(k, 0.0) * 1024.0 / max(fcnt.get(k, 1), 1)
        wb = write.get(k, 0.0) * 1024.0 / max(wcnt.get(k, 1), 1)
        out[k] = {"launches": n, "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb,
                  "hbm_bytes_per_launch": fb + wb,
                  "note": "FETCH_SIZE x2 (gfx950 correction), KiB units, separate --pmc passes"}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in out.items():
        print(f"{k:24s} launches {v['launches']:7d}  fetch {v['fetch_bytes_per_launch']/1e3:10.1f} KB  write {v['write_bytes_per_launch']/1e3:10.1f} KB")


if __name__ == "__main__":
    main()
