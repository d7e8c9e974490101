"""per-kernel summary of a rocprofv3 rocpd database: python tools/prof_db_summary.py file.db [first_kernel_name_of_the_phase]"""
import collections, re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end, grid_x, workgroup_x from kernels order by start"))
cut = 0
if len(sys.argv) > 2:
    for i, r in enumerate(rows):
        if sys.argv[2] in r[0]:
            cut = i
            break
agg = collections.defaultdict(lambda: [0, 0, 0, []])
for name, st, en, gx, wx in rows[cut:]:
    k = re.sub(r"\(.*", "", name).replace("ppcsr::", "")
    a = agg[k]
    a[0] += 1; a[1] += en - st; a[2] = max(a[2], en - st); a[3].append(en - st)
print("kernels", len(rows) - cut, "wall ms", (rows[-1][2] - rows[cut][1]) / 1e6)
for k, (c, t, m, ds) in sorted(agg.items(), key=lambda x: -x[1][1])[:14]:
    ds.sort()
    print(f"{k:28s} calls {c:6d} total {t/1e6:8.2f} ms avg {t/c/1e3:7.1f} us p50 {ds[len(ds)//2]/1e3:7.1f} p90 {ds[int(len(ds)*0.9)]/1e3:7.1f} max {m/1e3:8.1f}")
