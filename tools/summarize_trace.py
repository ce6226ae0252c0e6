"""Per-kernel summary of the LAST program (from the last k_prepare on) of a rocprofv3 kernel trace csv."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
mark = sys.argv[2] if len(sys.argv) > 2 else "k_prepare"
idx = [i for i, r in enumerate(rows) if mark in r["Kernel_Name"]]
seg = rows[idx[-1]:]
t0 = int(seg[0]["Start_Timestamp"])
agg = {}
for r in seg:
    n = r["Kernel_Name"].split("(")[0][-34:]
    agg.setdefault(n, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = (int(seg[-1]["End_Timestamp"]) - t0) / 1e3
print("last program: %d kernels, span %.1f us, kernel sum %.1f us" % (len(seg), tot, sum(sum(v) for v in agg.values())))
for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print("  %-36s n %3d  sum %8.1f  med %7.1f  max %7.1f" % (n, len(v), sum(v), sorted(v)[len(v) // 2], max(v)))
