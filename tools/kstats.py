"""print a rocprofv3 kernel_stats.csv (found under a directory) as a table: name, calls, avg / min / total"""
import csv, sys, glob, os
d = sys.argv[1]
f = d if d.endswith(".csv") else sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True))[0]
rows = list(csv.DictReader(open(f)))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for r in rows[:top]:
    print(f"{r['Name'][:64]:64s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.1f} min_us {float(r['MinNs'])/1e3:9.1f} tot_ms {float(r['TotalDurationNs'])/1e6:8.2f} {float(r['Percentage']):5.1f}%")
