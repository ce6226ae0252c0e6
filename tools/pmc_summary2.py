"""Per-kernel averages of a rocprofv3 --pmc counter_collection.csv: python tools/pmc_summary2.py <csv> [<csv> ...]"""
import csv, sys, collections
for path in sys.argv[1:]:
    rows = list(csv.DictReader(open(path)))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        name = r.get("Kernel_Name", r.get("Kernel-Name", "?")).split("(")[0][-40:]
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(path)
    for k, cs in agg.items():
        print("  %-42s %s" % (k, "  ".join("%s avg %.4g (n=%d)" % (c, sum(v) / len(v), len(v)) for c, v in cs.items())))
