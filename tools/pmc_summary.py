"""Per-kernel averages of rocprofv3 --pmc counters (counter_collection.csv) joined with kernel durations."""
import csv
import sys
from collections import defaultdict


def main(cc_csv, out_csv=None):
    acc = defaultdict(lambda: defaultdict(float))
    n = defaultdict(lambda: defaultdict(int))
    for r in csv.DictReader(open(cc_csv)):
        k = r["Kernel_Name"].split("(")[0]
        c = r["Counter_Name"]
        acc[k][c] += float(r["Counter_Value"])
        n[k][c] += 1
    rows = []
    counters = sorted({c for k in acc for c in acc[k]})
    for k in acc:
        rows.append([k] + [acc[k][c] / max(n[k][c], 1) for c in counters] + [max(n[k].values())])
    rows.sort(key=lambda r: -r[1])
    hdr = ["Kernel"] + counters + ["dispatches"]
    if out_csv:
        with open(out_csv, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(hdr)
            w.writerows(rows)
    print(hdr)
    for r in rows:
        print([r[0][:48]] + [f"{v:.4g}" for v in r[1:-1]] + [r[-1]])


if __name__ == "__main__":
    main(*sys.argv[1:])
