"""Per-kernel summary (calls, total/avg/min/max ns, share) from a rocprofv3 rocpd sqlite file
(`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd` writes DIR/NAME_results.db on ROCm 7.2)."""
import csv
import sqlite3
import sys


def main(db_path, out_csv):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
    scol = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    name_col = "kernel_name" if "kernel_name" in scol else ("display_name" if "display_name" in scol else scol[-1])
    q = (f"select s.{name_col}, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
         "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by 1 order by 3 desc")
    rows = list(cur.execute(q))
    tot = sum(r[2] for r in rows) or 1
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
        for r in rows:
            w.writerow([r[0], r[1], r[2], f"{r[3]:.1f}", r[4], r[5], f"{100.0 * r[2] / tot:.2f}"])
    for r in rows[:12]:
        print(f"{r[0][:70]:70s} calls={r[1]:5d} avg={r[3] / 1e3:9.1f} us  {100.0 * r[2] / tot:5.1f}%")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
