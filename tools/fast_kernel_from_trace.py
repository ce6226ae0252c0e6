"""The launches `roofline.kernel_ms` is measured on, taken from the rocprofv3 kernel trace of the same `python bench.py`
command: bench.py times 20 back-to-back launches of fast_cells_kernel over the full batch with HIP events
(aos2_extractor_bench_fast) after the timed steps; they are the last 20 launches of that kernel in the trace.  (The
per-kernel average of the --stats summary mixes them with the launches of the timed steps, which share the GPU with the
other step in flight and the LocalBA batch.)
    python tools/fast_kernel_from_trace.py <..._kernel_trace.csv> [n_last=20]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "fast_cells_kernel" in r["Kernel_Name"]]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-n:]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in last]
allv = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print("fast_cells_kernel: %d launches in the trace, average %.1f us (all of them: timed steps share the GPU)" % (len(rows), sum(allv) / len(allv)))
print("last %d launches (aos2_extractor_bench_fast, grid %s x %s, alone on the device): average %.1f us, min %.1f, max %.1f"
      % (n, last[0]["Grid_Size_X"], last[0]["Grid_Size_Y"], sum(d) / len(d), min(d), max(d)))
