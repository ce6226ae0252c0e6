"""From a rocprofv3 --kernel-trace csv of bench.py: which kernels run on which hardware queue, and how much of the time two or more
kernels are in flight.  python tools/queue_overlap.py <trace dir>"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the timed region: the last 60 % of the trace
t_lo = int(rows[int(len(rows) * 0.4)]['Start_Timestamp'])
rows = [r for r in rows if int(r['Start_Timestamp']) >= t_lo]
byq = collections.defaultdict(lambda: collections.Counter())
for r in rows:
    byq[r['Queue_Id']][r['Kernel_Name'].split('(')[0].replace('aos2::', '').replace('void ', '')[:34]] += 1
for q, c in sorted(byq.items()):
    print("queue", q, dict(c.most_common(6)))
ev = []
for r in rows:
    ev.append((int(r['Start_Timestamp']), 1))
    ev.append((int(r['End_Timestamp']), -1))
ev.sort()
depth, last, hist = 0, ev[0][0], collections.Counter()
for t, d in ev:
    hist[min(depth, 4)] += t - last
    last = t
    depth += d
tot = sum(hist.values())
print("time with k kernels in flight:", {k: "%.1f %%" % (100.0 * v / tot) for k, v in sorted(hist.items())})
print("sum of kernel durations / wall: %.2f" % (sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows) / tot))
