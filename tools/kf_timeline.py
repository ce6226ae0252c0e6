"""kernel timeline of the last steps of a profiled bench run, grouped by HIP queue: when do the keyframe legs' kernels run relative to their step's extraction?"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
qs = sorted(set(r['Queue_Id'] for r in rows))
name = lambda r: r['Kernel_Name'].split('(')[0].replace('aos2::', '').replace('void ', '').replace('(anonymous namespace)::', '')[:34]
fb = [i for i, r in enumerate(rows) if 'frames_build_kernel' in r['Kernel_Name']]   # one per step
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
a, z = fb[-nsteps - 1], fb[-1]
# back to the extraction that belongs to step a: the first resize_level launch before it
while a > 0 and 'resize_level' not in rows[a]['Kernel_Name']:
    a -= 1
while a > 0 and 'resize_level' in rows[a - 1]['Kernel_Name']:
    a -= 1
rows = rows[:z]
t0 = int(rows[a]['Start_Timestamp'])
print("queues:", len(qs), " rows", len(rows), " the last %d steps before the final one; times in us" % nsteps)
def cls(n):
    if n.startswith('k_'): return 'LBA'
    if n.startswith('voc_') or n.startswith('bow_'): return 'BOW'
    if 'triang' in n or 'fuse' in n: return 'KFW'
    if 'pose_opt' in n or n.startswith('frames_'): return 'TRK'
    return 'EXT'
last_end = {}
for r in rows[a:]:
    n = name(r); c = cls(n); q = qs.index(r['Queue_Id'])
    s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
    gap = s - last_end.get(q, s)
    last_end[q] = e
    if c == 'LBA' and e - s < 150: continue
    print('q%-2d %-3s %-34s grid %7s x %-4s %9.1f -> %9.1f (%7.1f us; %7.1f after the queue\'s previous kernel)' % (q, c, n, r['Grid_Size_X'], r['Grid_Size_Y'], s, e, e - s, gap))
# occupancy summary: per class, sum of durations and span
import collections
d = collections.defaultdict(float)
for r in rows[a:]:
    d[cls(name(r))] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
span = (int(rows[-1]['End_Timestamp']) - t0) / 1e3
print("span %.1f us; summed kernel durations per class:" % span, {k: round(v, 1) for k, v in d.items()})
