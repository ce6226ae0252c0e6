# prints the dispatch timeline (start / end in us, queue) of the last full bench step from a rocprofv3 --kernel-trace csv
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# a step = everything between two consecutive first-level resize launches on the same queue
idx = [i for i, r in enumerate(rows) if 'resize_level' in r['Kernel_Name'] and r['Grid_Size_Y'] == '50']
qs = sorted(set(r['Queue_Id'] for r in rows))
nper = int(sys.argv[2]) if len(sys.argv) > 2 else 2
step = int(sys.argv[3]) if len(sys.argv) > 3 else None
a, b = (idx[-2 * nper], idx[-nper]) if step is None else (idx[nper * step], idx[nper * (step + 1)])
t0 = int(rows[a]['Start_Timestamp'])
for r in rows[a:b]:
    n = r['Kernel_Name'].split('(')[0].replace('aos2::', '')
    print('q%d %-28s grid %6s x %-4s  %8.1f -> %8.1f  (%6.1f us)' % (qs.index(r['Queue_Id']), n[:28], r['Grid_Size_X'], r['Grid_Size_Y'],
          (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
