import csv,glob,collections
f=glob.glob('gpurun_out/rsz/**/*kernel_trace.csv',recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name']
    if 'resize_level' in n or 'fast_cells' in n or 'describe' in n:
        key=(n.split('(')[0][-24:], r['Grid_Size_X'],r['Grid_Size_Y'])
        d[key].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for k,v in sorted(d.items()):
    v=v[len(v)//3:]
    print(k, len(v), 'avg us %.1f'%(sum(v)/len(v)/1000))
