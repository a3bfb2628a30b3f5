import csv,sys
path=sys.argv[1]; key=sys.argv[2]
rows=[]
for r in csv.DictReader(open(path)):
    rows.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name']))
rows.sort()
idx=[i for i,r in enumerate(rows) if key in r[2]]
a,b=idx[-2],idx[-1]
step=rows[a:b]
t0=step[0][0]
prev_end=None
agg={}
for s,e,n in step:
    gap = (s-prev_end)/1e3 if prev_end else 0
    print('%9.1f us  dur %8.1f  gap %6.1f  %s'%((s-t0)/1e3,(e-s)/1e3,gap,n[:100]))
    prev_end=e
print('span', (step[-1][1]-t0)/1e3, 'launches', len(step))
