import csv,glob,collections,os,sys
f=max(glob.glob(sys.argv[1]+'/*/*_kernel_trace.csv'),key=os.path.getmtime)
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last 40% of the run = steady loop
n=len(rows); seg=rows[int(n*0.5):int(n*0.95)]
t0=int(seg[0]['Start_Timestamp']); t1=max(int(r['End_Timestamp']) for r in seg)
d=collections.defaultdict(list)
for r in seg:
    nme=r['Kernel_Name']; i=nme.find('::'); d[(nme[i+2:i+32] if i>=0 else nme[:30], r['Grid_Size_X'])].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
print('span %.2f ms, kernels %d'%((t1-t0)/1e6,len(seg)))
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1])): print(k, len(v), 'avg %.1f us total %.2f ms'%(sum(v)/len(v)/1e3, sum(v)/1e6))
gap=0; prev=int(seg[0]['End_Timestamp'])
for r in seg[1:]:
    s=int(r['Start_Timestamp'])
    if s>prev: gap+=s-prev
    prev=max(prev,int(r['End_Timestamp']))
print('idle gaps %.2f ms'%(gap/1e6))
