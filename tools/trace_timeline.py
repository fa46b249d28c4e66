import csv,sys
k=list(csv.DictReader(open(sys.argv[1])))
m=list(csv.DictReader(open(sys.argv[2])))
ev=[]
for r in k: ev.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'].replace('void ','').replace('melf::','')[:34], r.get('Queue_Id')))
for r in m: ev.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),'COPY '+r['Direction'][12:], ''))
ev.sort()
N=int(sys.argv[3]) if len(sys.argv)>3 else 60
last=ev[-N:]
t0=last[0][0]
for s,e,n,q in last:
    print('%8.3f %8.3f  %7.3f  %s %s'%((s-t0)/1e6,(e-t0)/1e6,(e-s)/1e6,n,q))
