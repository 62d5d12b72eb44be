import csv,sys,glob
f=glob.glob('gpurun_out/shprof/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
n=len(rows)
t0=int(rows[n//2]['Start_Timestamp'])
for r in rows[n//2:n//2+24]:
    print("%9.1f %9.1f dur %7.1f q%s  %s"%((int(r['Start_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,r.get('Queue_Id','?'),r['Kernel_Name'][:70]))
