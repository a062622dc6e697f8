"""Concurrency of the overlapped headline schedule from a rocprofv3 kernel trace (rocpd .db):
    rocprofv3 --kernel-trace -d gpurun_out/tl -o tl -- python bench.py --no-cpu-baseline --no-kernel-times --no-train-step --no-general-route --steps 16 --warmup 2
    python profiles/analysis/timeline_concurrency.py gpurun_out/tl
Prints, over the second half of the run (the timed region): the fraction of time any kernel runs, the distribution of the
number of kernels running at once, every kernel's summed duration over the span (= how many instances run on average)
and the occupancy of every hardware queue."""
import sqlite3,glob,sys
f=glob.glob((sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/tl") + "/*.db")[0]
con=sqlite3.connect(f); cur=con.cursor()
tabs=[r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks=[t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols=[r[1] for r in cur.execute(f"pragma table_info({kd})")]
print(cols)
rows=list(cur.execute(f"select d.start,d.end,s.kernel_name,d.queue_id from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
print(len(rows))
# take the last 60% of the timeline (timed region)
t0=rows[0][0]; t1=rows[-1][1]
lo=t0+(t1-t0)*0.5; hi=t0+(t1-t0)*0.95
sel=[r for r in rows if r[0]>=lo and r[1]<=hi]
# union coverage and concurrency
ev=[]
for s,e,n,q in sel: ev.append((s,1)); ev.append((e,-1))
ev.sort()
cur_c=0; last=ev[0][0]; busy=0; hist={}
for t,dl in ev:
    dt=t-last
    if dt>0:
        hist[cur_c]=hist.get(cur_c,0)+dt
        if cur_c>0: busy+=dt
    cur_c+=dl; last=t
span=ev[-1][0]-ev[0][0]
print("span ms",span/1e6,"busy frac",busy/span)
for c in sorted(hist): print("concurrency",c,"frac %.3f"%(hist[c]/span))
# per kernel total time share
agg={}
for s,e,n,q in sel: agg[n[:40]]=agg.get(n[:40],0)+(e-s)
for n,v in sorted(agg.items(),key=lambda x:-x[1])[:12]: print("%-42s %.3f (sum dur / span)"%(n,v/span))
qs={}
for s,e,n,q in sel: qs[q]=qs.get(q,0)+(e-s)
print({q:round(v/span,3) for q,v in qs.items()})
