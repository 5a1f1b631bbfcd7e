#!/bin/bash
# kernel trace of a default-style run with three views in flight: union of kernel intervals, idle
# gaps, mean number of kernels running, kernel time per view.  Under the tracer the run is ~1.5x
# slower and at most two kernels overlap (mean 1.1): read the per-kernel times per view from it,
# not the overlap an untraced run reaches.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/q_ov
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/q_ov -o k -- python3 $R/bench.py --steps 40 --warmup 3 --launch eager --no-cpu-baseline --no-kernel-timing --gnt-rays 0 > $R/gpurun_out/q_ov.log 2>&1
tail -1 $R/gpurun_out/q_ov.log | cut -c1-220
python3 - <<'PY'
import csv,glob,os,collections
f=glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/q_ov/**/*kernel_trace.csv',recursive=True)[0]
rows=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'].split('(')[0].replace('pgdvs::','').replace('void ','')) for r in csv.DictReader(open(f))]
rows.sort()
# last 40 views: find the raster_tile dispatches, take the window between the 20th-last and the last
rt=[r for r in rows if 'raster_tile' in r[2]]
t0,t1=rt[-31][0],rt[-1][0]
win=[r for r in rows if r[0]>=t0 and r[0]<t1]
ev=[]
for s,e,_ in win: ev.append((s,1)); ev.append((min(e,t1),-1))
ev.sort()
busy=0;conc=0;cur=0;last=t0;hist=collections.Counter()
for t,d in ev:
    hist[min(cur,6)]+=t-last
    if cur>0: busy+=t-last
    conc+=cur*(t-last); last=t; cur+=d
tot=t1-t0
print(f"window {tot/1e6:.2f} ms for 30 views = {tot/30e3:.1f} us/view; busy {busy/tot:.3f}; mean kernels running {conc/tot:.2f}")
print("time share by number of kernels running:", {k: round(v/tot,3) for k,v in sorted(hist.items())})
per=collections.Counter()
for s,e,n in win: per[n]+=e-s
print("kernel time per view (us):", [(n,round(v/30e3,1)) for n,v in per.most_common(12)])
PY
find $R/gpurun_out/q_ov -name "*kernel_trace.csv" -delete
