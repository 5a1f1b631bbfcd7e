cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/g_stats
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/g_stats -o k -- python3 $R/bench.py --steps 20 --warmup 3 --launch graph --no-cpu-baseline --no-kernel-timing --gnt-rays 0 > $R/gpurun_out/g_stats.log 2>&1
tail -1 $R/gpurun_out/g_stats.log | cut -c1-300
python3 - <<'PY'
import csv,glob,os
f=glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/g_stats/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:16]: print(r['Name'][:60].ljust(60), r['Calls'].rjust(6), f"{float(r['TotalDurationNs'])/1e3:10.0f}us", f"{float(r['AverageNs'])/1e3:8.1f}us")
PY
find $R/gpurun_out/g_stats -name "*kernel_trace.csv" -delete
