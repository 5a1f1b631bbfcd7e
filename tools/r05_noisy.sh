cd $GRAFT_REPO_ROOT
python bench.py --scene noisy_depth --steps 20 --warmup 5 --no-cpu-baseline --gnt-rays 0 --no-scene-sweep --no-kernel-timing 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['config'].get('views_in_flight'), d.get('counters'))"
python bench.py --scene noisy_depth --steps 8 --warmup 3 --inflight 1 --no-cpu-baseline --gnt-rays 0 --no-scene-sweep 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print({x: round(v['ms_per_step']*1e3,1) for x,v in sorted(k.items(), key=lambda kv:-kv[1]['ms_per_step']) if v['ms_per_step']>=0.02})"
PGDVS_KNN_STATS=1 python bench.py --scene noisy_depth --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline --gnt-rays 0 --no-scene-sweep --no-kernel-timing 2>&1 | grep knn_grid | tail -2
