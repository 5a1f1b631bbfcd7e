cd $GRAFT_REPO_ROOT
for c in "$@"; do
  for sc in nominal noisy_depth; do
    echo -n "c $c $sc: "
    PGDVS_DBG_C=$c PGDVS_KNN_STATS=1 python bench.py --scene $sc --steps 2 --warmup 1 --inflight 1 --no-side-stream --no-cpu-baseline --gnt-rays 0 --no-kernel-timing --no-scene-sweep 2>&1 | grep "knn_grid. n=311070" | awk '{print $4, $12, $13, $14, $NF}' | sort | uniq -c | tr '\n' ' '
    echo
    PGDVS_DBG_C=$c python bench.py --scene $sc --steps 8 --warmup 3 --inflight 1 --no-cpu-baseline --gnt-rays 0 --no-scene-sweep 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('   sum', round(sum(v['ms_per_step'] for v in k.values())*1e3), {x: round(v['ms_per_step']*1e3,1) for x,v in k.items() if x.startswith('grid') and v['ms_per_step']>=0.012})"
    PGDVS_DBG_C=$c python bench.py --scene $sc --steps 40 --warmup 5 --no-cpu-baseline --gnt-rays 0 --no-kernel-timing --no-scene-sweep 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   frames/s', d['value'])"
  done
done
