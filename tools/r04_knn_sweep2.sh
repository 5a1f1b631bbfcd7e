#!/bin/bash
# round 4: kNN density target against tpq + ring + coarse times (nominal and noisy scenes), with the grid it builds
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r04
for sc in ${SCENES:-nominal noisy_depth}; do
for pc in 11 13 15 16 17 18 20; do
  h=$(PGDVS_KNN_PER_CELL=$pc PGDVS_KNN_STATS=1 python bench.py --scene $sc --steps 2 --warmup 1 --no-cpu-baseline --gnt-rays 0 --no-scene-sweep --inflight 1 --no-kernel-timing 2>&1 | grep knn_grid | head -1 | sed 's/.*h=\([0-9.e-]*\) .*rings: \([0-9]* [0-9]*\).*to level 2: \([0-9]*\).*/h=\1 rings=\2 lvl2=\3/')
  PGDVS_KNN_PER_CELL=$pc python bench.py --scene $sc --steps 6 --warmup 2 --inflight 1 --no-side-stream --no-cpu-baseline --gnt-rays 0 --no-scene-sweep 2>/dev/null | python -c "
import sys,json
b=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=b['kernels']
g=lambda n: round(k.get(n,{}).get('ms_per_step',0)*1e3,1)
print('$sc pc=$pc $h', 'tpq', g('grid_query_tpq'), 'ring', g('grid_query'), 'coarse', g('grid2_query'), 'sum3', round(g('grid_query_tpq')+g('grid_query')+g('grid2_query'),1))"
done; done
