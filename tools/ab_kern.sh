#!/bin/bash
# Tuning aid: isolated kernel times (one view in flight, one stream) of two builds of libpgdvs_hip.so on ONE box:
# gpurun_ab_old.so / gpurun_ab_new.so at the repo root, `gpurun -- bash tools/ab_kern.sh [kernel ...]` (tools/ab_lib.sh is the
# throughput form of the same comparison).
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
LIB=ml-pgdvs_amd/lib/libpgdvs_hip.so
cp "$LIB" /tmp/libpgdvs_hip.orig.so
trap 'cp /tmp/libpgdvs_hip.orig.so "$LIB"' EXIT
for r in 1 2; do
  for v in old new; do
    cp gpurun_ab_$v.so "$LIB"
    python bench.py ${BENCH_ARGS:-} --steps 8 --warmup 3 --inflight 1 --no-cpu-baseline --gnt-rays 0 --no-scene-sweep 2>/dev/null |
      python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; want=sys.argv[2:]
sel={x: round(v['ms_per_step']*1e3,1) for x,v in sorted(k.items(), key=lambda kv:-kv[1]['ms_per_step']) if (x in want if want else v['ms_per_step']>=0.02)}
print(sys.argv[1], 'lat', d['latency_ms']['median'], 'sum', round(sum(v['ms_per_step'] for v in k.values())*1e3), sel, d.get('knn_queries_to_ring_search'))" $v "$@"
  done
done
