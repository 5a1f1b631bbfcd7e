#!/bin/bash
# Tuning aid: several builds of libpgdvs_hip.so (gpurun_ab_<tag>.so at the repo root) on ONE box, throughput and isolated
# kernel times: `gpurun -- bash tools/ab_many.sh "tag1 tag2 ..." [bench flags]`; tag `tree` = the tree's own library.
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
LIB=ml-pgdvs_amd/lib/libpgdvs_hip.so
cp "$LIB" /tmp/libpgdvs_hip.orig.so
cp "$LIB" gpurun_ab_tree.so
trap 'cp /tmp/libpgdvs_hip.orig.so "$LIB"' EXIT
tags=$1; shift
for r in 1 2; do
  for v in $tags; do
    cp gpurun_ab_$v.so "$LIB"
    echo -n "$v: "
    python bench.py --steps 40 --warmup 5 --no-cpu-baseline --gnt-rays 0 --no-kernel-timing --no-scene-sweep "$@" 2>/dev/null |
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['latency_ms']['median'], end=' | ')"
    python bench.py --steps 8 --warmup 3 --inflight 1 --no-cpu-baseline --gnt-rays 0 --no-scene-sweep "$@" 2>/dev/null |
      python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('sum', round(sum(v['ms_per_step'] for v in k.values())*1e3), {x: round(v['ms_per_step']*1e3,1) for x,v in k.items() if (x.startswith('grid_query') or x.startswith('raster')) and v['ms_per_step']>=0.015})"
  done
done
