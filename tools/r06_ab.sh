#!/bin/bash
# A/B of several builds of libpgdvs_hip.so on ONE box (round 6; reads bench.py's detail record, which left the stdout line):
#   make -C ml-pgdvs_amd/csrc OUT=../../gpurun_ab_<tag>.so OBJDIR=/tmp/ab_<tag> [EXTRA=-D...]     (here, per variant)
#   gpurun -- bash tools/r06_ab.sh "<tag> <tag> ..." [kernel names whose isolated time to print ...]
# per variant and round: isolated kernel times (one view in flight, one stream) of the named kernels, then the throughput of
# 100 views with the arrangement probe (value / steady state).  BENCH_ARGS: extra bench flags (scene, size).  ROUNDS (default 2).
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
tags=$1; shift
LIB=ml-pgdvs_amd/lib/libpgdvs_hip.so
cp "$LIB" /tmp/libpgdvs_hip.orig.so
trap 'cp /tmp/libpgdvs_hip.orig.so "$LIB"' EXIT
for r in $(seq 1 ${ROUNDS:-2}); do
  for v in $tags; do
    cp gpurun_ab_$v.so "$LIB"
    python bench.py ${BENCH_ARGS:-} --steps 8 --warmup 3 --inflight 1 --no-cpu-baseline --gnt-rays 0 --no-scene-sweep 2>&1 >/dev/null | grep "^bench detail: " | cut -c15- |
      python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['kernels']; want=sys.argv[2:]
sel={x: round(v['ms_per_step']*1e3,1) for x,v in sorted(k.items(), key=lambda kv:-kv[1]['ms_per_step']) if (x in want if want else v['ms_per_step']>=0.02)}
print(sys.argv[1], 'kernels: lat', d['latency_ms']['median'], 'sum', round(sum(v['ms_per_step'] for v in k.values())*1e3), sel)" $v "$@"
    python bench.py ${BENCH_ARGS:-} --steps 100 --warmup 5 --no-cpu-baseline --gnt-rays 0 --no-kernel-timing --no-scene-sweep 2>/dev/null |
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], 'throughput:', d['value'], 'lanes', d['config']['views_in_flight'], 'latency', d['latency_ms']['median'])" $v
  done
done
