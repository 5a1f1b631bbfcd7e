#!/bin/bash
# round 4: frames per workgroup row of the aggregation's launches, re-measured with three lanes and the native call
cd "${GRAFT_REPO_ROOT:?}" || exit 1
B="python bench.py --steps 60 --warmup 5 --no-cpu-baseline --gnt-rays 0 --no-kernel-timing --no-scene-sweep --inflight 3"
for r in 1 2; do
for cfg in "X=0" "PGDVS_AGG_FPG=6" "PGDVS_AGG_FPG=12" "PGDVS_AGG_STEP_FPG=8" "PGDVS_AGG_STEP_FPG=12" "PGDVS_AGG_FUSED0=1"; do
  echo -n "$cfg: "
  env $cfg $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['steady_state']['frames_per_s'], d['latency_ms']['median'])"
done; done
