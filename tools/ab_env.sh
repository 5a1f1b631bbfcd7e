#!/bin/bash
# Tuning aid: A/B of an environment switch of bench.py on ONE box: `gpurun -- bash tools/ab_env.sh NAME=VALUE [bench args]`
# alternates runs without and with the variable set (driver form: 20 timed views).
cd ${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}
kv=$1; shift
for r in 1 2 3 4; do
  for v in off on; do
    echo -n "$v: "
    if [ $v = on ]; then export "$kv"; else unset "${kv%%=*}"; fi
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --gnt-rays 0 --no-kernel-timing --no-scene-sweep "$@" 2>/dev/null |
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['steady_state']['frames_per_s'], d['host_enqueue_ms_per_step'], d['config']['views_in_flight'])"
  done
done
