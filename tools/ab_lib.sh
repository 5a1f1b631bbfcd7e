#!/bin/bash
# Tuning aid: A/B of two builds of libpgdvs_hip.so on ONE box (boxes of the pool differ by several per cent, runs on one
# box by ~0.5 %): put the two builds at the repo root as gpurun_ab_old.so / gpurun_ab_new.so (they travel with the gpurun
# snapshot), then `gpurun -- bash tools/ab_lib.sh`.  Alternates the library in place between short bench runs.
cd ${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}
for r in 1 2 3; do
  for v in old new; do
    cp gpurun_ab_$v.so ml-pgdvs_amd/lib/libpgdvs_hip.so
    echo -n "$v: "
    python bench.py --steps 60 --warmup 5 --no-cpu-baseline --gnt-rays 0 --no-kernel-timing --inflight 7 2>/dev/null |
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['steady_state']['frames_per_s'])"
  done
done
