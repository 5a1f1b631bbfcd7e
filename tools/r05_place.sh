#!/bin/bash
# round 5: lane streams picked by hardware queue (--place-streams) against creation order, on one box; STEPS=20: the driver's form
cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}" || exit 1
mkdir -p gpurun_out/r05
for r in 1 2 3; do
  for cfg in "--no-place-streams --inflight 3" "--place-streams --inflight 4 --no-side-stream" "--place-streams --inflight 2" "--place-streams --inflight 3 --no-side-stream"; do
    echo -n "$cfg: "
    python bench.py --steps ${STEPS:-100} --warmup 5 --no-cpu-baseline --gnt-rays 0 --no-kernel-timing --no-scene-sweep $cfg 2>/dev/null |
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], (d.get('steady_state') or {}).get('frames_per_s'), d['latency_ms']['median'])"
  done
done | tee gpurun_out/r05/place_${STEPS:-100}.txt
