#!/bin/bash
# round 5: hardware queues (GPU_MAX_HW_QUEUES, read by the HIP runtime at start) x views in flight, with and without the lanes' second streams
cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}" || exit 1
mkdir -p gpurun_out/r05
for q in 4 8 12; do
  for lanes in 3 5 7; do
    for ss in --side-stream --no-side-stream; do
      echo -n "queues=$q lanes=$lanes $ss: "
      GPU_MAX_HW_QUEUES=$q python bench.py --steps 100 --warmup 5 --no-cpu-baseline --gnt-rays 0 --no-kernel-timing --no-scene-sweep --inflight $lanes $ss 2>/dev/null |
        python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['latency_ms']['median'], d.get('host_enqueue_ms_per_step'))"
    done
  done
done | tee gpurun_out/r05/queues.txt
