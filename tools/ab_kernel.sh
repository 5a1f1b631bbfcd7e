#!/bin/bash
# Tuning aid: isolated kernel times (bench.py's per-kernel HIP events, one view in flight) of two builds of the library on ONE
# box: gpurun_ab_old.so / gpurun_ab_new.so at the repo root, `gpurun -- bash tools/ab_kernel.sh raster_tile grid_query_tpq`.
cd ${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}
for r in 1 2; do
  for v in old new; do
    cp gpurun_ab_$v.so ml-pgdvs_amd/lib/libpgdvs_hip.so
    echo -n "$v: "
    python bench.py --steps 8 --warmup 2 --inflight 1 --launch eager --no-cpu-baseline --gnt-rays 0 2>/dev/null |
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print({n: round(k[n]['avg_ms']*1e3,1) for n in sys.argv[1:]}, d['latency_ms']['median'])" "$@"
  done
done
