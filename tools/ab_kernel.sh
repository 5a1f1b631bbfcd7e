#!/bin/bash
# Tuning aid: isolated kernel times (bench.py's per-kernel HIP events, one view in flight) of two builds of the library on ONE
# box: gpurun_ab_old.so / gpurun_ab_new.so at the repo root, `gpurun -- bash tools/ab_kernel.sh raster_tile grid_query_tpq`.
# The tree's own library is put back when the script ends, however it ends.
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
LIB=ml-pgdvs_amd/lib/libpgdvs_hip.so
for v in old new; do [ -f gpurun_ab_$v.so ] || { echo "missing gpurun_ab_$v.so" >&2; exit 1; }; done
cp "$LIB" /tmp/libpgdvs_hip.orig.so
trap 'cp /tmp/libpgdvs_hip.orig.so "$LIB"' EXIT
for r in 1 2; do
  for v in old new; do
    cp gpurun_ab_$v.so "$LIB"
    echo -n "$v: "
    python bench.py --steps 8 --warmup 2 --inflight 1 --no-cpu-baseline --gnt-rays 0 --no-scene-sweep 2>/dev/null |
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print({n: round(k[n]['ms_per_step']*1e3,1) for n in sys.argv[1:] if n in k}, 'sum', round(sum(v['ms_per_step'] for v in k.values())*1e3), d['latency_ms']['median'])" "$@"
  done
done
