#!/bin/bash
# round 5: the depth-sliced tile pass -- libraries at the repo root (gpurun_ab_<tag>.so) against the tree's, isolated kernel times
# (one view in flight) and throughput on ONE box
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}" || exit 1
mkdir -p gpurun_out/r04 gpurun_out/r05
LIB=ml-pgdvs_amd/lib/libpgdvs_hip.so
cp "$LIB" /tmp/libpgdvs_hip.orig.so
trap 'cp /tmp/libpgdvs_hip.orig.so "$LIB"' EXIT
for v in "$@"; do
  if [ $v = tree ]; then cp /tmp/libpgdvs_hip.orig.so "$LIB"; else cp gpurun_ab_$v.so "$LIB"; fi
  bash tools/r04_kern.sh rast_$v | cut -c1-330
  echo -n "$v throughput: "
  python bench.py --steps 100 --warmup 5 --no-cpu-baseline --gnt-rays 0 --no-kernel-timing --no-scene-sweep --inflight 3 2>/dev/null |
    python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['latency_ms']['median'])"
done | tee gpurun_out/r05/raster_ab.txt
