#!/bin/bash
# Tuning aid: A/B of two builds of libpgdvs_hip.so on ONE box (boxes of the pool differ by several per cent, runs on one
# box by ~0.5 %): put the two builds at the repo root as gpurun_ab_old.so / gpurun_ab_new.so (they travel with the gpurun
# snapshot), then `gpurun -- bash tools/ab_lib.sh [extra bench flags]`.  Alternates the library in place between short bench
# runs; the tree's own library is put back when the script ends, however it ends.
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
LIB=ml-pgdvs_amd/lib/libpgdvs_hip.so
for v in old new; do [ -f gpurun_ab_$v.so ] || { echo "missing gpurun_ab_$v.so" >&2; exit 1; }; done
cp "$LIB" /tmp/libpgdvs_hip.orig.so
trap 'cp /tmp/libpgdvs_hip.orig.so "$LIB"' EXIT
for r in 1 2 3; do
  for v in old new; do
    cp gpurun_ab_$v.so "$LIB"
    echo -n "$v: "
    python bench.py --steps 60 --warmup 5 --no-cpu-baseline --gnt-rays 0 --no-kernel-timing --no-scene-sweep --inflight 3 "$@" 2>/dev/null |
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['steady_state']['frames_per_s'], d['latency_ms']['median'])"
  done
done
