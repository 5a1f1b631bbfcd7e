#!/bin/bash
# Tuning aid: GNT sub-benchmark of bench.py (A13 + A14, median of ten) for two builds of the library on ONE box:
# gpurun_ab_old.so / gpurun_ab_new.so at the repo root; the tree's own library is put back when the script ends.
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
    python bench.py --steps 4 --warmup 2 --inflight 1 --no-cpu-baseline --no-kernel-timing --no-scene-sweep --height 270 --width 480 --frames 24 2>/dev/null |
      python -c "import sys,json; g=json.loads(sys.stdin.read().strip().splitlines()[-1])['gnt']; print(g['tflops'], g['frac_of_peak'], g['ms_transformer_A14'], g['repetitions']['tflops_min'], g['repetitions']['tflops_max'])"
  done
done
