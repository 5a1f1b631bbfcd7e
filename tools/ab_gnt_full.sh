#!/bin/bash
# Tuning aid: as ab_gnt.sh, at the bench line's own GNT configuration (1080p x 24 resident source frames: the sizes the
# committed figure is quoted on; ab_gnt.sh's 270 x 480 frames leave more (tile, view) pairs without a valid projection,
# which the view layer skips while the FLOP count stays nominal).
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
    python bench.py --steps 4 --warmup 2 --inflight 1 --no-cpu-baseline --no-kernel-timing --no-scene-sweep 2>/dev/null |
      python -c "import sys,json; g=json.loads(sys.stdin.read().strip().splitlines()[-1])['gnt']; print(g['tflops'], g['frac_of_peak'], g['ms_transformer_A14'], g['ms_gather_A13'], g['valid_projection_fraction'], g['repetitions']['tflops_min'], g['repetitions']['tflops_max'])"
  done
done
