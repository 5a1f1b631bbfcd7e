#!/bin/bash
# round 4: isolated kernel times of three builds on one box: the tree's library, gpurun_ab_old.so, gpurun_ab_new.so
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r04
LIB=ml-pgdvs_amd/lib/libpgdvs_hip.so
cp "$LIB" /tmp/libpgdvs_hip.orig.so
trap 'cp /tmp/libpgdvs_hip.orig.so "$LIB"' EXIT
for r in 1 2; do
  for v in tree old new; do
    if [ $v = tree ]; then cp /tmp/libpgdvs_hip.orig.so "$LIB"; else cp gpurun_ab_$v.so "$LIB"; fi
    bash tools/r04_kern.sh ab3_${v}_$r | cut -c1-420
  done
done
