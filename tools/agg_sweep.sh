#!/bin/bash
# tuning aid: pgdvs_static_aggregate per kernel (tools/agg_probe.py) under a few settings of the chain
set -e
cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
for cfg in "PGDVS_AGG_STEP_FPG=6" "PGDVS_AGG_STEP_FPG=6 PGDVS_AGG_FPG=12" "PGDVS_AGG_STEP_FPG=6 PGDVS_AGG_FPG=6" "PGDVS_AGG_STEP_FPG=6 PGDVS_AGG_FPG=4"; do
  echo "== $cfg"
  env $cfg python3 tools/agg_probe.py 2>&1 | grep -E "whole|agg_"
done
