#!/bin/bash
# round 5 (review item 3): what the static aggregation's chain of 23 links costs the THROUGHPUT, and which part of it --
# by ADDING things behind the real chain (a link is idempotent, so a duplicate chain changes no result):
#   dup        a second, complete copy of the chain                     -> the chain's full marginal cost
#   dup_dry    a copy that computes everything and stores nothing       -> its workgroups + reads, without the stamp / row traffic
#   dup_small  23 launches of 512 workgroups that leave at once         -> launch boundaries + dispatch
#   dup_empty  23 launches of one empty workgroup                       -> launch boundaries (cache write-back / invalidate) alone
# Needs gpurun_ab_chain.so at the repo root: make -C ml-pgdvs_amd/csrc OUT=../../gpurun_ab_chain.so OBJDIR=/tmp/ab_chain EXTRA=-DPGDVS_AB_CHAIN
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}" || exit 1
LIB=ml-pgdvs_amd/lib/libpgdvs_hip.so
cp "$LIB" /tmp/libpgdvs_hip.orig.so
trap 'cp /tmp/libpgdvs_hip.orig.so "$LIB"' EXIT
cp gpurun_ab_chain.so "$LIB"
mkdir -p gpurun_out/r05
for r in 1 2 3; do
  for m in none dup dup_dry dup_small dup_empty; do
    echo -n "$m: "
    PGDVS_DBG_CHAIN=$m python bench.py --steps 100 --warmup 5 --no-cpu-baseline --gnt-rays 0 --no-kernel-timing --no-scene-sweep --inflight 3 "$@" 2>gpurun_out/r05/chain_cost.err |
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['latency_ms']['median'])"
  done
done | tee gpurun_out/r05/chain_cost.txt
