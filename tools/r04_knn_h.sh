#!/bin/bash
# the grid the kNN builds for the benchmark's dynamic cloud at different density targets (PGDVS_KNN_STATS prints it)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
for pc in 8 10 12 14 16 18 20 22 24 28 32 40; do
  echo "pc=$pc $(PGDVS_KNN_PER_CELL=$pc PGDVS_KNN_STATS=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --gnt-rays 0 --no-scene-sweep --inflight 1 --no-kernel-timing 2>&1 | grep knn_grid | head -1)"
done
