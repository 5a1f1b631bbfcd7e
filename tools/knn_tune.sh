#!/bin/bash
for pc in 8 24 64; do
  PGDVS_KNN_STATS=1 PGDVS_KNN_PER_CELL=$pc python tools/knn_probe.py 2>&1 | grep "knn_grid" | head -1
done
