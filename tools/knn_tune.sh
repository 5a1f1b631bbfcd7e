#!/bin/bash
for pc in 12 16 20 24 32 40; do
  echo "== per_cell $pc"
  PGDVS_KNN_PER_CELL=$pc python tools/knn_probe.py 2>&1 | grep "^ms\|grid_query \|grid2_query"
done
