#!/bin/bash
for a in 0 16; do
  export PGDVS_AGG_ABL=$a
  echo "== abl $a"
  bash tools/prof_stats.sh abl$a | grep "agg_\|frames/s"
done
