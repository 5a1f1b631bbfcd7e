cd $GRAFT_REPO_ROOT
for f in 0.5 0.75 0.9; do
  for sc in nominal noisy_depth; do
    echo -n "frac $f $sc: "
    PGDVS_DBG_FRAC=$f PGDVS_KNN_STATS=1 python bench.py --scene $sc --steps 2 --warmup 1 --inflight 1 --no-side-stream --no-cpu-baseline --gnt-rays 0 --no-kernel-timing --no-scene-sweep 2>&1 | grep "knn_grid. n=311070" | awk "{print \$3, \$4}" | sort | uniq -c | tr '\n' ' '
    echo
  done
done
