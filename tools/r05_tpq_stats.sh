cd $GRAFT_REPO_ROOT
cp gpurun_ab_stats.so ml-pgdvs_amd/lib/libpgdvs_hip.so
PGDVS_KNN_STATS=1 python bench.py --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline --gnt-rays 0 --no-scene-sweep --no-kernel-timing 2>&1 | grep knn_grid | tail -2
