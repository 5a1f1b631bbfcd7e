cd $GRAFT_REPO_ROOT
cp ml-pgdvs_amd/lib/libpgdvs_hip.so gpurun_ab_tree.so
for v in tree r25 r40; do
  cp gpurun_ab_$v.so ml-pgdvs_amd/lib/libpgdvs_hip.so
  echo "== $v"
  bash tools/r05_c.sh 1.5 2>&1 | cut -c1-330
done
cp gpurun_ab_tree.so ml-pgdvs_amd/lib/libpgdvs_hip.so
