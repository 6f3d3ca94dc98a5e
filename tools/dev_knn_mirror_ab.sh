# dev (round 6): the coarse pass over the tiled bf16 mirror against the row-major kernel, same box: cfg B and the 125 000-row shard
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for v in 0 1; do
    if [ $v = 1 ]; then export GLOC3D_KNN_NO_MIRROR=1; else unset GLOC3D_KNN_NO_MIRROR; fi
    echo "NO_MIRROR=$v cfgB : $(GLOC3D_KNN_PROF=1 python3 tools/bench_knn.py --kind 1 --reps 300 2>/dev/null | tr '\n' ' ' | sed 's/stats {[^}]*}//' | cut -c1-330)"
    echo "NO_MIRROR=$v shard: $(GLOC3D_KNN_PROF=1 python3 tools/bench_knn.py --kind 1 --n 125000 --reps 40 2>/dev/null | tr '\n' ' ' | sed 's/stats {[^}]*}//' | cut -c1-330)"
  done
done
