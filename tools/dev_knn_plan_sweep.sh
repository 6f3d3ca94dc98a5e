# dev: the split-bf16 coarse kernel (round 6: over the tiled mirror) under every tile / split-K plan at the 125 000-row shard and cfg B
cd $GRAFT_REPO_ROOT
for plan in "" "1,1" "1,2" "2,1" "2,2" "2,4" "1,4"; do
  if [ -n "$plan" ]; then export GLOC3D_KNN_B3=$plan; else unset GLOC3D_KNN_B3; fi
  echo "B3=${plan:-default} shard: $(GLOC3D_KNN_PROF=1 python3 tools/bench_knn.py --kind 1 --n 125000 --reps 40 2>/dev/null | tr '\n' ' ' | sed 's/stats {[^}]*}//;s/(wall, device resident).*TB\/s;//' | cut -c1-230)"
done
for plan in "" "1,2" "1,4" "1,8" "2,4" "2,8"; do
  if [ -n "$plan" ]; then export GLOC3D_KNN_B3=$plan; else unset GLOC3D_KNN_B3; fi
  echo "B3=${plan:-default} cfgB : $(GLOC3D_KNN_PROF=1 python3 tools/bench_knn.py --kind 1 --reps 300 2>/dev/null | tr '\n' ' ' | sed 's/stats {[^}]*}//;s/(wall, device resident).*TB\/s;//' | cut -c1-230)"
done
