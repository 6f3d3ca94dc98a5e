cd $GRAFT_REPO_ROOT
for n in 98304 125000 147456 196608 250000; do
  echo "n=$n: $(GLOC3D_KNN_PROF=1 python3 tools/bench_knn.py --kind 1 --n $n --reps 30 2>/dev/null | tr '\n' ' ' | sed 's/stats {[^}]*}//;s/(wall, device resident).*TB\/s;//' | cut -c1-200)"
done
