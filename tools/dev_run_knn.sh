#!/bin/bash
export GLOC3D_KNN_PROF=1
for e in "X=1" "GLOC3D_KNN_B3=2,2" "GLOC3D_KNN_B3=1,1" "GLOC3D_KNN_B3=1,2" "GLOC3D_KNN_B3=2,4" "GLOC3D_KNN_B3_QRAW=0" "GLOC3D_KNN_B3_KO=8 GLOC3D_KNN_B3=1,1"; do
echo "== 125000 $e"
env $e python tools/bench_knn.py --n 125000 --algo 2 --reps 50 --kind 1 | grep -o "^kNN [^:]*: [0-9.]* \|stage us.*"
done
