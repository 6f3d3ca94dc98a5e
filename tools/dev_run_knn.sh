#!/bin/bash
# dev: cost of the adaptive coarse form's bookkeeping + the fuzz log
mkdir -p gpurun_out
for e in "X=1" "GLOC3D_KNN_NO_ADAPT=1" "X=1" "GLOC3D_KNN_NO_ADAPT=1"; do
echo "== $e"
env $e python tools/bench_knn.py --n 10000 --algo 2 --reps 300 --kind 1 | grep -o "^kNN [^:]*: [0-9.]* "
env $e python tools/bench_knn.py --n 4541 --q 25 --algo 2 --reps 300 --kind 1 | grep -o "^kNN [^:]*: [0-9.]* "
done
timeout -k 10 600 python tools/fuzz_knn.py 300 11 > gpurun_out/r04_knn_fuzz.txt 2>&1; tail -n 2 gpurun_out/r04_knn_fuzz.txt
