#!/bin/bash
# dev: kNN tests + stage timings
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_knn_gpu.py tests/test_knn_cfgE_gpu.py -x -q -m gpu > gpurun_out/knn_tests.log 2>&1 || { tail -30 gpurun_out/knn_tests.log; exit 1; }
tail -n 1 gpurun_out/knn_tests.log
export GLOC3D_KNN_PROF=1
for i in 1 2; do
python tools/bench_knn.py --n 10000 --algo 2 --reps 200 --kind 1 | grep -o "^kNN [^:]*: [0-9.]* \|stage us.*"
python tools/bench_knn.py --n 125000 --algo 2 --reps 50 --kind 1 | grep -o "^kNN [^:]*: [0-9.]* \|stage us.*"
python tools/bench_knn.py --n 4541 --q 25 --algo 2 --reps 200 --kind 1 | grep -o "^kNN [^:]*: [0-9.]* \|stage us.*"
done
GLOC3D_KNN_TRACE=1 python tools/bench_knn.py --n 10000 --algo 2 --reps 100 --kind 1 | grep -v "^kNN"
