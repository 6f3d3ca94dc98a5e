#!/bin/bash
# dev: kNN tests + stage timings, split-bf16 coarse pass against the fp32 MFMA one
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_knn_gpu.py tests/test_knn_cfgE_gpu.py -x -q -m gpu > gpurun_out/knn_tests.log 2>&1 || { tail -30 gpurun_out/knn_tests.log; exit 1; }
tail -3 gpurun_out/knn_tests.log
export GLOC3D_KNN_PROF=1
for n in 10000 125000; do
for e in "X=1" "GLOC3D_KNN_NO_BF16X3=1" "GLOC3D_KNN_B3=1,1" "GLOC3D_KNN_B3=1,2" "GLOC3D_KNN_B3=1,4" "GLOC3D_KNN_B3=2,1" "GLOC3D_KNN_B3=2,2" "GLOC3D_KNN_B3=2,4"; do
echo "== $n $e"
env $e python tools/bench_knn.py --n $n --algo 2 --reps 50 --kind 1 | grep -o "^kNN [^:]*: [0-9.]* \|queries_fallback': [0-9]*\|last_n_tile[^}]*\|stage us.*"
done
done
