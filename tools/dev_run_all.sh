#!/bin/bash
# dev: the GPU suite, then the bench line, the kNN profiles and the HBM access-shape probe
set -e
O=gpurun_out; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1 || { tail -40 $O/gpu_tests.log; exit 1; }
tail -n 2 $O/gpu_tests.log
python bench.py --steps 20 --warmup 5 > $O/r04_bench_n1.json 2> $O/r04_bench_n1.err
python tools/bench_line.py < $O/r04_bench_n1.json
bash tools/prof_knn.sh r04_knn_cfgB --algo 2 --reps 200 --kind 1
bash tools/prof_knn.sh r04_knn_shard125k --algo 2 --reps 30 --n 125000 --kind 1
timeout -k 5 120 tools/_bin/stream_pattern > $O/r04_hbm_stream_patterns.txt 2>&1
bash tools/dev_knn_pmc.sh r04_knn_shard125k dist_bf16x3 --n 125000 --algo 2 --kind 1 > $O/r04_knn_shard125k_pmc.txt 2>&1
for e in "GLOC3D_KNN_B3_PHASE=0" "GLOC3D_KNN_B3_PHASE=5" "GLOC3D_KNN_NO_BF16X3=1" "GLOC3D_KNN_NO_SLICES=1"; do
echo "== $e"; GLOC3D_KNN_PROF=1 env $e python tools/bench_knn.py --n 125000 --algo 2 --reps 50 --kind 1 | grep -o "^kNN [^:]*: [0-9.]* \|stage us.*"
done > $O/r04_knn_shard125k_switches.txt 2>&1
cat $O/r04_knn_shard125k_switches.txt
