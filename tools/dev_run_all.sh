#!/bin/bash
# dev: the GPU suite, then the bench line and the kNN profiles
set -e
O=gpurun_out; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1 || { tail -40 $O/gpu_tests.log; exit 1; }
tail -n 2 $O/gpu_tests.log
python bench.py --steps 20 --warmup 5 > $O/r04_bench_n1.json 2> $O/r04_bench_n1.err
python tools/bench_line.py < $O/r04_bench_n1.json
bash tools/prof_knn.sh r04_knn_cfgB --algo 2 --reps 200 --kind 1
bash tools/prof_knn.sh r04_knn_shard125k --algo 2 --reps 30 --n 125000 --kind 1
