set -e
O=gpurun_out
python bench.py --steps 20 --warmup 5 > $O/r04_bench_n1.json 2> $O/r04_bench_n1.err
python tools/bench_line.py < $O/r04_bench_n1.json
bash tools/prof_knn.sh r04_knn_cfgB --algo 2 --reps 200
bash tools/prof_knn.sh r04_knn_shard125k --algo 2 --reps 30 --n 125000
python tools/dev_nn_trace3.py 20 25 > $O/r04_nn_trace_500jobs.txt 2>&1
python tools/dev_nn_trace3.py 0 8 > $O/r04_nn_trace_cold_160jobs.txt 2>&1
python tools/dev_nn_timeline.py 256 60000 > $O/r04_nn_timeline_lone_query_planned.txt 2>&1
python tools/dev_nn_timeline.py 0 0 > $O/r04_nn_timeline_lone_query_unplanned.txt 2>&1
python tools/dev_split_sweep.py 0,0 256,60000 > $O/r04_lone_query_plans.txt 2>&1
cat $O/r04_lone_query_plans.txt
python tools/bench_ground.py > $O/r04_ground_stage_bench_noprof.txt 2>&1; tail -7 $O/r04_ground_stage_bench_noprof.txt
