# dev: everything the round's committed evidence comes from, in the order that keeps it consistent: tests, PMC passes of
# the bench (-> profiles/r04_pmc_traffic_nn_compact.json, which bench.py reads), THEN the bench line, then stage profiles
set -e
O=gpurun_out; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1 || { tail -40 $O/gpu_tests.log; exit 1; }
tail -n 2 $O/gpu_tests.log
bash tools/profile_bench.sh r04_bench_n1 --steps 5 --warmup 2 --reps 1 > $O/r04_profile_bench.log 2>&1 || { tail -20 $O/r04_profile_bench.log; exit 1; }
python tools/pmc_traffic_json.py $O/r04_bench_n1_pmc_summary.json $O/r04_bench_n1_under_rocprof.json 4 > profiles/r04_pmc_traffic_nn_compact.json
cp profiles/r04_pmc_traffic_nn_compact.json $O/r04_pmc_traffic_nn_compact.json
python bench.py --steps 20 --warmup 5 > $O/r04_bench_n1.json 2> $O/r04_bench_n1.err
python tools/bench_line.py < $O/r04_bench_n1.json
bash tools/prof_knn.sh r04_knn_cfgB --algo 2 --reps 200 --kind 1
bash tools/prof_knn.sh r04_knn_shard125k --algo 2 --reps 30 --n 125000 --kind 1
python tools/dev_nn_trace3.py 20 25 > $O/r04_nn_trace_500jobs.txt 2>&1
python tools/dev_nn_trace3.py 0 8 > $O/r04_nn_trace_cold_160jobs.txt 2>&1
python tools/dev_nn_timeline.py 256 60000 > $O/r04_nn_timeline_lone_query_planned.txt 2>&1
python tools/dev_nn_timeline.py 0 0 > $O/r04_nn_timeline_lone_query_unplanned.txt 2>&1
python tools/dev_split_sweep.py 0,0 256,60000 > $O/r04_lone_query_plans.txt 2>&1
cat $O/r04_lone_query_plans.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
