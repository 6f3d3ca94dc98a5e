# dev: everything the round's committed evidence comes from, in the order that keeps it consistent: tests, PMC passes of
# the bench (-> profiles/r05_pmc_traffic_nn_compact.json, which bench.py reads), THEN the bench line, then stage profiles
set -e
O=gpurun_out; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1 || { tail -40 $O/gpu_tests.log; exit 1; }
tail -n 2 $O/gpu_tests.log
bash tools/profile_bench.sh r05_bench_n1 --steps 5 --warmup 2 --reps 1 > $O/r05_profile_bench.log 2>&1 || { tail -20 $O/r05_profile_bench.log; exit 1; }
python tools/pmc_traffic_json.py $O/r05_bench_n1_pmc_summary.json $O/r05_bench_n1_under_rocprof.json 5 > profiles/r05_pmc_traffic_nn_compact.json
cp profiles/r05_pmc_traffic_nn_compact.json $O/r05_pmc_traffic_nn_compact.json
python bench.py --steps 20 --warmup 5 > $O/r05_bench_n1.json 2> $O/r05_bench_n1.err
python tools/bench_line.py < $O/r05_bench_n1.json
timeout -k 10 300 python tools/fuzz_reg.py 3000 55 > $O/r05_reg_fuzz.txt 2>&1; tail -2 $O/r05_reg_fuzz.txt
timeout -k 10 300 python tools/fuzz_knn.py 300 12 > $O/r05_knn_fuzz.txt 2>&1; tail -2 $O/r05_knn_fuzz.txt
