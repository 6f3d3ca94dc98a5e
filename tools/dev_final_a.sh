# dev: everything the round's committed evidence comes from, in the order that keeps it consistent: tests, PMC passes of
# the bench (-> profiles/r06_pmc_traffic_nn_compact.json, which bench.py reads), THEN the bench line, then stage profiles
set -e
O=gpurun_out; mkdir -p $O


bash tools/profile_bench.sh r06_bench_n1 --steps 5 --warmup 2 --reps 1 > $O/r06_profile_bench.log 2>&1 || { tail -20 $O/r06_profile_bench.log; exit 1; }
python tools/pmc_traffic_json.py $O/r06_bench_n1_pmc_summary.json $O/r06_bench_n1_under_rocprof.json 6 > profiles/r06_pmc_traffic_nn_compact.json
cp profiles/r06_pmc_traffic_nn_compact.json $O/r06_pmc_traffic_nn_compact.json
python bench.py --steps 20 --warmup 5 > $O/r06_bench_n1.json 2> $O/r06_bench_n1.err
python tools/bench_line.py < $O/r06_bench_n1.json
timeout -k 10 300 python tools/fuzz_reg.py 3000 55 > $O/r06_reg_fuzz.txt 2>&1; tail -2 $O/r06_reg_fuzz.txt
timeout -k 10 300 python tools/fuzz_knn.py 300 12 > $O/r06_knn_fuzz.txt 2>&1; tail -2 $O/r06_knn_fuzz.txt
