set -e; O=gpurun_out; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_knn_gpu.py tests/test_fuzz_gpu.py tests/test_knn_cfgE_gpu.py -x -q -m gpu > $O/r06_knn_tests.log 2>&1 || { tail -30 $O/r06_knn_tests.log; exit 1; }; tail -1 $O/r06_knn_tests.log
bash tools/prof_knn.sh r06_knn_cfgB --algo 2 --reps 200 --kind 1
bash tools/dev_knn_pmc.sh r06_knn_cfgB dist_bf16x3 --algo 2 --kind 1 > $O/r06_knn_cfgB_pmc.log 2>&1; cp $O/r06_knn_cfgB_pmc.json profiles/ 2>/dev/null || true
bash tools/dev_knn_pmc.sh r06_knn_cfgB_select select_rerank --algo 2 --kind 1 > $O/r06_knn_cfgB_select_pmc.log 2>&1; cp $O/r06_knn_cfgB_select_pmc.json profiles/ 2>/dev/null || true
bash tools/prof_knn.sh r06_knn_shard125k --algo 2 --reps 30 --n 125000 --kind 1
python tools/dev_nn_trace3.py 20 25 > $O/r06_nn_trace_500jobs.txt 2>&1
python tools/dev_nn_trace3.py 0 8 > $O/r06_nn_trace_cold_160jobs.txt 2>&1


python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
