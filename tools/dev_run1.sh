set -e
mkdir -p gpurun_out
python -m pytest tests/test_reg_gpu.py -x -q -m gpu -k "pipeline or bit_identical or first_success" > gpurun_out/r4_t1.log 2>&1 || (tail -30 gpurun_out/r4_t1.log; exit 1)
tail -3 gpurun_out/r4_t1.log
python tools/dev_nn_trace3.py 20 8 > gpurun_out/r4_trace8.log 2>&1 || (tail -30 gpurun_out/r4_trace8.log; exit 1)
python tools/dev_nn_trace3.py 20 1 > gpurun_out/r4_trace1.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r4_bench0.json 2> gpurun_out/r4_bench0.err
python tools/bench_line.py < gpurun_out/r4_bench0.json
