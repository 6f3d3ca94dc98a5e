set -e
python -m pytest tests/test_reg_gpu.py -x -q -m gpu -k "split or bit_identical or multi or pipeline" > gpurun_out/r4_t4.log 2>&1 || (tail -30 gpurun_out/r4_t4.log; exit 1)
tail -2 gpurun_out/r4_t4.log
python tools/dev_split_sweep.py 0,0,24,1 0,0,24,2 0,0,24,4 0,0,24,8 0,0,40,2 128,60000,24,4 256,60000,24,4 128,90000,24,4 128,60000,24,2 128,60000,24,8 > gpurun_out/r4_split5.log 2>&1
cat gpurun_out/r4_split5.log
python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline > gpurun_out/r4_bench1.json 2> gpurun_out/r4_bench1.err
python tools/bench_line.py < gpurun_out/r4_bench1.json
