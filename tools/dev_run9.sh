set -e
python -m pytest tests/test_ground_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu 2>&1 | tail -3
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04_ground_stats -o g -- python3 $GRAFT_REPO_ROOT/tools/bench_ground.py > $GRAFT_REPO_ROOT/gpurun_out/r04_ground_stage_bench.txt 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/r04_ground_stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/r04_ground_stage_kernel_stats.csv \;
rm -rf gpurun_out/r04_ground_stats
cat gpurun_out/r04_ground_stage_bench.txt | tail -8
cut -d, -f1-4 gpurun_out/r04_ground_stage_kernel_stats.csv | cut -c1-120 | head -24
