#!/bin/bash
# dev: kernel stats of the all-hypotheses RANSAC configuration
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
python3 -c "import sys; sys.path.insert(0, '$R'); import bench; bench.build_views('/tmp/views.npz')"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ransac_stats -o b -- python3 $R/bench.py --no-cpu-baseline --no-legs --views-cache /tmp/views.npz --ransac-confidence 0 --steps 4 --warmup 1 --reps 1 > /dev/null 2> $O/ransac_stats.err
find $O/ransac_stats -name "*kernel_stats.csv" -exec cp {} $O/ransac_kernel_stats.csv \;
rm -rf $O/ransac_stats
cut -d, -f1-4 $O/ransac_kernel_stats.csv | cut -c1-60,140-400 | head -14
