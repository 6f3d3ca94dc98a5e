#!/bin/bash
# dev: job-group size vs throughput and L2-miss traffic of the culled 1-NN kernel
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
python3 -c "import sys; sys.path.insert(0, '$R'); import bench; bench.build_views('/tmp/views.npz')"
for G in 8 16 24 40 64; do
  python3 $R/bench.py --no-cpu-baseline --views-cache /tmp/views.npz --steps 3 --warmup 1 --reps 2 --nn-job-group $G 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('G $G', round(d['value'],1), 'q/s  nn launch ms', round(d['roofline']['launch_ms'],3))"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/grp_$G -o p -- python3 $R/bench.py --no-cpu-baseline --views-cache /tmp/views.npz --steps 2 --warmup 1 --reps 1 --nn-job-group $G > /dev/null 2>&1
  python3 $R/tools/pmc_summary.py $O/grp_$G --match nn_compact_kernel | cut -c1-220
  rm -rf $O/grp_$G
done
