#!/bin/bash
# dev: registration tests + a short bench
set -e
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_reg_gpu.py tests/test_reg_variant_gpu.py tests/test_headline_gpu.py -x -q -m gpu > $O/reg_tests.log 2>&1 || { tail -40 $O/reg_tests.log; exit 1; }
tail -n 2 $O/reg_tests.log
cd /tmp; python3 -c "import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT'); import bench; bench.build_views('/tmp/views.npz')"; cd $GRAFT_REPO_ROOT
python bench.py --views-cache /tmp/views.npz --steps 10 --warmup 3 --reps 3 --no-cpu-baseline 2>/dev/null > $O/quick_bench.json
python tools/bench_line.py < $O/quick_bench.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/quick_bench.json').read().strip().splitlines()[-1])
print("lone", d["sub_records"]["cfgC_lone_query"]["ms_per_query"], d["sub_records"]["cfgC_lone_query"]["nn_launch_ms"])
print("roofline", d["roofline"].get("launch_ms"), d["roofline"].get("cold_launch_ms"), [k for k in d["roofline"].keys()][:30])
PY
