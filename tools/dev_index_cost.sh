cd /tmp; export TMPDIR=/tmp
python3 -c "import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT'); import bench; bench.build_views('/tmp/views.npz')"
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_reg_gpu.py tests/test_pipeline_gpu.py -m gpu -x -q 2>&1 | tail -2
python bench.py --views-cache /tmp/views.npz --steps 8 --warmup 1 --reps 2 --no-cpu-baseline 2>/dev/null | python tools/bench_line.py
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o b -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-lone-query --views-cache /tmp/views.npz --steps 2 --warmup 1 --reps 1 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/ks/**/*kernel_stats.csv', recursive=True)[0]
tot = 0
for r in csv.DictReader(open(f)):
    n = r['Name']
    if any(k in n for k in ('rocprim', 'morton', 'pack_bbox', 'gather_sorted', 'chunk_boxes', 'subblock', 'super_boxes', 'group_extent', 'scan_header', 'scan_variant')):
        tot += float(r['TotalDurationNs'])
        print(n[:90].replace('rocprim::ROCPRIM_400200_NS::detail::', ''), r['Calls'], round(float(r['AverageNs']) / 1000, 1))
print('index build GPU time total ms', tot / 1e6)
PY
