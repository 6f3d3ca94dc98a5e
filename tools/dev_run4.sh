cd /tmp; python3 -c "import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT'); import bench; bench.build_views('/tmp/views.npz')"; cd $GRAFT_REPO_ROOT
run() { echo "== $@"; python bench.py --views-cache /tmp/views.npz --steps 8 --warmup 2 --reps 2 --no-cpu-baseline --no-legs --nn-split-helpers 0 $@ 2>/dev/null | python tools/bench_line.py; }
run
run --nn-job-group 16
run --nn-job-group 32
run --nn-job-group 48
run --nn-job-group 8
run --nn-src-per-lane 4
run --nn-src-per-lane 1
bash tools/dev_sweep.sh q192 thin16
