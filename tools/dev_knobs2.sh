# dev: the shipped library under different bench options on ONE box: dev_knobs2.sh "opts A" "opts B" ...   (no options first and last)
cd /tmp; python3 -c "import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT'); import bench; bench.build_views('/tmp/views.npz')"; cd $GRAFT_REPO_ROOT
run() { echo "[$1]"; python bench.py --views-cache /tmp/views.npz --steps 8 --warmup 2 --reps 2 --no-cpu-baseline --no-legs --min-success 0 $1 2>/dev/null | python tools/bench_line.py; }
run ""
for o in "$@"; do run "$o"; done
run ""
