# dev: the shipped library under different ENVIRONMENT settings on ONE box: dev_env_ab.sh "VAR=a" "VAR=b" ...   (each run: short bench)
cd /tmp; python3 -c "import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT'); import bench; bench.build_views('/tmp/views.npz')"; cd $GRAFT_REPO_ROOT
for e in "$@"; do echo "[$e]"; env $e python bench.py --views-cache /tmp/views.npz --steps 8 --warmup 2 --reps 2 --no-cpu-baseline --no-legs 2>/dev/null | python tools/bench_line.py; done
