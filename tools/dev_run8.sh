cd /tmp; python3 -c "import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT'); import bench; bench.build_views('/tmp/views.npz')"; cd $GRAFT_REPO_ROOT
for p in 4 8 16 31; do echo "parts $p"; GLOC3D_RANSAC_PARTS=$p python bench.py --views-cache /tmp/views.npz --steps 6 --warmup 2 --reps 1 --no-cpu-baseline --no-legs --ransac-confidence 0 2>/dev/null | python tools/bench_line.py; done
