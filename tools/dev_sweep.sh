# dev: A/B of library variants (gloc3d_amd/lib/libgloc3d_<tag>.so) on a short bench; usage: dev_sweep.sh tag...
cd /tmp; python3 -c "import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT'); import bench; bench.build_views('/tmp/views.npz')"; cd $GRAFT_REPO_ROOT
run() { echo "$@"; GLOC3D_LIB_PATH=$PWD/gloc3d_amd/lib/$1 python bench.py --views-cache /tmp/views.npz --steps 8 --warmup 2 --reps 2 --no-cpu-baseline --no-legs --nn-split-helpers 0 ${@:2} 2>/dev/null | python tools/bench_line.py; }
run libgloc3d.so
for t in "$@"; do run libgloc3d_$t.so; done
run libgloc3d.so
