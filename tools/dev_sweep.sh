cd /tmp; python3 -c "import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT'); import bench; bench.build_views('/tmp/views.npz')"; cd $GRAFT_REPO_ROOT
run() { echo "$@"; GLOC3D_LIB_PATH=$PWD/gloc3d_amd/lib/$1 python bench.py --views-cache /tmp/views.npz --steps 4 --warmup 1 --reps 1 --no-cpu-baseline --no-lone-query ${@:2} 2>/dev/null | python tools/bench_line.py; }
run libgloc3d.so
run libgloc3d_wpb4.so
run libgloc3d_wpb2.so
run libgloc3d_eu5.so
run libgloc3d.so --nn-src-per-lane 1
run libgloc3d.so --nn-src-per-lane 4
run libgloc3d.so --nn-job-group 8
run libgloc3d.so --nn-job-group 16
run libgloc3d.so --nn-job-group 40
run libgloc3d.so --nn-job-group 64
