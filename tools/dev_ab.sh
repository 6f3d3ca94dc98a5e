# dev: A/B of library variants on ONE box (boxes differ by 2-3 %): dev_ab.sh [bench args --] tag...   (the shipped library first and last)
cd /tmp; python3 -c "import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT'); import bench; bench.build_views('/tmp/views.npz')"; cd $GRAFT_REPO_ROOT
EXTRA=""
if [[ " $* " == *" -- "* ]]; then EXTRA="${*%% -- *}"; set -- ${*##* -- }; fi
run() { echo "$1"; GLOC3D_LIB_PATH=$PWD/gloc3d_amd/lib/$1 python bench.py --views-cache /tmp/views.npz --steps 8 --warmup 2 --reps 2 --no-cpu-baseline --no-legs --min-success 0 $EXTRA 2>/dev/null | python tools/bench_line.py; }
run libgloc3d.so
for t in "$@"; do run libgloc3d_$t.so; done
run libgloc3d.so
