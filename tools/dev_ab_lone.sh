# dev: A/B of library variants on ONE box, one query alone only (five queries, registration of 20 jobs): dev_ab_lone.sh tag...
cd $GRAFT_REPO_ROOT
run() { echo -n "$1: "; GLOC3D_LIB_PATH=$PWD/gloc3d_amd/lib/$1 timeout -k 10 300 python3 tools/dev_lone_cold_sweep.py 2>/dev/null | tail -1; }
run libgloc3d.so
for t in "$@"; do run libgloc3d_$t.so; done
run libgloc3d.so
for t in "$@"; do run libgloc3d_$t.so; done
