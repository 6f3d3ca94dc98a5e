# dev: A/B of library variants on ONE box (boxes differ by 2-3 %): dev_ab.sh [bench args --] tag...   (the shipped library first and last)
# a variant = gloc3d_amd/lib/libgloc3d_<tag>.so (tools/build_variant.py tag -D..., or a copy of an earlier build)
cd $GRAFT_REPO_ROOT
EXTRA=""
if [[ " $* " == *" -- "* ]]; then EXTRA="${*%% -- *}"; set -- ${*##* -- }; fi
run() { echo -n "$1: "; GLOC3D_LIB_PATH=$PWD/gloc3d_amd/lib/$1 python3 bench.py --steps 20 --warmup 2 --reps 3 --no-cpu-baseline --only-lone --min-success 0 $EXTRA 2>/dev/null | tail -1 | python3 -c "
import json,sys
f=json.loads(sys.stdin.read())
print('%.1f q/s  nn %.2f ms/step  warm %.3f cold %.3f ms  | one query alone %.3f ms (pass %.4f cold %.3f)'%(f['value'],f['nn_ms_per_step'],f['roofline']['launch_ms'],f['roofline']['cold_launch_ms'],f['lone_query_ms'],f['lone_query_nn_launch_ms'],f['lone_query_nn_cold_launch_ms']))"; }
run libgloc3d.so
for t in "$@"; do run libgloc3d_$t.so; done
run libgloc3d.so
for t in "$@"; do run libgloc3d_$t.so; done
