#!/bin/bash
for l in libgloc3d.so libgloc3d_kp3.so libgloc3d_kp6.so libgloc3d_kp8.so; do
echo "== $l"
GLOC3D_LIB_PATH=$GRAFT_REPO_ROOT/gloc3d_amd/lib/$l bash tools/dev_kstats.sh cold_$l 2>&1 | grep "nn_compact" | awk -F'",' '{print $2}' | cut -c1-60
done
