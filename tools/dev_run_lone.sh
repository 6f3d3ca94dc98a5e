#!/bin/bash
cd /tmp; python3 -c "import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT'); import bench; bench.build_views('/tmp/views.npz')"; cd $GRAFT_REPO_ROOT
for l in libgloc3d.so libgloc3d_wpb2.so libgloc3d_wpb4.so; do
echo "== $l"
GLOC3D_LIB_PATH=$PWD/gloc3d_amd/lib/$l python tools/dev_split_sweep.py 256,60000 0,0 2>&1 | grep -v amdgpu.ids | tail -3
done
