cd /tmp; python3 -c "import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT'); import bench; bench.build_views('/tmp/views.npz')"; cd $GRAFT_REPO_ROOT
python tools/dev_split_sweep.py 0,0 0,0,24,1 256,60000 128,60000 256,50000 256,80000 > gpurun_out/r4_split6.log 2>&1
cat gpurun_out/r4_split6.log
python - <<'PY'
import subprocess,sys
for subs in (1,2,4,8):
    pass
PY
