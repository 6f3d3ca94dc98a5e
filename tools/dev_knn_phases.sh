# dev: time the fused kNN select+rerank kernel with phases compiled out (libgloc3d_skip1/2.so built with -DGLOC_KNN_DEV_SKIP)
cd /tmp; export TMPDIR=/tmp
for v in skip1 skip2; do
GLOC3D_LIB_PATH=$GRAFT_REPO_ROOT/gloc3d_amd/lib/libgloc3d_$v.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp_$v -o knn -- python3 $GRAFT_REPO_ROOT/tools/bench_knn.py > /dev/null 2>&1
python3 - "$v" $(find /tmp/kp_$v -name "*kernel_stats.csv") <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[2])):
    if 'select_rerank' in r['Name']: print(sys.argv[1], r['Calls'], round(float(r['AverageNs'])/1000, 2), 'us')
PY
done
