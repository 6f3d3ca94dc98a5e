# dev: the 32x32x2 MFMA tiles against the shipped plan on cfg B (64 x 10 000 x 4096) and a 125 k shard
run() { echo "== $1 BK=$2"; GLOC3D_MFMA_T32=$1 GLOC3D_MFMA_BK=$2 python tools/bench_knn.py --algo 2 --reps 300 2>&1 | grep "kNN"; }
echo "== shipped"; python tools/bench_knn.py --algo 2 --reps 300 2>&1 | grep kNN
run 1,4 64
run 1,2 64
run 1,8 64
run 2,4 64
run 2,8 64
run 1,4 32
run 2,4 32
run 2,8 32
echo "== shard 125k shipped"; python tools/bench_knn.py --algo 2 --reps 50 --n 125000 2>&1 | grep kNN
echo "== shard 125k t32 1,1"; GLOC3D_MFMA_T32=1,1 python tools/bench_knn.py --algo 2 --reps 50 --n 125000 2>&1 | grep kNN
echo "== shard 125k t32 2,1 bk32"; GLOC3D_MFMA_T32=2,1 GLOC3D_MFMA_BK=32 python tools/bench_knn.py --algo 2 --reps 50 --n 125000 2>&1 | grep kNN
GLOC3D_MFMA_T32=1,4 python -m pytest tests/test_knn_gpu.py -x -q -m gpu 2>&1 | tail -3
