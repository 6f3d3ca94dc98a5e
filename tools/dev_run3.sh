cd /tmp; python3 -c "import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT'); import bench; bench.build_views('/tmp/views.npz')"; cd $GRAFT_REPO_ROOT
run() { echo "== $@"; python bench.py --views-cache /tmp/views.npz --steps 8 --warmup 2 --reps 2 --no-cpu-baseline --no-legs $@ 2>/dev/null | python tools/bench_line.py; }
python -m pytest tests/test_reg_gpu.py -x -q -m gpu -k "split or bit_identical or multi or golden or lattice" 2>&1 | tail -2
run --nn-split-helpers 0
run --nn-split-helpers 16
run --nn-split-helpers 0
python tools/dev_split_sweep.py 0,0,24,1 256,60000,24,4
