set -e; O=gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_reg_gpu.py tests/test_headline_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu > $O/reg_tests.log 2>&1 || { tail -40 $O/reg_tests.log; exit 1; }
tail -3 $O/reg_tests.log
if grep -q "core dump\|Memory access fault" $O/reg_tests.log; then exit 9; fi
for c in 4096 1024 2048 512; do
export GLOC3D_RANSAC_CHUNK=$c
echo chunk $c
timeout -k 10 300 python tools/dev_lone_cold_sweep.py > $O/lone_cold_sweep2.txt 2>&1; head -2 $O/lone_cold_sweep2.txt | tail -1
done
bash tools/dev_lone_timeline.sh > $O/lone_timeline2.txt 2>&1; tail -24 $O/lone_timeline2.txt | grep ransac
