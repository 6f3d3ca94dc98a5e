# dev: the chained launch's tests alone, then its time line and one query alone with and without it (same box)
set -e; O=gpurun_out; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_reg_gpu.py -x -q -m gpu -k "chained or stalls" > $O/chain_tests.log 2>&1 || { tail -40 $O/chain_tests.log; exit 1; }
tail -3 $O/chain_tests.log
if grep -q "core dump\|Memory access fault" $O/chain_tests.log; then exit 9; fi
timeout -k 10 300 python tools/dev_chain_trace.py > $O/chain_trace.txt 2>&1 || { tail -30 $O/chain_trace.txt; exit 1; }
if grep -q "core dump\|Memory access fault" $O/chain_trace.txt; then exit 9; fi
grep "^chain\|launch spans" $O/chain_trace.txt; tail -4 $O/chain_trace.txt
timeout -k 10 300 python tools/dev_lone_cold_sweep.py > $O/lone_cold_sweep2.txt 2>&1; head -3 $O/lone_cold_sweep2.txt | tail -2
