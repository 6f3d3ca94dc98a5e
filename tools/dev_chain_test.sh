# dev: the chained launch's tests alone, then its time line and one query alone with and without it (same box)
set -e; O=gpurun_out; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_reg_gpu.py -x -q -m gpu -k "chained or stalls" > $O/chain_tests.log 2>&1 || { tail -40 $O/chain_tests.log; exit 1; }
tail -3 $O/chain_tests.log
if grep -q "core dump\|Memory access fault" $O/chain_tests.log; then exit 9; fi
timeout -k 10 300 python tools/dev_chain_trace.py > $O/chain_trace.txt 2>&1 || { tail -30 $O/chain_trace.txt; exit 1; }
if grep -q "core dump\|Memory access fault" $O/chain_trace.txt; then exit 9; fi
grep "^chain\|launch spans" $O/chain_trace.txt; tail -21 $O/chain_trace.txt
for v in 0 1 0 1; do
  if [ $v = 0 ]; then export GLOC3D_NN_NO_CHAIN=1; else unset GLOC3D_NN_NO_CHAIN; fi
  timeout -k 10 300 python bench.py --only-lone --no-cpu-baseline --steps 3 --warmup 1 --reps 1 > $O/chain_lone_$v.json 2> $O/chain_lone_$v.err || { tail -20 $O/chain_lone_$v.err; exit 1; }
  if grep -q "core dump\|Memory access fault" $O/chain_lone_$v.err; then exit 9; fi
  python - <<P
import json
d=json.loads([l for l in open("$O/chain_lone_$v.json") if l.startswith("{")][-1])
print("chain" if $v else "plain", "lone_query_ms", d.get("lone_query_ms"), "value", d.get("value"))
P
done
