# dev: the headline (500 jobs per launch) with the ICP passes chained (GLOC3D_NN_CHAIN_FORCE=1) against launch by launch, same box
set -e; O=gpurun_out; mkdir -p $O
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export GLOC3D_NN_CHAIN_FORCE=1; else unset GLOC3D_NN_CHAIN_FORCE; fi
  timeout -k 10 400 python bench.py --no-legs --no-cpu-baseline --steps 10 --warmup 2 --reps 3 > $O/chain500_$v.json 2> $O/chain500_$v.err || { tail -20 $O/chain500_$v.err; exit 1; }
  if grep -q "core dump\|Memory access fault" $O/chain500_$v.err; then exit 9; fi
  python - <<P
import json
d=json.loads([l for l in open("$O/chain500_$v.json") if l.startswith("{")][-1])
print("chain" if $v else "plain", "value %.1f ms_per_step %.2f nn %.2f solve %.2f success %s wrong %s" % (d["value"], d["ms_per_step"], d["stage_ms_per_step_rank0"]["nn"], d["stage_ms_per_step_rank0"]["solve"], d.get("success_rate"), d.get("located_but_wrong")))
P
done
