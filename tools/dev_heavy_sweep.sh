# dev (round 6): the cold pass's give-up threshold (GLOC_REG_OPT_NN_HEAVY_THRESH) on ONE box: 500 jobs per launch and one query alone
cd $GRAFT_REPO_ROOT
for t in 0 32 64 96 160 0; do
  python3 bench.py --only-lone --no-cpu-baseline --steps 20 --warmup 2 --reps 3 --nn-heavy-thresh $t 2>/dev/null | tail -1 | python3 -c "
import json,sys
f=json.loads(sys.stdin.read())
print('thresh $t: %.1f q/s nn %.2f ms/step warm %.3f cold %.3f | lone %.3f ms (pass %.4f cold %.3f)'%(f['value'],f['nn_ms_per_step'],f['roofline']['launch_ms'],f['roofline']['cold_launch_ms'],f['lone_query_ms'],f['lone_query_nn_launch_ms'],f['lone_query_nn_cold_launch_ms']))"
done
