cd $GRAFT_REPO_ROOT
for jg in 24 16 32 48 24; do
  python3 bench.py --only-lone --no-cpu-baseline --steps 20 --warmup 2 --reps 3 --nn-job-group $jg 2>/dev/null | tail -1 | python3 -c "
import json,sys
f=json.loads(sys.stdin.read())
print('job_group $jg: %.1f q/s nn %.2f ms/step warm %.3f cold %.3f'%(f['value'],f['nn_ms_per_step'],f['roofline']['launch_ms'],f['roofline']['cold_launch_ms']))"
done
