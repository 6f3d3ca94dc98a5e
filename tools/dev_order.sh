# dev: adaptive launch order (work of the previous pass) against the scans' own extent order
cd /tmp; python3 -c "import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT'); import bench; bench.build_views('/tmp/views.npz')"; cd $GRAFT_REPO_ROOT
for e in "A=1" "GLOC3D_NN_FIXED_ORDER=1"; do for a in "--batch 1 --steps 16" "--batch 4 --steps 8" "--batch 25 --steps 6"; do echo "$e $a"; env $e python bench.py --views-cache /tmp/views.npz --warmup 1 --reps 1 --no-cpu-baseline --no-lone-query $a 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1),'q/s; nn launch ms',round(r['launch_ms'],3), d['stage_ms_per_step_rank0'] if 'stage_ms_per_step_rank0' in d else '')"; done; done
