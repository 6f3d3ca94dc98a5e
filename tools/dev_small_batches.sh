cd /tmp; python3 -c "import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT'); import bench; bench.build_views('/tmp/views.npz')"; cd $GRAFT_REPO_ROOT
for a in "--batch 1" "--batch 1 --nn-src-per-lane 1" "--batch 2" "--batch 2 --nn-src-per-lane 1" "--batch 4"; do echo "$a"; python bench.py --views-cache /tmp/views.npz --steps 8 --warmup 1 --reps 1 --no-cpu-baseline --no-lone-query $a 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1),'q/s; nn launch ms',round(r['launch_ms'],3),'jobs',r['jobs_per_launch'])"; done
