cd /tmp; python3 -c "import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT'); import bench; bench.build_views('/tmp/views.npz')"; cd $GRAFT_REPO_ROOT
for t in 0.03 0.025 0.05; do
echo "== max_final_step $t"
GLOC3D_MAX_FINAL_STEP=$t python bench.py --views-cache /tmp/views.npz --steps 8 --warmup 2 --reps 1 --no-cpu-baseline > gpurun_out/r4_gate_$t.json 2>/dev/null
python - <<PY
import json
d=json.loads([x for x in open('gpurun_out/r4_gate_$t.json') if x.startswith('{')][-1])
print(round(d['value'],1), d['accuracy']['success_rate'], len(d['accuracy']['located_but_wrong']), d['selected_candidate_rank_histogram'])
for k,v in d['legs'].items():
    a=v.get('accuracy') or {}
    print('  ',k, round(v.get('value',0),1), 'success', a.get('success_rate'), 'located', a.get('located'), 'wrong', [(w['err_pos_m'], w['err_rot_deg']) for w in a.get('located_but_wrong',[])][:8])
PY
done
