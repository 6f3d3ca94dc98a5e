python -m pytest tests/test_reg_gpu.py tests/test_headline_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu 2>&1 | tail -4
cd /tmp; python3 -c "import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT'); import bench; bench.build_views('/tmp/views.npz')"; cd $GRAFT_REPO_ROOT
python bench.py --views-cache /tmp/views.npz --steps 20 --warmup 3 --reps 2 --no-cpu-baseline > gpurun_out/r4_bench2.json 2> gpurun_out/r4_bench2.err
echo "rc=$?"; tail -3 gpurun_out/r4_bench2.err
python - <<'PY'
import json
d=json.loads([x for x in open('gpurun_out/r4_bench2.json') if x.startswith('{')][-1])
print(d['value'], d['accuracy']['success_rate'], d['accuracy']['located_but_wrong'][:3], d['accuracy']['rmse_max_m_of_successes'])
for k,v in d['legs'].items():
    a=v.get('accuracy') or {}
    print(k, round(v.get('value',0),1), 'success', a.get('success_rate'), 'located', a.get('located'), 'wrong', [(w['err_pos_m'], w['err_rot_deg'], w['rmse_m'], w['place']) for w in a.get('located_but_wrong',[])])
print(d['sub_records']['cfgC_lone_query'])
PY
