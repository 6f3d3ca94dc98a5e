python -m pytest tests/test_knn_gpu.py tests/test_knn_cfgE_gpu.py -x -q -m gpu 2>&1 | tail -3
python bench.py --gpus 2 --backend gloo --same-device --cfge-rows 200000 --steps 2 --warmup 1 --reps 1 --batch 5 --places 600 > gpurun_out/r4_rehearse2.json 2> gpurun_out/r4_rehearse2.err
echo "rc=$?"
tail -5 gpurun_out/r4_rehearse2.err
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r4_rehearse2.json') if x.startswith('{')]
d=json.loads(l[-1])
print(d['value'], d['n_gpus'], d.get('per_gpu_value'), d['config']['collectives'], d['config'].get('collectives_requested'))
print(json.dumps(d['sub_records'], indent=1)[:2500])
PY
