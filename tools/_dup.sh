python3 -c "import bench; bench.build_views('/tmp/views.npz')"
for v in base ret1 ret6 ret2 ret4; do
  if [ $v = base ]; then unset GLOC3D_LIB_PATH; else export GLOC3D_LIB_PATH=$PWD/gloc3d_amd/lib/libgloc3d_$v.so; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-legs --min-success 0 --steps 6 --warmup 2 --reps 2 --views-cache /tmp/views.npz > gpurun_out/b_$v.json 2> gpurun_out/b_$v.err
  tail -1 gpurun_out/b_$v.json | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), round(d['stage_ms_per_step_rank0']['nn'],2), round(d['roofline']['launch_ms'],3), d['accuracy']['success_rate'])"
done
