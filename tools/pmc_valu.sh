# dev: VALU / SALU / LDS instructions per wave of the culled 1-NN kernel (one PMC pass over a short bench)
set -e
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
python3 -c "import sys; sys.path.insert(0, '$R'); import bench; bench.build_views('/tmp/views.npz')"
B="python3 $R/bench.py --no-cpu-baseline --no-legs --views-cache /tmp/views.npz --steps 2 --warmup 1 --reps 1"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $O/valu1 -o p -- $B > /dev/null 2> $O/valu1.err
cd $R
python3 tools/pmc_summary.py $O/valu1 --match nn_compact --json $O/valu_summary.json > $O/valu_summary.txt
find $O/valu1 -name "*.csv" -size +1M -delete
python3 - <<'PY'
import json, os
d = json.load(open(os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/valu_summary.json"))
for k, c in d.items():
    w = c["SQ_WAVES"]
    print(k[:48], "us", round(c["mean_us_under_pmc"], 1), "VALU/wave", round(c["SQ_INSTS_VALU"] / w), "SALU", round(c["SQ_INSTS_SALU"] / w), "LDS", round(c["SQ_INSTS_LDS"] / w),
          "valu busy", round(4 * c["SQ_ACTIVE_INST_VALU"] / 1024 / (c["SQ_BUSY_CYCLES"] / 32), 3))
PY
