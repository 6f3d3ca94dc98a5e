set -e
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
python3 -c "import sys; sys.path.insert(0, '$R'); import bench; bench.build_views('/tmp/views.npz')"
B="python3 $R/bench.py --no-cpu-baseline --no-legs --views-cache /tmp/views.npz --steps 2 --warmup 1 --reps 1"
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS --output-format csv -d $O/lds1 -o p -- $B > /dev/null 2> $O/lds1.err
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES --output-format csv -d $O/lds2 -o p -- $B > /dev/null 2> $O/lds2.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_ATOMIC SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_LEVEL_LDS SQ_LDS_UNALIGNED_STALL --output-format csv -d $O/lds3 -o p -- $B > /dev/null 2> $O/lds3.err
cd $R
python3 tools/pmc_summary.py $O/lds1 $O/lds2 $O/lds3 --match nn_compact --json $O/lds_summary.json > $O/lds_summary.txt
find $O/lds1 $O/lds2 $O/lds3 -name "*.csv" -size +1M -delete
cat $O/lds_summary.txt
