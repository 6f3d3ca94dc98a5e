#!/bin/bash
# dev: PMC passes over tools/bench_knn.py.  Usage: tools/dev_knn_pmc.sh <tag> <match> [bench_knn args...]
tag=$1; match=$2; shift; shift
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
B="python3 $R/tools/bench_knn.py --reps 5 $*"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $O/${tag}_a -o p -- $B > /dev/null 2> $O/${tag}_a.err
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM --output-format csv -d $O/${tag}_b -o p -- $B > /dev/null 2> $O/${tag}_b.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${tag}_c -o p -- $B > /dev/null 2> $O/${tag}_c.err
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --output-format csv -d $O/${tag}_d -o p -- $B > /dev/null 2> $O/${tag}_d.err
cd $R
python3 tools/pmc_summary.py $O/${tag}_a $O/${tag}_b $O/${tag}_c $O/${tag}_d --match $match --json $O/${tag}_pmc.json
find $O/${tag}_a $O/${tag}_b $O/${tag}_c $O/${tag}_d -name "*.csv" -delete; find $O -name "*.db" -delete
tail -3 $O/${tag}_a.err $O/${tag}_b.err | cut -c1-200
