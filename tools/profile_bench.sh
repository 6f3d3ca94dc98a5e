#!/bin/bash
# Kernel trace + PMC passes of bench.py on the GPU box.  Usage: tools/profile_bench.sh <tag> [bench args...]
# Writes gpurun_out/<tag>_{stats,pmc_fetch,pmc_write,pmc_sq}/ ; copy the summaries to profiles/.
set -e
tag=$1; shift
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
B="python3 $R/bench.py --no-cpu-baseline --no-legs $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_stats -o bench -- $B > $O/${tag}_under_rocprof.json 2> $O/${tag}_stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${tag}_pmc_fetch -o p -- $B > /dev/null 2> $O/${tag}_pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${tag}_pmc_write -o p -- $B > /dev/null 2> $O/${tag}_pmc_write.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $O/${tag}_pmc_sq -o p -- $B > /dev/null 2> $O/${tag}_pmc_sq.err
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/${tag}_pmc_l2 -o p -- $B > /dev/null 2> $O/${tag}_pmc_l2.err
cd $R
python3 tools/pmc_summary.py $O/${tag}_pmc_fetch $O/${tag}_pmc_write $O/${tag}_pmc_sq $O/${tag}_pmc_l2 --match gloc --json $O/${tag}_pmc_summary.json > $O/${tag}_pmc_summary.txt
find $O/${tag}_stats -name "*kernel_stats.csv" -exec cp {} $O/${tag}_kernel_stats.csv \;
rm -rf $O/${tag}_stats/*/*.db $O/${tag}_stats/*.db
# keep the merge small: drop the per-dispatch traces
find $O/${tag}_pmc_fetch $O/${tag}_pmc_write $O/${tag}_pmc_sq $O/${tag}_pmc_l2 $O/${tag}_stats -name "*kernel_trace.csv" -delete
find $O/${tag}_pmc_fetch $O/${tag}_pmc_write $O/${tag}_pmc_sq $O/${tag}_pmc_l2 -name "*counter_collection.csv" -delete
head -12 $O/${tag}_kernel_stats.csv
cat $O/${tag}_pmc_summary.txt | grep "nn_compact\|solve\|accum\|ransac_score"
tail -1 $O/${tag}_under_rocprof.json | python3 tools/bench_line.py
