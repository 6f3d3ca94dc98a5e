#!/bin/bash
# kernel stats of the kNN stage bench.  Usage: tools/prof_knn.sh <tag> [bench_knn args]
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; tag=$1; shift
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag} -o knn -- python3 $R/tools/bench_knn.py "$@" > $O/${tag}.txt 2>&1
find $O/${tag} -name "*kernel_stats.csv" -exec cp {} $O/${tag}_kernel_stats.csv \;
rm -rf $O/${tag}
tail -1 $O/${tag}.txt
cut -d, -f1-4 $O/${tag}_kernel_stats.csv | cut -c1-150 | head -14
