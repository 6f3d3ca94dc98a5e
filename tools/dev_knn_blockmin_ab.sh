# dev (round 6): the large window's selection from the coarse kernel's block minima against the slices, per kernel under rocprof, same box
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
for rep in 1 2; do
for v in 0 1; do
  if [ $v = 1 ]; then export GLOC3D_KNN_NO_BLOCKMIN=1; else unset GLOC3D_KNN_NO_BLOCKMIN; fi
  echo "== NO_BLOCKMIN=$v: $(python3 $R/tools/bench_knn.py --kind 1 --n 125000 --reps 150 2>/dev/null | cut -c1-70)"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/bm_$v -o knn -- python3 $R/tools/bench_knn.py --kind 1 --n 125000 --reps 60 > /dev/null 2>&1
  find $O/bm_$v -name "*kernel_stats.csv" -exec cat {} \; | python3 -c "
import sys,csv
for r in csv.reader(sys.stdin):
    if r[0]=='Name' or 'fill' in r[0] or 'norms' in r[0] or 'mirror' in r[0] or 'rocclr' in r[0]: continue
    print('   %-52s calls %4s mean %8.2f us'%(r[0].split('(')[0][-52:], r[1], float(r[3])/1e3))"
  rm -rf $O/bm_$v
done
done
