# dev (round 6): how many groups a cold pass gives up, and the two kernels' durations under rocprof, per threshold
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
for t in ${THRESHES:-16 32 64}; do
  echo "== thresh $t"
  GLOC3D_NN_HEAVY_DEBUG=1 python3 $R/bench.py --no-legs --no-cpu-baseline --steps 3 --warmup 1 --reps 1 --nn-heavy-thresh $t 2>&1 >/dev/null | grep "given up" | head -3
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/hp_$t -o bench -- python3 $R/bench.py --no-cpu-baseline --no-legs --steps 5 --warmup 2 --reps 1 --nn-heavy-thresh $t > /dev/null 2> /dev/null
  find $O/hp_$t -name "*kernel_stats.csv" -exec grep "nn_compact" {} \; | python3 -c "
import sys,csv
for r in csv.reader(sys.stdin):
    print('   %-60s calls %4s mean %9.1f us min %9.1f max %9.1f'%(r[0].split('(')[0][-60:], r[1], float(r[3])/1e3, float(r[5])/1e3, float(r[6])/1e3))"
  rm -rf $O/hp_$t
done
