# dev: rocprofv3 kernel stats of a short bench run (per-kernel mean durations); usage: dev_kstats.sh tag [bench args]
set -e
tag=$1; shift
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_stats -o bench -- python3 $R/bench.py --no-cpu-baseline --no-legs --steps 5 --warmup 2 --reps 1 $* > $O/${tag}_under_rocprof.json 2> $O/${tag}_stats.err
find $O/${tag}_stats -name "*kernel_stats.csv" -exec cp {} $O/${tag}_kernel_stats.csv \;
rm -rf $O/${tag}_stats
grep "nn_compact\|solve_kernel\|ransac\|accum" $O/${tag}_kernel_stats.csv | sed 's/(gloc::reg::Job const.*)"/"/' | cut -c1-160
