# dev: SQ and L2 counters of one query's registration kernels (12 repetitions), chained and (GLOC3D_NN_NO_CHAIN=1) launch by launch
set -e
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
summ() {
python3 - "$1" <<P
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); seen = set()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][:64]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (k, r["Dispatch_Id"])
    if key not in seen:
        seen.add(key); cnt[k] += 1
for k, c in agg.items():
    if "nn_c" not in k: continue
    if "SQ_WAVES" in c:
        w = c["SQ_WAVES"]
        print(k, "dispatches", cnt[k], "waves/dispatch %.0f" % (w / cnt[k]), "VALU/wave %.0f SALU/wave %.0f LDS/wave %.0f VMEM_RD/wave %.1f" % (c["SQ_INSTS_VALU"] / w, c["SQ_INSTS_SALU"] / w, c["SQ_INSTS_LDS"] / w, c["SQ_INSTS_VMEM_RD"] / w),
              "VALU busy %.3f" % (c["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / max(c["SQ_BUSY_CYCLES"] / 32, 1)), "busy cycles/dispatch %.0f" % (c["SQ_BUSY_CYCLES"] / 32 / cnt[k]))
    else:
        h, m = c["TCC_HIT_sum"], c["TCC_MISS_sum"]
        print(k, "dispatches", cnt[k], "L2 hit %.3f" % (h / max(h + m, 1)), "requests/dispatch %.0f misses/dispatch %.0f" % ((h + m) / cnt[k], m / cnt[k]))
P
}
for mode in chain plain; do
  if [ $mode = plain ]; then export GLOC3D_NN_NO_CHAIN=1; fi
  echo "== $mode"
  rm -rf $O/chain_pmc
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $O/chain_pmc -o p -- python3 $R/tools/dev_lone_timeline.py 2964 > $O/chain_pmc.txt 2>&1
  summ $O/chain_pmc
  rm -rf $O/chain_pmc
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/chain_pmc -o p -- python3 $R/tools/dev_lone_timeline.py 2964 > $O/chain_pmc.txt 2>&1
  summ $O/chain_pmc
  rm -rf $O/chain_pmc
done
