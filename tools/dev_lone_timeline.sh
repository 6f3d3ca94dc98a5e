# dev: kernel time line of one query's registration (20 jobs), last repetition
set -e
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf $O/lone_tl
rocprofv3 --kernel-trace --output-format csv -d $O/lone_tl -o t -- python3 $R/tools/dev_lone_timeline.py $1 > $O/lone_tl.txt 2>&1
tail -1 $O/lone_tl.txt
python3 - <<P
import csv, glob
f = glob.glob("$O/lone_tl/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# the last batch: from the last nn_compact kernel with PAIRS (cold) back
idx = [i for i, r in enumerate(rows) if "nn_compact_kernel<2, true" in r[2]]
i0 = max(idx[-1] - 8, 0)  # (a few operations before the cold pass: the batch's memsets)
t0 = rows[i0][0]
prev_end = t0
tot = 0
for s, e, n in rows[i0:]:
    nm = n.split("(")[0].replace("void gloc::reg::", "").replace("gloc::reg::", "")[:60]
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:6.1f}  {nm}")
    prev_end = e
    tot += e - s
print("span", (prev_end - t0) / 1e3, "us; kernels", tot / 1e3, "us")
P
rm -rf $O/lone_tl
