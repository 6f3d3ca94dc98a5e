# dev: the operations of one query alone from its scan's upload to the cold pass (last repetition), under rocprof
set -e
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf $O/lone_tl
rocprofv3 --kernel-trace --output-format csv -d $O/lone_tl -o t -- python3 $R/tools/dev_lone_prep_timeline.py > $O/lone_prep_tl.txt 2>&1
grep "query alone" $O/lone_prep_tl.txt
python3 - <<P
import csv, glob
f = glob.glob("$O/lone_tl/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
idx = [i for i, r in enumerate(rows) if "nn_chain_kernel" in r[2]]
i0, i1 = idx[-2] + 1, idx[-1]
t0 = rows[i0][0]
prev_end = t0
for s, e, n in rows[i0:i1 + 1]:
    nm = n.split("(")[0].replace("void gloc::reg::", "").replace("gloc::reg::", "").replace("void gloc::", "").replace("gloc::", "")[:64]
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:6.1f}  {nm}")
    prev_end = e
P
rm -rf $O/lone_tl
