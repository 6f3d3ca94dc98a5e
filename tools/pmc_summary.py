"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel: mean counter value per dispatch.

    python tools/pmc_summary.py <dir-or-csv> [<dir-or-csv> ...] [--match nn_compact] [--json out.json]
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def files(args):
    out = []
    for a in args:
        if os.path.isdir(a):
            out += glob.glob(os.path.join(a, "**", "*counter_collection.csv"), recursive=True)
        else:
            out.append(a)
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    match = sys.argv[sys.argv.index("--match") + 1] if "--match" in sys.argv else ""
    jout = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    if "--match" in sys.argv:
        args.remove(match)
    if jout:
        args.remove(jout)
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    dur = defaultdict(lambda: [0.0, 0])
    for f in files(args):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if match and match not in k:
                continue
            a = acc[k][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                d = dur[k]
                d[0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
                d[1] += 1
    out = {}
    for k in sorted(acc):
        out[k] = {"dispatches": dur[k][1], "mean_us_under_pmc": dur[k][0] / max(dur[k][1], 1)}
        for c, (s, n) in sorted(acc[k].items()):
            out[k][c] = s / n
        print(k, json.dumps(out[k]))
    if jout:
        json.dump(out, open(jout, "w"), indent=1)


if __name__ == "__main__":
    main()
