#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel trace and/or PMC passes) per kernel name.
usage: prof_summary.py <dir> [<dir> ...]  -> prints a table; used to fill profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "")
    return name[:110]


def main():
    for d in sys.argv[1:]:
        print(f"## {d}")
        for path in sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)):
            dur = defaultdict(list)
            for row in csv.DictReader(open(path)):
                dur[row["Kernel_Name"]].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
            print(f"# kernel trace: {os.path.relpath(path, d)}")
            print(f"{'calls':>6} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'total_ms':>10}  kernel")
            for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
                print(f"{len(v):6d} {sum(v)/len(v)/1e3:10.2f} {min(v)/1e3:10.2f} {max(v)/1e3:10.2f} {sum(v)/1e6:10.3f}  {short(k)}")
        for path in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
            acc = defaultdict(list)
            for row in csv.DictReader(open(path)):
                acc[(row["Kernel_Name"], row["Counter_Name"])].append(float(row["Counter_Value"]))
            print(f"# counters: {os.path.relpath(path, d)}")
            print(f"{'calls':>6} {'avg':>16} {'counter':>14}  kernel")
            for (k, c), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
                print(f"{len(v):6d} {sum(v)/len(v):16.1f} {c:>14}  {short(k)}")


if __name__ == "__main__":
    main()
