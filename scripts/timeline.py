#!/usr/bin/env python3
"""Print the kernel timeline (start offset, duration, gap to previous end) of the last N
dispatches of a rocprofv3 kernel-trace CSV."""
import csv, glob, os, sys
d, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40
path = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"]); prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s-t0)/1e3:10.1f} us  dur {(e-s)/1e3:8.1f}  gap {(s-prev_end)/1e3:8.1f}  q{r.get('Queue_Id','?')}  grid {r.get('Grid_Size','?'):>9}  {r['Kernel_Name'][:70]}")
    prev_end = max(prev_end, e)
