#!/usr/bin/env python3
"""Print the last kernels of a rocprofv3 kernel trace as a timeline (start offset, duration, queue, name)."""
import csv, sys, glob
path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-int(sys.argv[2]) if len(sys.argv) > 2 else -40:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f} us  q{r.get('Queue_Id', '?'):>3}  {r['Kernel_Name'][:60]}")
