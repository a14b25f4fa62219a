#!/usr/bin/env python3
"""Is a flow bound by the host that issues it?  From `rocprofv3 --hip-trace --kernel-trace --output-format csv`: for every launch of the named
kernels, the time from the END of the HIP call that launched it to the START of the kernel on the device.  A queue that always has work
waiting shows lags of tens of microseconds (the kernel sat in the queue); a host-bound flow shows lags of a few (it started when it arrived).
Usage: launch_lag.py <dir> [--last N]"""
import csv
import glob
import json
import sys
from collections import defaultdict

root = sys.argv[1]
last = int(sys.argv[sys.argv.index("--last") + 1]) if "--last" in sys.argv else 150
k = list(csv.DictReader(open(glob.glob(root + "/**/*kernel_trace.csv", recursive=True)[0])))
a = {r["Correlation_Id"]: r for r in csv.DictReader(open(glob.glob(root + "/**/*hip_api_trace.csv", recursive=True)[0]))}
lag = defaultdict(list)
for r in k:
    c = a.get(r["Correlation_Id"])
    if not c:
        continue
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
    lag[name].append((int(r["Start_Timestamp"]), (int(r["Start_Timestamp"]) - int(c["End_Timestamp"])) / 1e3, (int(c["End_Timestamp"]) - int(c["Start_Timestamp"])) / 1e3))
out = {}
for name, v in lag.items():
    v = sorted(v)[-last:]
    if len(v) < 20:
        continue
    lags = sorted(x[1] for x in v); calls = sorted(x[2] for x in v)
    out[name] = {"launches": len(v), "lag_us_p10": round(lags[len(lags) // 10], 1), "lag_us_median": round(lags[len(lags) // 2], 1), "lag_us_p90": round(lags[9 * len(lags) // 10], 1),
                 "host_call_us_median": round(calls[len(calls) // 2], 1)}
print(json.dumps(out, indent=1))
