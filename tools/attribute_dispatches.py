#!/usr/bin/env python3
"""Which HIP call stands behind the runtime's own dispatches (__amd_rocclr_fillBufferAligned / copyBuffer) of a traced run?
Reads the csv output of `rocprofv3 --hip-trace --kernel-trace --output-format csv` (kernel_trace.csv + hip_api_trace.csv), joins the
two on the correlation id and prints, per runtime kernel, the HIP functions that launched it, with counts, the calling thread and the
call's place relative to this library's own launches (hipExtLaunchKernel / hipLaunchKernel of sctick kernels).
Usage: attribute_dispatches.py <dir> [--steps N]"""
import csv
import glob
import json
import sys
from collections import Counter

root = sys.argv[1]
steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 0
kfile = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)[0]
afile = glob.glob(root + "/**/*hip_api_trace.csv", recursive=True)[0]
api = {}
for r in csv.DictReader(open(afile)):
    api[r["Correlation_Id"]] = (r["Function"], r["Thread_Id"])
out = {}
threads = Counter()
rows = list(csv.DictReader(open(kfile)))
# the window of the last `steps` steps: from the start of the step's first library kernel, `steps` launches of the fused kernel before
# the end, to the end of the last library kernel -- what is dispatched inside it is dispatched per step, not at set-up
lib = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "k_xform_cull" in r["Kernel_Name"])
w0 = lib[-steps][0] if steps and len(lib) >= steps else (lib[0][0] if lib else 0)
w1 = max((int(r["End_Timestamp"]) for r in rows if "sctick" in r["Kernel_Name"]), default=0)
in_window = Counter()
for r in rows:
    name = r["Kernel_Name"]
    if w0 <= int(r["Start_Timestamp"]) <= w1:
        in_window[name.split("(")[0][:48]] += 1
    short = name.split("(")[0][:48]
    fn, tid = api.get(r["Correlation_Id"], ("?", "?"))
    out.setdefault(short, Counter())[fn] += 1
    if "sctick" in name:
        threads[tid] += 1
main_thread = threads.most_common(1)[0][0] if threads else "?"
res = {"steps": steps, "library_launch_thread": main_thread, "kernels": {}}
for k, c in sorted(out.items(), key=lambda kv: -sum(kv[1].values())):
    res["kernels"][k] = {"dispatches_in_the_whole_run": sum(c.values()), "launched_by": dict(c),
                         "dispatches_inside_the_last_steps": in_window.get(k, 0), "per_step": round(in_window.get(k, 0) / steps, 3) if steps else None}
# every HIP call of the traced process, by function: what the library itself never calls per step (hipMemsetAsync, hipMemcpyAsync
# outside learn ticks) shows here as somebody else's
calls = Counter(fn for fn, _ in api.values())
res["hip_calls"] = {k: v for k, v in calls.most_common(24)}
print(json.dumps(res, indent=1))
