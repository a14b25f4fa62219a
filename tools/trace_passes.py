#!/usr/bin/env python3
"""Per-pass kernel averages from a rocprofv3 --kernel-trace of `python bench.py`.

A default bench.py run launches the tick in four passes -- warm-up, the timed region (every n-th step also takes dispatch
timestamps), the every-launch-timed pass, the end-to-end pass (draw emission + overlapped read-back) -- and rocprofv3's
--stats summary blends them.  This splits the trace by launch index of the fused kernel (the counts are known from the
command line) and prints each kernel's average duration per pass, so that (the line's own bytes per entity) x N / (timed-region average)
can be checked against the line's `roofline.frac` from tracked files alone (--bench-line: the figure is read from the line).
    python tools/trace_passes.py <trace dir> --warmup W --steps K [--every 64] > profiles/r03/config3_kernel_passes.json"""
import argparse
import csv
import glob
import json
import os
from collections import defaultdict

ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--warmup", type=int, required=True)
ap.add_argument("--steps", type=int, required=True)
ap.add_argument("--every", type=int, default=None, help="launches of the every-launch pass (default min(steps, 64))")
ap.add_argument("--bytes-per-entity", type=float, default=None, help="algorithmic bytes per entity of the fused kernel; default: roofline.bytes_per_entity of --bench-line")
ap.add_argument("--bench-line", default=None, help="the JSON line bench.py printed in the traced run (or a file holding it): the figure the roofline is priced at comes from there")
ap.add_argument("--entities", type=int, default=1048576)
args = ap.parse_args()
every = min(args.steps, 64) if args.every is None else args.every
if args.bytes_per_entity is None:
    if not args.bench_line:
        ap.error("give --bench-line (the run's own line: its roofline.bytes_per_entity is what the kernel is priced at) or --bytes-per-entity")
    text = open(args.bench_line).read() if os.path.exists(args.bench_line) else args.bench_line
    line = next(ln for ln in reversed(text.splitlines()) if ln.startswith("{"))
    args.bytes_per_entity = float(json.loads(line)["roofline"]["bytes_per_entity"])

rows = []
for f in glob.glob(os.path.join(args.dir, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()


def short(name):
    for k in ("k_xform_cull", "k_compact_pairs", "k_compact_pack", "k_pairs", "k_compact", "k_emit_draws_staged", "k_nudge_roots_x", "k_advance_movers"):
        if k in name:
            return k
    return None


# pass boundaries by the fused kernel's launch index (the secondary legs of the line, if present, come after these)
bounds = [("warmup", args.warmup), ("timed_region", args.steps), ("every_launch_pass", every), ("end_to_end_warmup", max(args.warmup, 10)), ("end_to_end_pass", max(args.steps, 200))]      # (bench.py: that leg is at least 10 + 200 steps long)
edges, at = [], 0
for name, n in bounds:
    edges.append((name, at, at + n)); at += n
acc = defaultdict(lambda: defaultdict(list))
k1 = 0
current = None
for st, en, name in rows:
    s = short(name)
    if s is None:
        continue
    if s == "k_xform_cull":
        current = next((nm for nm, a, b in edges if a <= k1 < b), "after (secondary legs, parity)")
        k1 += 1
    if current is not None:
        acc[current][s].append((en - st) / 1000.0)
out = {"source": "rocprofv3 --kernel-trace of python bench.py --steps %d --warmup %d, split by launch index (tools/trace_passes.py)" % (args.steps, args.warmup),
       "launches_of_k_xform_cull": k1, "passes": {}}
for nm, _, _ in edges + [("after (secondary legs, parity)", 0, 0)]:
    if nm in acc:
        out["passes"][nm] = {k: {"launches": len(v), "avg_us": round(sum(v) / len(v), 3)} for k, v in sorted(acc[nm].items())}
tr = out["passes"].get("timed_region", {}).get("k_xform_cull")
if tr:
    gbs = args.bytes_per_entity * args.entities / (tr["avg_us"] * 1e-6) / 1e9
    out["roofline_from_timed_region"] = {"bytes_per_entity": args.bytes_per_entity, "achieved_GBs": round(gbs, 1), "frac_of_8000": round(gbs / 8000.0, 4)}
print(json.dumps(out, indent=1))
