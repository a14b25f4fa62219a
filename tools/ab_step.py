#!/usr/bin/env python3
"""In-process A/B of library BUILDS on one device, by whole-step time (box-to-box variance makes cross-run numbers useless).

    python tools/ab_step.py [--workload config3|config3dyn|config5] libA.so libB.so ...

Each library is loaded side by side (ctypes, separate handles), gets its own context on the same world, and the builds
are timed in interleaved rounds: per build the wall time of a burst of steps (device-bound: the host issues two launches
per step) plus the kernels' own durations from dispatch timestamps on a separate profiled burst."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sc_gameengine_amd import capi, synth_world as sw          # noqa: E402
from sc_gameengine_amd.tick import WorldTick, camera_view_proj  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--workload", default="config3")
ap.add_argument("--rounds", type=int, default=6)
ap.add_argument("--burst", type=int, default=400)
ap.add_argument("--separate-producer", action="store_true", help="the frame producer as a launch of its own (what the end-of-tick kernel costs without it)")
args = ap.parse_args()

if args.workload == "config5":
    w = sw.generate_config5(128, 256)
    kind, param = 2, 1.0 / 60.0
else:
    w = sw.config("config3")
    if args.workload == "config3dyn":
        dyn = (np.arange(w.n) % 16) == 4
        w.group[dyn], w.mask[dyn] = sw.GROUP_DYNAMIC, sw.MASK_ALL
    kind, param = 1, 0.01
vp = camera_view_proj(w.camera)
flags = capi.FULL | (0 if args.separate_producer else capi.PRODUCE_NEXT)

ctxs = {}
for path in args.libs:
    path_env, _, envs = path.partition("#")               # "lib.so#SC_TICK_FAST_PAIRS=0,SC_TICK_HOME_PERIOD=8": environment for this context's creation
    for kv in filter(None, envs.split(",")):
        k, _, v = kv.partition("=")
        os.environ[k] = v
    lib_path, _, variant = path_env.partition("@")        # "lib.so@48": SC_TICK_VARIANT for this context (tuning knobs)
    variant, _, spans = variant.partition(":")            # "lib.so@0:4096": SC_TICK_SPANS too (workgroups of the fused kernel)
    os.environ["SC_TICK_VARIANT"] = variant or "0"
    if spans:
        os.environ["SC_TICK_SPANS"] = spans
    else:
        os.environ.pop("SC_TICK_SPANS", None)
    capi._LIB = None
    capi.LIB_PATH = os.path.abspath(lib_path)
    t = WorldTick.from_world(w, broadphase=True)
    for kv in filter(None, envs.split(",")):
        os.environ.pop(kv.partition("=")[0], None)
    t.set_view_proj(vp)
    t.set_frame_producer(kind, param)
    (t.advance_movers if kind == 2 else t.nudge_roots_x)(param)
    for _ in range(30):
        t.run(flags)
    t.sync()
    ctxs[path] = t

res = {p: {"step_us": [], "k1_us": [], "eot_us": []} for p in args.libs}
for rnd in range(args.rounds):
    for path, t in ctxs.items():
        for _ in range(20):
            t.run(flags)
        t.sync()
        t0 = time.perf_counter()
        for _ in range(args.burst):
            t.run(flags)
        t.sync()
        res[path]["step_us"].append((time.perf_counter() - t0) / args.burst * 1e6)
        t.set_profiling(1)
        for _ in range(40):
            t.run(flags)
        res[path]["k1_us"] += [x * 1e3 for x in t.kernel_times_ms(capi.K_XFORM_CULL)]
        res[path]["eot_us"] += [x * 1e3 for x in t.kernel_times_ms(capi.K_PAIRS)]
        t.set_profiling(0)
vis = {p: int(t.counts().visible) for p, t in ctxs.items()}
prs = {p: int(t.counts().pairs) for p, t in ctxs.items()}
for path in args.libs:
    r = res[path]
    print(json.dumps({"lib": os.path.basename(path.partition("#")[0].partition("@")[0]) + path[len(path.partition("#")[0].partition("@")[0]):], "workload": args.workload,
                      "step_us_median": round(float(np.median(r["step_us"])), 2), "step_us_min": round(float(np.min(r["step_us"])), 2),
                      "k_xform_cull_us": round(float(np.median(r["k1_us"])), 2), "end_of_tick_us": round(float(np.median(r["eot_us"])), 2) if r["eot_us"] else None,
                      "gap_us": round(float(np.median(r["step_us"])) - float(np.median(r["k1_us"])) - (float(np.median(r["eot_us"])) if r["eot_us"] else 0.0), 2),
                      "visible": vis[path], "pairs": prs[path]}), flush=True)
for t in ctxs.values():
    t.close()
