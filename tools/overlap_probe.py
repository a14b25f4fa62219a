#!/usr/bin/env python3
"""One GPU, no neighbours: the in-order step (fused kernel, then compaction + pair search in one launch) against the
pipelined step (pair search of tick t on the pairs stream while the tick stream runs tick t+1).  Whole-step wall time."""
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
ap.add_argument("--workload", default="config5")
ap.add_argument("--steps", type=int, default=400)
ap.add_argument("--rounds", type=int, default=4)
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
flags = capi.FULL | capi.PRODUCE_NEXT

ctxs = {}
for depth in (0, 2, 3):
    t = WorldTick.from_world(w, broadphase=True)
    t.set_view_proj(vp)
    t.set_frame_producer(kind, param)
    (t.advance_movers if kind == 2 else t.nudge_roots_x)(param)
    if depth:
        t.set_pipelined(depth)
    ctxs[depth] = t

def step(t, depth):
    if depth:
        t.run(flags | capi.SPLIT_PAIRS); t.run_pairs()
    else:
        t.run(flags)

res = {d: [] for d in ctxs}
for rnd in range(args.rounds):
    for d, t in ctxs.items():
        for _ in range(30):
            step(t, d)
        t.sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(t, d)
        issued = time.perf_counter() - t0
        t.sync()
        res[d].append(((time.perf_counter() - t0) / args.steps * 1e6, issued / args.steps * 1e6))
out = {"workload": args.workload, "entities": int(w.n)}
for d, t in ctxs.items():
    c = t.counts()
    out["in_order" if not d else f"pipelined_depth{d}"] = {"us_per_step": round(float(np.median([r[0] for r in res[d]])), 2),
        "host_issue_us": round(float(np.median([r[1] for r in res[d]])), 2), "visible": int(c.visible), "pairs": int(c.pairs)}
    t.close()
print(json.dumps(out))
