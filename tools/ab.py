#!/usr/bin/env python3
"""In-process A/B of kernel variants on ONE device (box-to-box variance makes cross-run numbers useless).
Contexts with different SC_TICK_VARIANT / SC_TICK_SPANS are created side by side and timed in
interleaved rounds; per-kernel times come from HIP events on every tick."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sc_gameengine_amd import capi, synth_world as sw
from sc_gameengine_amd.tick import WorldTick, camera_view_proj

# a variant is "BITS[:SPANS]": SC_TICK_VARIANT bits (8 = compaction and pair search as separate launches,
# bits 8+ = pair-kernel grid cap) and optionally SC_TICK_SPANS (workgroups of the fused kernel)
variants = [v for v in (sys.argv[1] if len(sys.argv) > 1 else "0,8").split(",")]
w = sw.config("config3") if os.environ.get("AB_WORLD", "config3") == "config3" else sw.generate(256, 256, 15, hierarchy=False)
vp = camera_view_proj(w.camera)
ctxs = {}
for v in variants:
    var, _, spans = v.partition(":")
    os.environ["SC_TICK_VARIANT"] = var
    os.environ["SC_TICK_SPANS"] = spans or "1536"
    t = WorldTick.from_world(w, broadphase=True)
    t.set_view_proj(vp)
    ctxs[v] = t
cases = {"xc_dirty": (capi.XFORM | capi.CULL, True), "full_dirty": (capi.FULL, True), "xc_static": (capi.XFORM | capi.CULL, False)}
res = {c: {v: {k: [] for k in ("k1", "k2", "k3", "nudge")} for v in variants} for c in cases}
for rnd in range(5):
    for cname, (flags, nudge) in cases.items():
        for v, t in ctxs.items():
            for _ in range(5):
                if nudge: t.nudge_roots_x(0.01)
                t.run(flags)
            t.sync(); t.set_profiling(1)
            for _ in range(40):
                if nudge: t.nudge_roots_x(0.01)
                t.run(flags)
            r = res[cname][v]
            r["k1"] += list(t.kernel_times_ms(capi.K_XFORM_CULL)); r["k2"] += list(t.kernel_times_ms(capi.K_COMPACT))
            r["k3"] += list(t.kernel_times_ms(capi.K_PAIRS)); r["nudge"] += list(t.kernel_times_ms(capi.K_NUDGE))
            t.set_profiling(0)
for cname in cases:
    for v in variants:
        r = res[cname][v]
        print(cname, "variant", v, {k: (round(float(np.median(x)) * 1e3, 2), round(float(np.min(x)) * 1e3, 2)) for k, x in r.items() if len(x)}, flush=True)
