#!/usr/bin/env python3
"""Tuning sweep on one GPU: span count (grid size), hipGraph replay, event-profiling overhead."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sc_gameengine_amd import capi, synth_world as sw
from sc_gameengine_amd.tick import WorldTick, camera_view_proj

w = sw.config("config3")
vp = camera_view_proj(w.camera)
flags = capi.XFORM | capi.CULL
steps = 300
for spans in (512, 768, 1024, 1280, 1536, 2048, 4096):
    os.environ["SC_TICK_SPANS"] = str(spans)
    t = WorldTick.from_world(w, broadphase=False)
    t.set_view_proj(vp)
    for graph in (0, 1):
        t.set_graph_mode(bool(graph))
        for prof in (0, 1):
            for _ in range(20):
                t.nudge_roots_x(0.01); t.run(flags)
            t.sync()
            t.set_profiling(prof)
            t0 = time.perf_counter()
            for _ in range(steps):
                t.nudge_roots_x(0.01); t.run(flags)
            t.sync()
            dt = (time.perf_counter() - t0) / steps * 1e6
            k1 = t.kernel_times_ms(capi.K_XFORM_CULL) if prof else []
            t.set_profiling(0)
            print(json.dumps({"spans": spans, "graph": graph, "prof": prof, "us_per_step": round(dt, 2),
                              "k1_us": round(float(np.mean(k1)) * 1e3, 2) if len(k1) else None}), flush=True)
    # static regime (nothing dirty) and xform-only
    t.set_graph_mode(False)
    for name, fl, nudge in (("static_xc", flags, False), ("xform_only_dirty", capi.XFORM, True)):
        for _ in range(10):
            if nudge: t.nudge_roots_x(0.01)
            t.run(fl)
        t.sync(); t.set_profiling(1)
        for _ in range(100):
            if nudge: t.nudge_roots_x(0.01)
            t.run(fl)
        k1 = t.kernel_times_ms(capi.K_XFORM_CULL); t.set_profiling(0)
        print(json.dumps({"spans": spans, "case": name, "k1_us": round(float(np.mean(k1)) * 1e3, 2)}), flush=True)
    t.close()
