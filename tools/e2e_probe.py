#!/usr/bin/env python3
"""Which part of the per-frame read-back slows the fused kernel?  config 3, kernel times by dispatch timestamps, five loops:
tick only / + draw emission / + emission and the staged read-back / read-back without draws / read-back, host never looks."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sc_gameengine_amd import capi, synth_world as sw          # noqa: E402
from sc_gameengine_amd.tick import WorldTick, camera_view_proj  # noqa: E402

w = sw.config("config3")
vp = camera_view_proj(w.camera)
base = capi.FULL | capi.PRODUCE_NEXT


def loop(name, flags, readback, acquire, steps=300):
    t = WorldTick.from_world(w, broadphase=True, max_draws=6000)
    t.set_view_proj(vp)
    t.set_frame_producer(1, 0.01); t.nudge_roots_x(0.01)
    if readback:
        t.set_frame_readback(8192, 6000)
    for k in range(40):
        t.run(flags)
        if acquire and k:
            t.acquire_frame(frames_back=1, copy=False)
    t.sync()
    t0 = time.perf_counter()
    for k in range(steps):
        t.run(flags)
        if acquire:
            t.acquire_frame(frames_back=1, copy=False)
    t.sync()
    wall = (time.perf_counter() - t0) / steps * 1e6
    t.set_profiling(1)
    for k in range(64):
        t.run(flags)
        if acquire:
            t.acquire_frame(frames_back=1, copy=False)
    t.sync()
    k1 = np.mean(t.kernel_times_ms(capi.K_XFORM_CULL)) * 1e3
    kp = np.mean(t.kernel_times_ms(capi.K_PAIRS)) * 1e3
    t.set_profiling(0)
    t.close()
    print(json.dumps({"loop": name, "step_us": round(wall, 2), "k_xform_cull_us": round(float(k1), 2), "end_of_tick_us": round(float(kp), 2)}), flush=True)


loop("tick only", base, False, False)
loop("tick + draw emission", base | capi.DRAWS, False, False)
loop("tick + draws + read-back, host takes frame t-1", base | capi.DRAWS, True, True)
loop("tick + read-back without draws, host takes frame t-1", base, True, True)
loop("tick + draws + read-back, host never looks", base | capi.DRAWS, True, False)
loop("tick only (again)", base, False, False)
