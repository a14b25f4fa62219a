#!/usr/bin/env python3
"""Cost of the ray batch inside the tick: config-5 world, 1M entities, one front ray per vehicle
as the traffic AI casts it (sc_traffic_ai.cpp:303-319), tick time with and without SC_TICK_RAYS."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sc_gameengine_amd import capi, synth_world as sw
from sc_gameengine_amd.tick import WorldTick

w = sw.generate_config5(128, 256)
t = WorldTick.from_world(w, broadphase=True)
t.set_camera(w.camera)
veh = np.flatnonzero(w.mover_kind == 1)
yaw = w.rot[veh, 1]
fwd = np.stack([np.sin(yaw), np.zeros_like(yaw), np.cos(yaw)], axis=1).astype(np.float32)
o = (w.pos[veh] + fwd * np.float32(1.7) + np.float32([0, 0.6, 0])).astype(np.float32)
out = {"entities": int(w.n), "rays": int(len(veh))}
t.set_frame_producer(2, 1.0 / 60.0)
for name, rays in (("tick_us", 0), ("tick_with_rays_us", len(veh)), ("tick_with_4096_rays_us", 4096)):
    t.set_ray_queries(o[:rays], fwd[:rays], np.full(rays, 20.0, np.float32), np.full(rays, 1, np.uint32))
    fl = capi.FULL | capi.PRODUCE_NEXT | (capi.RAYS if rays else 0)
    for _ in range(20): t.run(fl)
    t.sync()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n): t.run(fl)
    t.sync()
    out[name] = round((time.perf_counter() - t0) / n * 1e6, 2)
    if rays:
        h = t.ray_hits(); out[name.replace("_us", "_hits")] = int(h["hit"].sum())
print(json.dumps(out))
