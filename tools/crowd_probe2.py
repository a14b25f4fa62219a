#!/usr/bin/env python3
"""A crowded DISTRICT inside a big world: 512 sectors at 200 boxes each (the engine's streaming budget, main.cpp:92-99) in the
middle of a 256 x 256-sector world of the usual 16 per sector.  The pair search hands consecutive sectors to one wave in runs
of 16, so before round 3 the district's sectors met 32 waves, sixteen crowded sectors each, one after the other."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sc_gameengine_amd import capi, synth_world as sw
from sc_gameengine_amd.tick import WorldTick

w = sw.config("config3")
rng = np.random.default_rng(3)
# district: sectors x 100..131, z 100..115; 184 extra boxes each, appended to the world as roots
sx, sz = np.meshgrid(np.arange(100, 132), np.arange(100, 116), indexing="xy")
sx, sz = sx.ravel(), sz.ravel()
extra = 184
n2 = len(sx) * extra
pos = np.zeros((n2, 3), np.float32)
pos[:, 0] = (np.repeat(sx, extra) * 64.0 + rng.uniform(1, 63, n2)).astype(np.float32)
pos[:, 2] = (np.repeat(sz, extra) * 64.0 + rng.uniform(1, 63, n2)).astype(np.float32)
pos[:, 1] = 1.0
scale = rng.uniform(0.4, 1.9, (n2, 3)).astype(np.float32)
rot = np.zeros((n2, 3), np.float32); rot[:, 1] = rng.uniform(0, 6.28, n2).astype(np.float32)
dyn = rng.random(n2) < 0.3
t = WorldTick.from_world(w, broadphase=True, max_pairs=1 << 21, capacity=w.n + n2)
t.append_entities(pos, rot, scale, group=np.where(dyn, 1, 2).astype(np.uint32), mask=np.where(dyn, 0xFFFFFFFF, 1).astype(np.uint32))
for _ in range(5):
    t.run(capi.XFORM | capi.BROADPHASE)
t.sync(); t.set_profiling(1)
for _ in range(20):
    t.run(capi.XFORM | capi.BROADPHASE)
k1, k3 = t.kernel_times_ms(capi.K_XFORM_CULL), t.kernel_times_ms(capi.K_PAIRS)
c = t.counts()
print(json.dumps({"world": "config 3 + a district of 512 sectors at 200 boxes", "entities": int(c.entities), "k_xform_cull_us": round(float(np.median(k1)) * 1e3, 1),
                  "end_of_tick_us": round(float(np.median(k3)) * 1e3, 1), "pairs": int(c.pairs), "bin_overflow": int(c.bin_overflow), "variant": os.environ.get("SC_TICK_VARIANT", "0")}))
t.close()
