#!/usr/bin/env python3
"""Per-kernel times of a crowded world (200 boxes per sector: every bin overflows) against a sparse one of the same size."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sc_gameengine_amd import capi
from sc_gameengine_amd.tick import WorldTick
from tests.test_gpu_broadphase import crowded_world

for name, w in (("dense", crowded_world(32, 16, 200, seed=41)), ("sparse", crowded_world(64, 32, 50, seed=41))):
    t = WorldTick.from_world(w, broadphase=True, max_pairs=1 << 21)
    for _ in range(5):
        t.run(capi.XFORM | capi.BROADPHASE)
    t.sync(); t.set_profiling(1)
    for _ in range(20):
        t.run(capi.XFORM | capi.BROADPHASE)
    k1, k3 = t.kernel_times_ms(capi.K_XFORM_CULL), t.kernel_times_ms(capi.K_PAIRS)
    c = t.counts()
    print(json.dumps({"world": name, "k_xform_cull_us": round(float(np.median(k1)) * 1e3, 1), "end_of_tick_us": round(float(np.median(k3)) * 1e3, 1),
                      "pairs": int(c.pairs), "bin_overflow": int(c.bin_overflow)}))
    t.close()
