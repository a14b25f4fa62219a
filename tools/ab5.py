#!/usr/bin/env python3
"""Per-kernel timing on the config-5 world (movers on device), merged vs separate compaction/pair launches."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sc_gameengine_amd import capi, synth_world as sw
from sc_gameengine_amd.tick import WorldTick, camera_view_proj
w = sw.generate_config5(128, 256)
vp = camera_view_proj(w.camera)
for var in (sys.argv[1] if len(sys.argv) > 1 else "0,8").split(","):
    os.environ["SC_TICK_VARIANT"] = var
    t = WorldTick.from_world(w, broadphase=True)
    t.set_view_proj(vp)
    for flags, name in ((capi.FULL, "full"), (capi.XFORM | capi.BROADPHASE, "xform+bp")):
        for _ in range(10):
            t.advance_movers(1 / 60); t.run(flags)
        t.sync(); t.set_profiling(1)
        for _ in range(60):
            t.advance_movers(1 / 60); t.run(flags)
        r = {k: t.kernel_times_ms(v) for k, v in (("k1", capi.K_XFORM_CULL), ("k2", capi.K_COMPACT), ("k3", capi.K_PAIRS), ("mv", capi.K_NUDGE))}
        t.set_profiling(0)
        c = t.counts()
        print("variant", var, name, {k: round(float(np.median(x)) * 1e3, 2) for k, x in r.items() if len(x)}, "pairs", c.pairs, "big", c.big_boxes, "binfull", c.bin_overflow, flush=True)
    t.close()
