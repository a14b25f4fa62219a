#!/usr/bin/env python3
"""On-rails traffic agents on random laned worlds over random step sizes, speed multipliers and run lengths, separate advance
calls or the fused frame producer: lane state, positions and world matrices against the oracle's restatement, as bit patterns.
python tools/stress_traffic.py [--seeds 12]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sc_gameengine_amd import capi                                # noqa: E402
from sc_gameengine_amd.tick import WorldTick, camera_view_proj     # noqa: E402
from oracle import oracle_py                                       # noqa: E402
from tests.test_gpu_traffic import laned_world, oracle_side, oracle_advance, bits   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seeds", type=int, default=12)
args = ap.parse_args()
oracle_py.build(); oracle = oracle_py
bad = 0
for seed in range(args.seeds):
    rng = np.random.default_rng(3000 + seed)
    sx, sz = int(rng.integers(4, 20)), int(rng.integers(4, 20))
    dt = float(rng.choice([1.0 / 120.0, 1.0 / 60.0, 1.0 / 30.0, 0.1]))
    mult = float(rng.choice([1.0, 0.5, 2.0, 0.0]))
    ticks = int(rng.integers(10, 90))
    fused = bool(rng.integers(0, 2))
    w = laned_world(sx, sz, seed=seed) if sx * sz * 12 >= 700 else laned_world(8, 8, seed=seed)
    ow, ol, st = oracle_side(oracle, w)
    vp = camera_view_proj(w.camera)
    t = WorldTick.from_world(w, broadphase=True, max_pairs=1 << 20)
    t.set_view_proj(vp)
    t.set_traffic_speed_multiplier(mult)
    why = None
    if fused:
        t.set_frame_producer(2, dt); t.advance_movers(dt); oracle_advance(ow, ol, w, st, dt, mult)
    for k in range(ticks):
        if not fused:
            oracle_advance(ow, ol, w, st, dt, mult); t.advance_movers(dt)
        ow.transform_system(); ow.culling_system(view_proj=vp)
        t.run(capi.FULL | (capi.PRODUCE_NEXT if fused else 0))
        if k % 7 == 0 or k == ticks - 1:
            if fused:
                t.sync()
            # (fused: the device has already produced frame k + 1; compare what tick k consumed -- the matrices -- and bring the oracle level after)
            if not np.array_equal(t.world_matrices(), ow.world_matrices()[:w.n]):
                why = f"tick {k}: world matrices differ"; break
            if not np.array_equal(t.visible(), ow.visible()):
                why = f"tick {k}: visible list differs"; break
        if fused:
            oracle_advance(ow, ol, w, st, dt, mult)
            if k % 7 == 0 or k == ticks - 1:
                ln, ls, sp, md = t.traffic_agents()
                a = w.is_agent.astype(bool)
                if not (np.array_equal(ln[a], st["lane"][a]) and np.array_equal(bits(ls[a]), bits(st["s"][a])) and np.array_equal(bits(sp[a]), bits(st["speed"][a]))):
                    why = f"tick {k}: lane state differs"; break
                if not np.array_equal(bits(t.positions()), bits(ow.local_positions()[:w.n])):
                    why = f"tick {k}: positions differ"; break
        elif k % 7 == 0 or k == ticks - 1:
            ln, ls, sp, md = t.traffic_agents()
            a = w.is_agent.astype(bool)
            if not (np.array_equal(ln[a], st["lane"][a]) and np.array_equal(bits(ls[a]), bits(st["s"][a])) and np.array_equal(bits(sp[a]), bits(st["speed"][a]))):
                why = f"tick {k}: lane state differs"; break
    if why:
        bad += 1
        print(f"seed {seed}: {sx}x{sz} sectors dt {dt:.4f} mult {mult} ticks {ticks} fused {fused}: {why}", flush=True)
    t.close(); ow.close(); ol.close()
print(f"{args.seeds - bad} of {args.seeds} traffic runs equal")
sys.exit(1 if bad else 0)
