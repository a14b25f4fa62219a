#!/usr/bin/env python3
"""Random small worlds against the oracle's brute force: pair SET, world AABBs and 600 ray hits per world must be identical.  Densities, layer mixes,
box sizes and sector counts vary per seed so that bins of every fill (empty ... beyond 64 with overflow lists) and every count
of dynamic records per bin come up.  A one-off confidence run for changes to the pair search; not part of the test suite.
    python tools/stress_broadphase.py [--seeds 60]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sc_gameengine_amd import capi, synth_world as sw            # noqa: E402
from sc_gameengine_amd.tick import WorldTick                      # noqa: E402
from oracle import oracle_py                                      # noqa: E402
from tests import worlds                                          # noqa: E402
from tests.test_gpu_rays import random_rays, compare               # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seeds", type=int, default=60)
args = ap.parse_args()
oracle_py.build()
oracle = oracle_py
FLAGS = capi.XFORM | capi.BROADPHASE | capi.DENSE_AABBS | capi.RAYS


def key(p):
    p = np.asarray(p, np.uint64).reshape(-1, 2)
    return np.sort(p[:, 0] << np.uint64(32) | p[:, 1])


bad = silent = 0
for seed in range(args.seeds):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(300, 6000))
    spread = float(rng.choice([20.0, 60.0, 150.0, 400.0]))          # 20 m: everything in one or two sectors (overflow lists)
    w = worlds.random_world(n, seed=seed, spread=spread, max_depth=int(rng.integers(0, 4)), p_child=float(rng.choice([0.0, 0.3])), p_no_bounds=0.05)
    pdyn = float(rng.choice([0.02, 0.1, 0.3, 0.6, 1.0]))
    dyn = rng.random(n) < pdyn
    w.group[:] = np.where(dyn, sw.GROUP_DYNAMIC, sw.GROUP_STATIC).astype(np.uint32)
    w.mask[:] = np.where(dyn, sw.MASK_ALL, sw.MASK_STATIC).astype(np.uint32)
    if seed % 7 == 3:                                                # a few worlds with layers that only pass across kinds
        k = rng.integers(0, 4, n)
        w.group[:] = np.choose(k, [1, 2, 4, 8]).astype(np.uint32); w.mask[:] = np.choose(k, [0xFFFFFFFF, 1, 8, 4]).astype(np.uint32)
    w.scale[:] *= float(rng.choice([0.3, 1.0, 2.5]))
    ow = worlds.oracle_world(oracle, w, camera=False)
    t = WorldTick.from_world(w, broadphase=True, max_pairs=1 << 22)
    rays = random_rays(rng, 600, spread * 1.2)
    t.set_ray_queries(*rays)
    ok = True
    for tick in range(2):
        if tick:
            ow.nudge_roots_x(0.8); t.nudge_roots_x(0.8)
        ow.transform_system(); t.run(FLAGS)
        mn, mx = ow.world_aabbs(); gmn, gmx = t.world_aabbs()
        want = oracle.broadphase_bruteforce(mn, mx, w.group, w.mask)
        got, total = t.pairs()
        c = t.counts()
        same = np.array_equal(gmn, mn) and np.array_equal(gmx, mx) and total == len(want) and np.array_equal(key(got), key(want))
        rays_ok = True
        if not c.border_lost:                      # (a lost record is lost to the rays as well)
            try:
                compare(t.ray_hits(), oracle.raycast_boxes(mn, mx, w.group, w.mask, *rays))
            except AssertionError as e:
                rays_ok = False
                print(f"seed {seed} tick {tick}: ray hits differ: {str(e)[:200]}", flush=True)
        if (not same or not rays_ok) and not (c.pairs_truncated or c.border_lost):
            silent += 1                            # a mismatch WITHOUT a reported loss: the only kind that is a bug
        if not same or not rays_ok or c.pairs_truncated or c.border_lost:
            ok = False
            print(f"seed {seed} tick {tick}: n={n} spread={spread} pdyn={pdyn} pairs gpu {total} oracle {len(want)} truncated {c.pairs_truncated} lost {c.border_lost} overflow {c.bin_overflow}", flush=True)
    bad += 0 if ok else 1
    t.close(); ow.close()
print(f"{args.seeds - bad} of {args.seeds} worlds equal, {bad} with differences of which {silent} tick(s) without a reported loss")
sys.exit(1 if silent else 0)
