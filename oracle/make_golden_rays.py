#!/usr/bin/env python3
"""Golden vectors for the ray/box slab test from the REAL reference function sc::editor::intersectRayAABB
(tools/world_editor/editor_core/editor_core.cpp:438-469), compiled unmodified into oracle/_ref/libsc_ref_editor.so.

TEST INFRASTRUCTURE ONLY.  Run where /root/reference exists:

    make -C oracle ref && python oracle/make_golden_rays.py

Writes tests/golden/ray_aabb_ref.npz: inputs and the reference's outputs (DATA; the reference itself does not travel).

Two groups of cases:
  * `a_*`  general cases for oracle-vs-reference: random rays and boxes, directions with |component| below / at / above the
           1e-6 "parallel" threshold, origins inside boxes and exactly on faces, grazing rays, inverted boxes, negative
           directions, infinite box bounds, tiny and huge magnitudes;
  * `b_*`  cases the GPU ray queries can run one per sector (tests/test_gpu_rays.py): case i lives in sector i of a row of
           sectors, its box has dyadic coordinates (so the device's centre / half-extent form of the box reproduces it
           exactly), everything lies within a few metres of the sector centre (a hit is nearer than the 64 m the device is
           given as its far limit, and no ray reaches another case's box), and the direction handed to the reference is
           the normalised one, computed with the device's arithmetic: dir * (1 / sqrt(len^2)) in float32."""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden", "ray_aabb_ref.npz")
LIB = os.path.join(HERE, "_ref", "libsc_ref_editor.so")
DRV = os.path.join(HERE, "_ref", "ref_ray_driver")
SECTOR = 64.0


def run_reference(origin, direction, bmin, bmax):
    rec = np.ascontiguousarray(np.concatenate([origin, direction, bmin, bmax], axis=1), dtype=np.float32)
    out = subprocess.run([DRV, LIB], input=rec.tobytes(), capture_output=True, check=True)
    r = np.frombuffer(out.stdout, dtype=[("hit", "<i4"), ("t", "<f4")])
    assert len(r) == len(rec)
    return r["hit"].astype(np.uint8), r["t"].copy()


def normalise_f32(d):
    """the device's (and PhysicsWorld::raycast's) normalisation in float32: dir * (1 / sqrt(x*x + y*y + z*z)), left to right"""
    d = d.astype(np.float32)
    len_sq = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(np.float32) + (d[:, 2] * d[:, 2]).astype(np.float32)
    inv = (np.float32(1.0) / np.sqrt(len_sq.astype(np.float32))).astype(np.float32)
    return (d * inv[:, None]).astype(np.float32)


def group_a(rng):
    n = 448
    o = rng.uniform(-20, 20, (n, 3)).astype(np.float32)
    d = rng.normal(0, 1, (n, 3)).astype(np.float32)
    c = rng.uniform(-10, 10, (n, 3)).astype(np.float32)
    h = rng.uniform(0.1, 6, (n, 3)).astype(np.float32)
    mn, mx = (c - h).astype(np.float32), (c + h).astype(np.float32)
    # rays aimed at their box (most of the random ones miss)
    aim = slice(0, 160)
    tgt = (c[aim] + rng.uniform(-1, 1, (160, 3)).astype(np.float32) * h[aim]).astype(np.float32)
    d[aim] = (tgt - o[aim]).astype(np.float32)
    # components around the "parallel" threshold |dir| < 1e-6 (exact zeros, signed zeros, just below, exactly at, just above)
    thr = [0.0, -0.0, 5e-7, -5e-7, 9.9999e-7, 1e-6, -1e-6, 1.0000001e-6, 2e-6, -2e-6]
    for k, i in enumerate(range(160, 250)):
        d[i, k % 3] = np.float32(thr[k % len(thr)])
        if k % 2:                                   # the origin inside the slab of that axis, or outside it
            o[i, k % 3] = c[i, k % 3]
    for k, i in enumerate(range(250, 280)):         # two parallel axes: an axis-aligned ray
        d[i] = 0.0
        d[i, k % 3] = np.float32(1.0 if k % 2 else -1.0)
        if k % 3:
            o[i] = c[i]
            o[i, k % 3] = c[i, k % 3] - np.float32(15.0 if k % 2 else -15.0)
    o[280:310] = (c[280:310] + rng.uniform(-0.9, 0.9, (30, 3)).astype(np.float32) * h[280:310]).astype(np.float32)   # origin inside
    for k, i in enumerate(range(310, 340)):         # origin exactly on a face, direction parallel to it, into it, out of it
        ax = k % 3
        o[i] = c[i]
        o[i, ax] = mn[i, ax] if k % 2 else mx[i, ax]
        d[i] = rng.normal(0, 1, 3).astype(np.float32)
        d[i, ax] = np.float32([0.0, 1.0, -1.0][(k // 3) % 3])
    for k, i in enumerate(range(340, 370)):         # grazing: along an edge / through a corner
        ax = k % 3
        o[i] = mn[i]
        o[i, ax] = mn[i, ax] - np.float32(7.0)
        d[i] = 0.0
        d[i, ax] = np.float32(1.0)
        if k % 2:
            d[i, (ax + 1) % 3] = np.float32(1e-7)
    mn[370:390], mx[370:390] = mx[370:390].copy(), mn[370:390].copy()        # inverted boxes
    mx[390:400, 0] = np.inf                                                   # half-infinite boxes
    mn[400:410, 1] = -np.inf
    o[410:425] *= np.float32(1e4); c2 = (c[410:425] * np.float32(1e4)).astype(np.float32)
    mn[410:425], mx[410:425] = (c2 - h[410:425] * 100).astype(np.float32), (c2 + h[410:425] * 100).astype(np.float32)
    d[410:425] = (c2 - o[410:425]).astype(np.float32)
    o[425:448] *= np.float32(1e-3); mn[425:448] *= np.float32(1e-3); mx[425:448] *= np.float32(1e-3)
    d[425:448] = ((mn[425:448] + mx[425:448]) * np.float32(0.5) - o[425:448]).astype(np.float32)
    d[432:440] *= np.float32(-1.0)                                            # pointing away
    return o, d, mn, mx


def group_b(rng):
    n = 512
    q = np.float32(1.0 / 64.0)
    centre = np.zeros((n, 3), np.float32)
    centre[:, 0] = (np.arange(n, dtype=np.float32) * np.float32(SECTOR) + np.float32(32.0))
    centre[:, 1] = np.float32(2.0)
    centre[:, 2] = np.float32(32.0)
    # dyadic centre offsets and half extents: (c - h, c + h) and back through ((mn + mx) / 2, (mx - mn) / 2) are exact in float32
    off = (rng.integers(-256, 257, (n, 3)).astype(np.float32) * q).astype(np.float32)           # +-4 m
    h = (rng.integers(8, 257, (n, 3)).astype(np.float32) * q).astype(np.float32)                # 0.125 .. 4 m
    c = (centre + off).astype(np.float32)
    mn, mx = (c - h).astype(np.float32), (c + h).astype(np.float32)
    o = (centre + rng.uniform(-9, 9, (n, 3)).astype(np.float32)).astype(np.float32)
    raw = rng.normal(0, 1, (n, 3)).astype(np.float32)
    aim = slice(0, 320)
    tgt = (c[aim] + rng.uniform(-1.05, 1.05, (320, 3)).astype(np.float32) * h[aim]).astype(np.float32)
    raw[aim] = (tgt - o[aim]).astype(np.float32)
    for k, i in enumerate(range(320, 400)):         # axis-aligned and nearly axis-aligned rays (components below the threshold after normalisation)
        ax = k % 3
        raw[i] = 0.0
        raw[i, ax] = np.float32(3.0 if k % 2 else -2.0)
        if k % 4 >= 2:
            raw[i, (ax + 1) % 3] = np.float32(1e-6)
        o[i] = c[i]
        o[i, ax] = c[i, ax] - np.float32(8.0 if k % 2 else -8.0)
        if k % 5 == 0:
            o[i, (ax + 2) % 3] = mx[i, (ax + 2) % 3]          # sliding along a face
        if k % 7 == 0:
            o[i, (ax + 2) % 3] = mx[i, (ax + 2) % 3] + np.float32(0.5)   # just beside the box
    o[400:440] = (c[400:440] + rng.uniform(-0.9, 0.9, (40, 3)).astype(np.float32) * h[400:440]).astype(np.float32)       # origin inside
    raw[440:470] = (o[440:470] - c[440:470]).astype(np.float32)                                                          # pointing away
    d = normalise_f32(raw)
    return o, raw, d, mn, mx


def main():
    if not (os.path.exists(LIB) and os.path.exists(DRV)):
        sys.exit("oracle/_ref/libsc_ref_editor.so or ref_ray_driver missing: run `make -C oracle ref` where /root/reference exists")
    rng = np.random.default_rng(20261005)
    ao, ad, amn, amx = group_a(rng)
    ahit, at = run_reference(ao, ad, amn, amx)
    bo, braw, bd, bmn, bmx = group_b(rng)
    bhit, bt = run_reference(bo, bd, bmn, bmx)
    assert np.all(bt[bhit == 1] < 60.0), "a group-b hit lies beyond the far limit the device test uses"
    np.savez_compressed(OUT, a_origin=ao, a_dir=ad, a_min=amn, a_max=amx, a_hit=ahit, a_t=at,
                        b_origin=bo, b_dir_raw=braw, b_dir=bd, b_min=bmn, b_max=bmx, b_hit=bhit, b_t=bt)
    print(f"wrote {OUT}: group a {len(ao)} cases ({int(ahit.sum())} hits), group b {len(bo)} cases ({int(bhit.sum())} hits)")


if __name__ == "__main__":
    main()
