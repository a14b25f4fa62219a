#!/usr/bin/env python3
"""Generate tests/golden/ fixtures from the REAL reference code (oracle/_ref/libsc_ref.so).

TEST INFRASTRUCTURE ONLY.  Run in a container where /root/reference exists:

    make -C oracle ref && python oracle/make_golden.py

libsc_ref.so is the reference's own src/core/src/sc_math.cpp compiled unmodified plus
oracle/ref_harness.cpp over its header-only ECS types.  The fixtures are DATA (inputs and the
reference's outputs); they travel to the GPU box, the reference does not.
"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden")
F = C.POINTER(C.c_float)


def fp(a):
    return a.ctypes.data_as(F)


def main():
    lib_path = os.path.join(HERE, "_ref", "libsc_ref.so")
    if not os.path.exists(lib_path):
        sys.exit("oracle/_ref/libsc_ref.so missing: run `make -C oracle ref` where /root/reference exists")
    ref = C.CDLL(lib_path)
    ref.ref_mat4_rotation_xyz.argtypes = [C.c_float, C.c_float, C.c_float, F]
    ref.ref_mat4_perspective_rh_zo.argtypes = [C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, F]
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(20261004)
    n = 384

    # ---- mat4_mul: general, affine, sparse (zeros / signed zeros / ones) operands
    a = rng.uniform(-4, 4, (n, 16)).astype(np.float32)
    b = rng.uniform(-4, 4, (n, 16)).astype(np.float32)
    a[64:128, [3, 7, 11]] = 0.0
    a[64:128, 15] = 1.0
    b[64:128, [3, 7, 11]] = 0.0
    b[64:128, 15] = 1.0
    sparse = rng.integers(0, 4, (64, 16))
    a[128:192] = np.where(sparse == 0, 0.0, np.where(sparse == 1, -0.0, a[128:192])).astype(np.float32)
    b[128:192] = np.where(sparse.T[:, :64].T == 2, 1.0, b[128:192]).astype(np.float32)
    a[192:256] *= np.float32(1e4)
    b[256:320] *= np.float32(1e-4)
    mul = np.zeros((n, 16), np.float32)
    for i in range(n):
        ref.ref_mat4_mul(fp(a[i]), fp(b[i]), fp(mul[i]))

    # ---- rotation / trs: random, axis-aligned, zero, large angles, negative and zero scales
    pos = rng.uniform(-2000, 2000, (n, 3)).astype(np.float32)
    rot = rng.uniform(-np.pi, np.pi, (n, 3)).astype(np.float32)
    scl = rng.uniform(0.1, 4.0, (n, 3)).astype(np.float32)
    rot[0:32] = 0.0
    rot[32:96, 0] = 0.0
    rot[32:96, 2] = 0.0                                   # yaw-only, as SynthWorld props
    rot[96:128] = (rng.integers(-4, 5, (32, 3)) * (np.pi / 2)).astype(np.float32)
    rot[128:160] *= np.float32(100.0)                     # large arguments
    scl[160:192] *= -1.0
    scl[192:200] = 0.0
    scl[200:208, 1] = 0.0
    pos[208:224] = 0.0
    rotm = np.zeros((n, 16), np.float32)
    trs = np.zeros((n, 16), np.float32)
    for i in range(n):
        ref.ref_mat4_rotation_xyz(rot[i, 0], rot[i, 1], rot[i, 2], fp(rotm[i]))
        ref.ref_mat4_trs(fp(pos[i]), fp(rot[i]), fp(scl[i]), fp(trs[i]))

    # ---- inverse: TRS matrices (camera-like), general, near-singular, singular
    inv_in = trs.copy()
    inv_in[224:288] = rng.uniform(-2, 2, (64, 16)).astype(np.float32)
    inv_in[288:320] = 0.0
    inv_in[320:352] = (rng.uniform(-1, 1, (32, 16)) * 1e-3).astype(np.float32)
    inv_in[352:384, 0:4] = inv_in[352:384, 4:8]           # two equal columns -> det ~ 0
    inv = np.zeros((n, 16), np.float32)
    for i in range(n):
        ref.ref_mat4_inverse(fp(inv_in[i]), fp(inv[i]))

    # ---- perspective: valid and rejected parameter sets
    persp_in = np.stack([
        rng.uniform(0.2, 2.8, n), rng.uniform(0.3, 3.0, n),
        rng.uniform(0.01, 2.0, n), rng.uniform(10.0, 5000.0, n)], axis=1).astype(np.float32)
    persp_in[0] = [60.0 * 3.1415926535 / 180.0, 16.0 / 9.0, 0.1, 1000.0]
    persp_in[1] = [0.0, 1.0, 0.1, 10.0]
    persp_in[2] = [1.0, 0.0, 0.1, 10.0]
    persp_in[3] = [1.0, 1.0, 0.0, 10.0]
    persp_in[4] = [1.0, 1.0, 5.0, 5.0]
    flip = (np.arange(n) % 2).astype(np.int32)
    flip[0] = 1
    persp = np.zeros((n, 16), np.float32)
    for i in range(n):
        ref.ref_mat4_perspective_rh_zo(persp_in[i, 0], persp_in[i, 1], persp_in[i, 2], persp_in[i, 3],
                                       int(flip[i]), fp(persp[i]))

    # ---- chained products the path actually performs: world = parent * local, viewProj = proj * inverse(cam)
    chain = np.zeros((n, 16), np.float32)
    for i in range(n):
        ref.ref_mat4_mul(fp(trs[(i * 7 + 3) % n]), fp(trs[i]), fp(chain[i]))

    np.savez_compressed(os.path.join(OUT, "sc_math_ref.npz"),
                        mul_a=a, mul_b=b, mul_out=mul,
                        trs_pos=pos, trs_rot=rot, trs_scale=scl, rot_out=rotm, trs_out=trs,
                        inv_in=inv_in, inv_out=inv,
                        persp_in=persp_in, persp_flip=flip, persp_out=persp,
                        chain_out=chain)

    # ---- sparse-set pool semantics + entity packing + layout, from the reference's own headers
    scripts = []
    for s in range(24):
        m = int(rng.integers(4, 40))
        alive, ops = [], []
        for _ in range(int(rng.integers(8, 120))):
            if alive and rng.random() < 0.4:
                v = alive.pop(int(rng.integers(0, len(alive))))
                ops.append(-(v + 1))
            else:
                v = int(rng.integers(0, m))
                ops.append(v + 1)
                if v not in alive:
                    alive.append(v)
        arr = np.asarray(ops, np.int32)
        out = np.zeros(256, np.uint32)
        cnt = ref.ref_pool_script(arr.ctypes.data_as(C.POINTER(C.c_int32)), len(ops),
                                  out.ctypes.data_as(C.POINTER(C.c_uint32)), 256)
        scripts.append({"ops": ops, "dense": [int(x) for x in out[:cnt]]})
    ref.ref_entity_pack.restype = C.c_uint32
    packs = [[i, g, int(ref.ref_entity_pack(i, g))] for i, g in
             [(0, 0), (1, 0), (5, 3), (0xFFFFFF, 0), (0xFFFFFF, 255), (0x1000000, 1), (12345, 256), (7, 511)]]
    lay = np.zeros(16, np.uint32)
    ref.ref_layout(lay.ctypes.data_as(C.POINTER(C.c_uint32)))
    names = ["sizeof_Transform", "off_parent", "off_localPos", "off_localRot", "off_localScale", "off_worldMatrix",
             "off_dirty", "sizeof_Mat4", "sizeof_DrawItem", "off_DrawItem_model", "sizeof_Bounds", "sizeof_Plane",
             "sizeof_Frustum", "sizeof_RenderMesh", "sizeof_Entity", "sizeof_Camera"]
    parent = C.c_uint32()
    dirty = C.c_uint8()
    dp, dr, ds, dw = (np.zeros(3, np.float32), np.zeros(3, np.float32), np.zeros(3, np.float32), np.zeros(16, np.float32))
    ref.ref_default_transform(C.byref(parent), fp(dp), fp(dr), fp(ds), fp(dw), C.byref(dirty))
    with open(os.path.join(OUT, "sc_ecs_ref.json"), "w") as f:
        json.dump({
            "generated_by": "oracle/make_golden.py over oracle/_ref/libsc_ref.so (reference headers sc_ecs.h, sc_world_partition.h)",
            "pool_scripts": scripts,
            "entity_pack": packs,
            "layout": {k: int(v) for k, v in zip(names, lay)},
            "default_transform": {"parent": int(parent.value), "localPos": dp.tolist(), "localRot": dr.tolist(),
                                  "localScale": ds.tolist(), "worldMatrix": dw.tolist(), "dirty": int(dirty.value)},
        }, f, indent=1)
    print("wrote", os.path.normpath(OUT))


if __name__ == "__main__":
    main()
