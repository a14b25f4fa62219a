"""Second, independent restatement of the tick path in numpy.  TEST INFRASTRUCTURE ONLY.

Purpose: the reference ships no test or fixture for TransformSystem / CullingSystem and their
translation units cannot be built here, so oracle/sc_oracle.c is "parity unpinned" for them.  This
module restates the same reference text a second time, in a different shape (vectorised, level-order
instead of an explicit-stack DFS, float32 array ops instead of scalar C), so that a transcription
slip in either restatement shows up as a disagreement.  It is not a pin.

All arithmetic is float32 element-wise numpy (one rounding per operation, never fused); sin/cos/tan
come from the host libm through ctypes (numpy's own float32 sin/cos are different algorithms).
"""
import ctypes as C
import ctypes.util

import numpy as np

_libm = C.CDLL(ctypes.util.find_library("m") or "libm.so.6")
for _n in ("sinf", "cosf", "tanf"):
    getattr(_libm, _n).restype = C.c_float
    getattr(_libm, _n).argtypes = [C.c_float]

f32 = np.float32


def _vec_libm(name, a):
    fn = getattr(_libm, name)
    a = np.asarray(a, f32)
    return np.array([fn(float(v)) for v in a.ravel()], f32).reshape(a.shape)


def mat4_mul(a, b):
    """a, b: (..., 16) column-major; sc_math.cpp:52-68 order ((a0*b0 + a1*b1) + a2*b2) + a3*b3."""
    a = np.asarray(a, f32)
    b = np.asarray(b, f32)
    out = np.zeros(np.broadcast(a, b).shape, f32)
    for col in range(4):
        for row in range(4):
            s = a[..., row] * b[..., col * 4]
            s = s + a[..., 4 + row] * b[..., col * 4 + 1]
            s = s + a[..., 8 + row] * b[..., col * 4 + 2]
            s = s + a[..., 12 + row] * b[..., col * 4 + 3]
            out[..., col * 4 + row] = s
    return out


def _identity(n):
    m = np.zeros((n, 16), f32)
    m[:, [0, 5, 10, 15]] = 1
    return m


def mat4_trs(pos, rot, scale):
    """sc_math.cpp:100-142 for (N,3) arrays."""
    pos, rot, scale = (np.asarray(x, f32) for x in (pos, rot, scale))
    n = len(pos)
    cx, sx = _vec_libm("cosf", rot[:, 0]), _vec_libm("sinf", rot[:, 0])
    cy, sy = _vec_libm("cosf", rot[:, 1]), _vec_libm("sinf", rot[:, 1])
    cz, sz = _vec_libm("cosf", rot[:, 2]), _vec_libm("sinf", rot[:, 2])
    rx, ry, rz = _identity(n), _identity(n), _identity(n)
    rx[:, 5], rx[:, 6], rx[:, 9], rx[:, 10] = cx, sx, -sx, cx
    ry[:, 0], ry[:, 2], ry[:, 8], ry[:, 10] = cy, -sy, sy, cy
    rz[:, 0], rz[:, 1], rz[:, 4], rz[:, 5] = cz, sz, -sz, cz
    r = mat4_mul(mat4_mul(rz, ry), rx)
    s = np.zeros((n, 16), f32)
    s[:, 0], s[:, 5], s[:, 10], s[:, 15] = scale[:, 0], scale[:, 1], scale[:, 2], 1
    t = _identity(n)
    t[:, 12], t[:, 13], t[:, 14] = pos[:, 0], pos[:, 1], pos[:, 2]
    return mat4_mul(t, mat4_mul(r, s))


def transform_system(pos, rot, scale, parent, dirty, world):
    """sc_ecs.cpp:118-211 on dense arrays.  parent: dense index or -1, already validated.
    Returns (world', dirty', scale').  Entities in / below a cycle are left untouched."""
    pos, rot = np.asarray(pos, f32), np.asarray(rot, f32)
    scale = np.asarray(scale, f32).copy()
    parent = np.asarray(parent, np.int64)
    dirty = np.asarray(dirty, bool).copy()
    world = np.asarray(world, f32).copy()
    n = len(pos)

    zero = (scale == 0).all(axis=1)                      # :143-149
    scale[zero] = 1
    dirty |= zero

    depth = np.full(n, -1, np.int64)
    depth[parent < 0] = 0
    level = 0
    while True:                                          # reachability from the roots, level by level
        nxt = (depth < 0) & (parent >= 0) & (depth[np.maximum(parent, 0)] == level)
        if not nxt.any():
            break
        level += 1
        depth[nxt] = level

    local = mat4_trs(pos, rot, scale)
    node_dirty = np.zeros(n, bool)
    for lv in range(level + 1):
        idx = np.flatnonzero(depth == lv)
        if lv == 0:
            nd = dirty[idx]
            world[idx[nd]] = local[idx[nd]]
        else:
            p = parent[idx]
            nd = dirty[idx] | node_dirty[p]              # :184
            world[idx[nd]] = mat4_mul(world[p[nd]], local[idx[nd]])
        node_dirty[idx] = nd
        dirty[idx] = np.where(nd, False, dirty[idx])     # :201 (a clean node is already false)
    return world, dirty.astype(np.uint8), scale


def frustum_from_viewproj(m):
    """sc_world_partition.cpp:1071-1103 -> (6,4) planes."""
    m = np.asarray(m, f32)
    rows = [m[[0, 4, 8, 12]], m[[1, 5, 9, 13]], m[[2, 6, 10, 14]], m[[3, 7, 11, 15]]]
    planes = np.zeros((6, 4), f32)
    combos = [rows[3] + rows[0], rows[3] - rows[0], rows[3] + rows[1], rows[3] - rows[1], rows[3] + rows[2], rows[3] - rows[2]]
    for i, v in enumerate(combos):
        v = v.astype(f32)
        len_sq = f32(f32(v[0] * v[0]) + f32(v[1] * v[1])) + f32(v[2] * v[2])
        if len_sq > f32(1e-8):
            inv = f32(1.0) / np.sqrt(len_sq, dtype=f32)
            planes[i] = v * inv
    return planes


def cull(world, bmin, bmax, has_bounds, planes, valid=True):
    """computeWorldBoundsSphere + sphereInFrustum (:1105-1144) for (N,16) matrices -> uint8 mask."""
    m = np.asarray(world, f32)
    bmin, bmax = np.asarray(bmin, f32), np.asarray(bmax, f32)
    c = (bmin + bmax) * f32(0.5)
    e = (bmax - bmin) * f32(0.5)
    ctr = np.zeros((len(m), 3), f32)
    for r in range(3):
        ctr[:, r] = m[:, r] * c[:, 0] + m[:, 4 + r] * c[:, 1] + m[:, 8 + r] * c[:, 2] + m[:, 12 + r]
    sc = [np.sqrt(m[:, 4 * k] * m[:, 4 * k] + m[:, 4 * k + 1] * m[:, 4 * k + 1] + m[:, 4 * k + 2] * m[:, 4 * k + 2], dtype=f32)
          for k in range(3)]
    syz = np.where(sc[1] < sc[2], sc[2], sc[1])
    max_scale = np.where(sc[0] < syz, syz, sc[0])
    radius = np.sqrt(e[:, 0] * e[:, 0] + e[:, 1] * e[:, 1] + e[:, 2] * e[:, 2], dtype=f32) * max_scale
    vis = np.ones(len(m), bool)
    if valid:
        for p in np.asarray(planes, f32):
            d = p[0] * ctr[:, 0] + p[1] * ctr[:, 1] + p[2] * ctr[:, 2] + p[3]
            vis &= ~(d < -radius)
    vis |= ~np.asarray(has_bounds, bool)                 # no Bounds => visible (:1252-1263)
    return vis.astype(np.uint8)


def world_aabb(world, bmin, bmax):
    """Broadphase spec: centre = M*c, half = |M3x3|*e, min = c-h, max = c+h."""
    m = np.asarray(world, f32)
    bmin, bmax = np.asarray(bmin, f32), np.asarray(bmax, f32)
    c = (bmin + bmax) * f32(0.5)
    e = (bmax - bmin) * f32(0.5)
    mn, mx = np.zeros((len(m), 3), f32), np.zeros((len(m), 3), f32)
    for r in range(3):
        ctr = m[:, r] * c[:, 0] + m[:, 4 + r] * c[:, 1] + m[:, 8 + r] * c[:, 2] + m[:, 12 + r]
        h = np.abs(m[:, r]) * e[:, 0] + np.abs(m[:, 4 + r]) * e[:, 1] + np.abs(m[:, 8 + r]) * e[:, 2]
        mn[:, r], mx[:, r] = ctr - h, ctr + h
    return mn, mx
