"""SynthWorld v1 -- deterministic synthetic worlds for the tick path.

The RNG and the prop distributions follow the reference's procedural sector generator
(src/engine/world/sc_world_partition.cpp:34-62 mix32 / hashCoordSeed / rand01 / lerp and
:105-169 generateSectorSpawnsStatic; seed 424242 and 64 m sectors from src/sandbox/src/main.cpp:73-79),
evaluated in numpy with exact uint32 / float32 arithmetic and vectorised over sectors.

Entities are created sector by sector, z-major then x, inside a tile, tiles in row-major order, so
the Transform pool's dense order is tile-major (a GPU owns a contiguous dense range).

Hierarchy pattern (configs 2-5): inside a sector the props come in runs of four
[root, child of previous, child of previous, root] (depths 0,1,2,0).  A child draws three more
numbers from the sector's stream for its local position (U[-1,1], U[0,1], U[-1,1]).
"""
from dataclasses import dataclass, field

import numpy as np

SECTOR_SIZE = np.float32(64.0)
SEED = 424242
K_PI = np.float32(3.1415926535)

MESH_TRIANGLE, MESH_CUBE = 0, 1
MAT_UNLIT, MAT_CHECKER, MAT_TEST = 0, 1, 2
GROUP_DYNAMIC, GROUP_STATIC = 1, 2           # sc_physics.cpp:372-379
MASK_ALL, MASK_STATIC = 0xFFFFFFFF, 1

CONFIGS = {
    # name: (sectors_x, sectors_z, props per sector K, hierarchy)
    "config1": (8, 8, 15, False),        # 1 024 entities, all roots
    "config2": (64, 64, 24, True),       # 102 400
    "config3": (256, 256, 15, True),     # 1 048 576
    "config4": (512, 512, 15, True),     # 4 194 304 (2x2 tiles of 256x256)
}


def _u32(x):
    return np.asarray(x, dtype=np.uint32)


def mix32(x):
    x = _u32(x).copy()
    with np.errstate(over="ignore"):
        x ^= x >> np.uint32(16)
        x *= np.uint32(0x7FEB352D)
        x ^= x >> np.uint32(15)
        x *= np.uint32(0x846CA68B)
        x ^= x >> np.uint32(16)
    return x


def hash_coord_seed(seed, cx, cz):
    with np.errstate(over="ignore"):
        h = np.uint32(seed) ^ mix32(_u32(cx.astype(np.int64) & 0xFFFFFFFF) * np.uint32(73856093))
        h = h ^ mix32(_u32(cz.astype(np.int64) & 0xFFFFFFFF) * np.uint32(19349663))
        return mix32(h + np.uint32(0x9E3779B9))


def rand01(state):
    """Advances `state` in place; returns float32 in [0, 1]."""
    with np.errstate(over="ignore"):
        state[...] = mix32(state + np.uint32(0x6D2B79F5))
    return (state & np.uint32(0x00FFFFFF)).astype(np.float32) / np.float32(16777215.0)


def lerp(a, b, t):
    a = np.float32(a) if np.isscalar(a) else a
    b = np.float32(b) if np.isscalar(b) else b
    return (a + (b - a) * t).astype(np.float32)


@dataclass
class SynthWorld:
    pos: np.ndarray
    rot: np.ndarray
    scale: np.ndarray
    parent: np.ndarray           # int32 dense index, -1 = root
    bmin: np.ndarray
    bmax: np.ndarray
    has_mesh: np.ndarray
    has_bounds: np.ndarray
    mesh: np.ndarray
    material: np.ndarray
    group: np.ndarray
    mask: np.ndarray
    sector_of: np.ndarray        # (N, 2) int32 sector coordinate of each entity's sector
    origin: tuple = (0, 0)       # first sector (x, z)
    sectors: tuple = (0, 0)      # sectors_x, sectors_z
    camera: dict = field(default_factory=dict)
    # upstream movers (config 5): kind 0 none / 1 vehicle (wrap) / 2 ped (reflect), velocity and sector box in xz
    mover_kind: np.ndarray = None
    mover_vel: np.ndarray = None
    mover_lo: np.ndarray = None
    mover_hi: np.ndarray = None
    # on-rails traffic agents (laned config 5): TrafficAgent + TrafficVehicle per entity and the lane graph they follow
    is_agent: np.ndarray = None
    agent_lane: np.ndarray = None
    agent_s: np.ndarray = None
    agent_speed: np.ndarray = None
    agent_mode: np.ndarray = None
    lane_graph: object = None

    @property
    def n(self):
        return len(self.pos)

    @property
    def world_side(self):
        return float(self.sectors[0]) * float(SECTOR_SIZE)


def default_camera(world_side):
    """Camera of SURVEY 8d: centre of the world, 30 m up, looking down a little."""
    return {"pos": np.array([world_side / 2, 30.0, world_side / 2], np.float32),
            "rot": np.array([-0.3, 0.7, 0.0], np.float32),
            "fovY": 60.0, "nearZ": 0.1, "farZ": 1000.0, "aspect": 16.0 / 9.0}


def generate(sectors_x, sectors_z, props, hierarchy=True, origin=(0, 0), seed=SEED,
             tiles=(1, 1), ground=True):
    """Build a SynthWorld v1 of sectors_x x sectors_z sectors starting at sector `origin`.

    tiles=(tx, tz) splits the rectangle into tx x tz equal tiles and orders entities tile-major.
    """
    tx, tz = tiles
    assert sectors_x % tx == 0 and sectors_z % tz == 0
    sxt, szt = sectors_x // tx, sectors_z // tz
    cxs, czs = [], []
    for tzi in range(tz):
        for txi in range(tx):
            zz, xx = np.meshgrid(np.arange(szt, dtype=np.int32), np.arange(sxt, dtype=np.int32), indexing="ij")
            cxs.append((xx + txi * sxt + origin[0]).ravel())
            czs.append((zz + tzi * szt + origin[1]).ravel())
    cx = np.concatenate(cxs).astype(np.int32)
    cz = np.concatenate(czs).astype(np.int32)
    S = cx.size
    per = props + (1 if ground else 0)

    size = SECTOR_SIZE
    min_x = cx.astype(np.float32) * size
    min_z = cz.astype(np.float32) * size
    center_x = min_x + size * np.float32(0.5)
    center_z = min_z + size * np.float32(0.5)

    pos = np.zeros((S, per, 3), np.float32)
    rot = np.zeros((S, per, 3), np.float32)
    scl = np.ones((S, per, 3), np.float32)
    parent_local = np.full((S, per), -1, np.int32)       # index inside the sector, -1 = root
    mesh = np.full((S, per), MESH_CUBE, np.uint32)
    mat = np.full((S, per), MAT_UNLIT, np.uint32)

    state = hash_coord_seed(seed, cx, cz)
    o = 0
    if ground:                                            # sc_world_partition.cpp:122-136
        pos[:, 0, 0] = center_x
        pos[:, 0, 1] = np.float32(-0.55)
        pos[:, 0, 2] = center_z
        scl[:, 0, 0] = size
        scl[:, 0, 1] = np.float32(0.10)
        scl[:, 0, 2] = size
        o = 1

    pad = np.float32(1.0)
    for p in range(props):                                # sc_world_partition.cpp:139-168
        k = o + p
        x = lerp(min_x + pad, min_x + size - pad, rand01(state))
        z = lerp(min_z + pad, min_z + size - pad, rand01(state))
        sx = lerp(0.4, 1.9, rand01(state))
        sy = lerp(0.5, 3.2, rand01(state))
        sz = lerp(0.4, 1.9, rand01(state))
        yaw = (rand01(state) * (K_PI * np.float32(2.0))).astype(np.float32)
        m = rand01(state)
        q = rand01(state)
        pos[:, k, 0] = x
        pos[:, k, 1] = sy * np.float32(0.5)
        pos[:, k, 2] = z
        rot[:, k, 1] = yaw
        scl[:, k, 0] = sx
        scl[:, k, 1] = sy
        scl[:, k, 2] = sz
        mat[:, k] = np.where(m < np.float32(0.40), MAT_CHECKER, np.where(m < np.float32(0.80), MAT_TEST, MAT_UNLIT))
        mesh[:, k] = np.where(q < np.float32(0.90), MESH_CUBE, MESH_TRIANGLE)
        if hierarchy and (p % 4) in (1, 2):
            parent_local[:, k] = k - 1
            pos[:, k, 0] = lerp(-1.0, 1.0, rand01(state))
            pos[:, k, 1] = lerp(0.0, 1.0, rand01(state))
            pos[:, k, 2] = lerp(-1.0, 1.0, rand01(state))

    N = S * per
    base = (np.arange(S, dtype=np.int64) * per)[:, None]
    parent = np.where(parent_local >= 0, parent_local + base, -1).astype(np.int32).reshape(N)
    w = SynthWorld(
        pos=pos.reshape(N, 3), rot=rot.reshape(N, 3), scale=scl.reshape(N, 3), parent=parent,
        bmin=np.full((N, 3), -0.5, np.float32), bmax=np.full((N, 3), 0.5, np.float32),   # kUnitCubeBounds, :27
        has_mesh=np.ones(N, np.uint8), has_bounds=np.ones(N, np.uint8),
        mesh=mesh.reshape(N), material=mat.reshape(N),
        group=np.full(N, GROUP_STATIC, np.uint32), mask=np.full(N, MASK_STATIC, np.uint32),
        sector_of=np.repeat(np.stack([cx, cz], axis=1), per, axis=0).astype(np.int32),
        origin=tuple(origin), sectors=(sectors_x, sectors_z))
    w.camera = default_camera(w.world_side)
    return w


VEHICLES_PER_SECTOR, PEDS_PER_SECTOR = 12, 4
VEHICLE_SCALE = (1.8, 0.7, 3.5)          # src/engine/traffic/sc_traffic_spawner.h:20
VEHICLE_Y = 0.35                         # src/engine/traffic/sc_traffic_spawner.cpp:273
VEHICLE_SPEED = 12.0                     # src/engine/traffic/sc_traffic_lanes.h:17
LANE_OFFSET = 1.75                       # src/engine/traffic/sc_traffic_lanes.cpp:158-225 (two lanes per axis)
PED_SCALE, PED_Y, PED_SPEED = (0.5, 1.8, 0.5), 0.9, 1.4


def generate_config5(sectors_x, sectors_z, origin=(0, 0), seed=SEED, tiles=(1, 1), laned=False):
    """SynthWorld v1 config 5: per sector ground + 15 props (static, hierarchy as config 3)
    + 12 vehicles on the sector's four lanes + 4 peds = 32 entities; vehicles and peds are dynamic
    (group 1 / mask all) roots that the mover kernel advances every tick.

    laned=False: vehicles are straight-line movers that wrap inside their sector (the round-1 stand-in).
    laned=True: vehicles are the engine's own on-rails traffic agents (TrafficAgent on a lane of the procedural lane
    graph, spawned as sc_traffic_spawner.cpp:267-318 does: pos = lane start + dir * laneS, targetSpeed 0, tier OnRails);
    they follow their lane across sectors and park at the world's edge.  Peds keep the reflecting stand-in."""
    base = generate(sectors_x, sectors_z, 15, hierarchy=True, origin=origin, seed=seed, tiles=tiles)
    S = base.n // 16
    per_old, extra = 16, VEHICLES_PER_SECTOR + PEDS_PER_SECTOR
    per = per_old + extra
    cx = base.sector_of[::per_old, 0].astype(np.int32)
    cz = base.sector_of[::per_old, 1].astype(np.int32)
    size = SECTOR_SIZE
    min_x, min_z = cx.astype(np.float32) * size, cz.astype(np.float32) * size
    ctr_x, ctr_z = min_x + size * np.float32(0.5), min_z + size * np.float32(0.5)
    # a second, independent stream per sector for the agents (the prop stream stays what config 3 draws)
    state = hash_coord_seed(seed ^ 0x5EED5, cx, cz)

    def grow(a, fill, dt):
        out = np.empty((S, per) + a.shape[1:], dt)
        out[:, :per_old] = a.reshape((S, per_old) + a.shape[1:])
        out[:, per_old:] = fill
        return out

    pos, rot, scl = grow(base.pos, 0, np.float32), grow(base.rot, 0, np.float32), grow(base.scale, 1, np.float32)
    mesh, mat = grow(base.mesh, MESH_CUBE, np.uint32), grow(base.material, MAT_TEST, np.uint32)
    group, mask = grow(base.group, GROUP_DYNAMIC, np.uint32), grow(base.mask, MASK_ALL, np.uint32)
    kind = np.zeros((S, per), np.uint8)
    vel = np.zeros((S, per, 2), np.float32)
    pad = np.float32(2.0)
    graph = None
    is_agent = agent_lane = agent_s = None
    if laned:
        from . import lanes as lanes_mod
        graph = lanes_mod.build_procedural(cx, cz, float(size))
        is_agent = np.zeros((S, per), np.uint8)
        agent_lane = np.full((S, per), lanes_mod.INVALID_LANE, np.uint32)
        agent_s = np.zeros((S, per), np.float32)
    lanes = [((1, 0), None, -LANE_OFFSET), ((-1, 0), None, LANE_OFFSET), ((0, 1), LANE_OFFSET, None), ((0, -1), -LANE_OFFSET, None)]
    for v in range(VEHICLES_PER_SECTOR):
        k = per_old + v
        (dx, dz), offx, offz = lanes[v % 4]
        along = rand01(state)
        if dx != 0:                                       # lane along x at z = centre + offz
            pos[:, k, 0] = lerp(min_x + pad, min_x + size - pad, along)
            pos[:, k, 2] = ctr_z + np.float32(offz)
        else:
            pos[:, k, 0] = ctr_x + np.float32(offx)
            pos[:, k, 2] = lerp(min_z + pad, min_z + size - pad, along)
        pos[:, k, 1] = np.float32(VEHICLE_Y)
        rot[:, k, 1] = np.float32(np.arctan2(np.float32(dx), np.float32(dz)))      # yaw = atan2(dir.x, dir.z), sc_traffic_ai.cpp:72-75
        scl[:, k] = np.float32(VEHICLE_SCALE)
        if laned:
            seg = graph.sector_segments[:, v % 4]
            s_along = lerp(pad, size - pad, along)
            p0 = (graph.seg_start[seg] + graph.seg_dir[seg] * s_along[:, None]).astype(np.float32)       # sc_traffic_spawner.cpp:272-274
            pos[:, k, 0], pos[:, k, 2] = p0[:, 0], p0[:, 2]
            is_agent[:, k] = 1
            agent_lane[:, k] = seg
            agent_s[:, k] = s_along
            continue
        kind[:, k] = 1
        vel[:, k, 0], vel[:, k, 1] = np.float32(VEHICLE_SPEED * dx), np.float32(VEHICLE_SPEED * dz)
    for q in range(PEDS_PER_SECTOR):
        k = per_old + VEHICLES_PER_SECTOR + q
        pos[:, k, 0] = lerp(min_x + pad, min_x + size - pad, rand01(state))
        pos[:, k, 2] = lerp(min_z + pad, min_z + size - pad, rand01(state))
        pos[:, k, 1] = np.float32(PED_Y)
        heading = (rand01(state) * (K_PI * np.float32(2.0))).astype(np.float32)
        rot[:, k, 1] = heading
        scl[:, k] = np.float32(PED_SCALE)
        kind[:, k] = 2
        vel[:, k, 0] = (np.float32(PED_SPEED) * np.sin(heading)).astype(np.float32)
        vel[:, k, 1] = (np.float32(PED_SPEED) * np.cos(heading)).astype(np.float32)

    N = S * per
    parent_local = np.full((S, per), -1, np.int64)
    old_parent = base.parent.reshape(S, per_old).astype(np.int64)
    parent_local[:, :per_old] = np.where(old_parent >= 0, old_parent - (np.arange(S, dtype=np.int64) * per_old)[:, None], -1)
    parent = np.where(parent_local >= 0, parent_local + (np.arange(S, dtype=np.int64) * per)[:, None], -1).astype(np.int32).reshape(N)
    lo = np.repeat(np.stack([min_x, min_z], axis=1), per, axis=0).astype(np.float32)
    w = SynthWorld(
        pos=pos.reshape(N, 3), rot=rot.reshape(N, 3), scale=scl.reshape(N, 3), parent=parent,
        bmin=np.full((N, 3), -0.5, np.float32), bmax=np.full((N, 3), 0.5, np.float32),
        has_mesh=np.ones(N, np.uint8), has_bounds=np.ones(N, np.uint8),
        mesh=mesh.reshape(N), material=mat.reshape(N), group=group.reshape(N), mask=mask.reshape(N),
        sector_of=np.repeat(np.stack([cx, cz], axis=1), per, axis=0).astype(np.int32),
        origin=tuple(origin), sectors=(sectors_x, sectors_z),
        mover_kind=kind.reshape(N), mover_vel=vel.reshape(N, 2), mover_lo=lo, mover_hi=(lo + size).astype(np.float32))
    if laned:
        w.is_agent = is_agent.reshape(N)
        w.agent_lane = agent_lane.reshape(N)
        w.agent_s = agent_s.reshape(N)
        w.agent_speed = np.zeros(N, np.float32)                                  # sc_traffic_spawner.cpp:318
        w.agent_mode = np.full(N, 2, np.uint8)                                   # TrafficSimMode::OnRails, sc_traffic_common.h:42
        w.lane_graph = graph
    w.camera = default_camera(w.world_side)
    return w


def config(name, **kw):
    sx, sz, k, hier = CONFIGS[name]
    return generate(sx, sz, k, hierarchy=hier, **kw)


def roots(world):
    return np.flatnonzero(world.parent < 0).astype(np.uint32)


def with_extra_entity(world, pos, rot, scale=(1, 1, 1), parent=-1, has_mesh=False, has_bounds=False):
    """Append one entity (e.g. the camera: a Transform without RenderMesh / Bounds)."""
    def cat(a, row, dt):
        return np.concatenate([a, np.asarray([row], dt)], axis=0)
    return SynthWorld(
        pos=cat(world.pos, pos, np.float32), rot=cat(world.rot, rot, np.float32), scale=cat(world.scale, scale, np.float32),
        parent=np.concatenate([world.parent, np.asarray([parent], np.int32)]),
        bmin=cat(world.bmin, (0, 0, 0), np.float32), bmax=cat(world.bmax, (0, 0, 0), np.float32),
        has_mesh=np.concatenate([world.has_mesh, np.asarray([1 if has_mesh else 0], np.uint8)]),
        has_bounds=np.concatenate([world.has_bounds, np.asarray([1 if has_bounds else 0], np.uint8)]),
        mesh=np.concatenate([world.mesh, np.zeros(1, np.uint32)]), material=np.concatenate([world.material, np.zeros(1, np.uint32)]),
        group=np.concatenate([world.group, np.asarray([GROUP_STATIC], np.uint32)]),
        mask=np.concatenate([world.mask, np.asarray([0], np.uint32)]),
        sector_of=np.concatenate([world.sector_of, np.zeros((1, 2), np.int32)]),
        origin=world.origin, sectors=world.sectors, camera=world.camera)
