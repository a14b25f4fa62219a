"""WorldTick -- thin Python host over the C ABI (libsc_tick.so) for tests and bench.py.

It mirrors what the C++ adapter systems do (sc_gameengine_amd/host/sc_tick_systems.cpp): push entity
state, set the frame's viewProj, run the tick, read results.  No CPU fallback, no oracle import.
"""
import ctypes as C

import numpy as np

from . import capi


def _f(a):
    return a.ctypes.data_as(capi.F32P)


def _u(a):
    return a.ctypes.data_as(capi.U32P)


def _c32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ---- host-side camera math (CameraSystem stays on the host, sc_ecs.cpp:213-272) ----------------
def host_mat4_trs(pos, rot, scale):
    p, r, s, o = _c32(pos), _c32(rot), _c32(scale), np.zeros(16, np.float32)
    assert capi.load().scTickHostMat4Trs(_f(p), _f(r), _f(s), _f(o))
    return o


def host_mat4_mul(a, b):
    a, b, o = _c32(a), _c32(b), np.zeros(16, np.float32)
    assert capi.load().scTickHostMat4Mul(_f(a), _f(b), _f(o))
    return o


def host_mat4_inverse(a):
    a, o = _c32(a), np.zeros(16, np.float32)
    assert capi.load().scTickHostMat4Inverse(_f(a), _f(o))
    return o


def host_mat4_perspective(fov, aspect, zn, zf, flip):
    o = np.zeros(16, np.float32)
    assert capi.load().scTickHostMat4PerspectiveRhZo(float(fov), float(aspect), float(zn), float(zf), int(flip), _f(o))
    return o


def host_camera_view_proj(camera_world, fov_y_deg=60.0, aspect=16.0 / 9.0, near=0.1, far=1000.0):
    m, o = _c32(camera_world), np.zeros(16, np.float32)
    assert capi.load().scTickHostCameraViewProj(_f(m), float(fov_y_deg), float(aspect), float(near), float(far), _f(o))
    return o


def camera_view_proj(cam):
    """viewProj of a root camera entity described by a SynthWorld camera dict."""
    world = host_mat4_trs(cam["pos"], cam["rot"], (1.0, 1.0, 1.0))
    return host_camera_view_proj(world, cam["fovY"], cam["aspect"], cam["nearZ"], cam["farZ"])


class WorldTick:
    def __init__(self, capacity, device=0, tile_origin=(0, 0), tile_sectors=(0, 0), sector_size=64.0,
                 max_pairs=0, max_draws=0):
        self.lib = capi.load()
        d = capi.ContextDesc()
        d.device_ordinal = device
        d.capacity = capacity
        d.tile_origin_x, d.tile_origin_z = tile_origin
        d.tile_sectors_x, d.tile_sectors_z = tile_sectors
        d.sector_size = sector_size
        d.max_pairs = max_pairs
        d.max_draws_budget = max_draws
        self.ctx = self.lib.scTickCreateContext(C.byref(d))
        if not self.ctx:
            raise capi.ScTickError("scTickCreateContext failed: " + (self.lib.scTickGetLastError(None) or b"").decode())
        self.capacity = capacity
        self.max_pairs = max_pairs if max_pairs else capacity * 4
        self.n = 0

    # ---- plumbing ----
    def _ok(self, r, what):
        if not r:
            raise capi.ScTickError(f"{what}: " + (self.lib.scTickGetLastError(self.ctx) or b"").decode())

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.scTickDestroyContext(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- state ----
    @classmethod
    def from_world(cls, w, device=0, broadphase=True, max_pairs=0, max_draws=0, capacity=None):
        t = cls(capacity or max(w.n, 1), device=device,
                tile_origin=w.origin if broadphase else (0, 0),
                tile_sectors=w.sectors if broadphase else (0, 0),
                max_pairs=max_pairs, max_draws=max_draws)
        t.upload_world(w)
        return t

    def upload_world(self, w):
        self.set_count(w.n)
        if w.n == 0:
            return
        self.upload_locals(0, w.pos, w.rot, w.scale)
        self.upload_bounds(0, w.bmin, w.bmax, w.has_bounds)
        self.upload_render_meshes(0, w.has_mesh, w.mesh, w.material)
        self.upload_layers(0, w.group, w.mask)
        self.set_topology(w.parent)
        if getattr(w, "mover_kind", None) is not None:
            self.upload_movers(0, w.mover_kind, w.mover_vel, w.mover_lo, w.mover_hi)
        if getattr(w, "is_agent", None) is not None:
            self.set_lane_graph(w.lane_graph)
            self.upload_traffic_agents(0, w.is_agent, w.agent_lane, w.agent_s, w.agent_speed, w.agent_mode)

    def set_count(self, n):
        self._ok(self.lib.scTickSetEntityCount(self.ctx, n), "scTickSetEntityCount")
        self.n = n

    def upload_locals(self, first, pos, rot, scale):
        p, r, s = _c32(pos), _c32(rot), _c32(scale)
        rep = np.zeros(len(p), np.uint8)
        self._ok(self.lib.scTickUploadLocals(self.ctx, first, len(p), _f(p), _f(r), _f(s), rep.ctypes.data_as(capi.U8P)),
                 "scTickUploadLocals")
        return rep

    def upload_positions(self, first, pos):
        p = _c32(pos)
        self._ok(self.lib.scTickUploadPositions(self.ctx, first, len(p), _f(p)), "scTickUploadPositions")

    def upload_bounds(self, first, bmin, bmax, has=None):
        a, b = _c32(bmin), _c32(bmax)
        h = None if has is None else np.ascontiguousarray(has, np.uint8)
        self._ok(self.lib.scTickUploadBounds(self.ctx, first, len(a), _f(a), _f(b),
                                             None if h is None else h.ctypes.data_as(capi.U8P)), "scTickUploadBounds")

    def upload_render_meshes(self, first, has, mesh, material):
        h = np.ascontiguousarray(has, np.uint8)
        m, t = np.ascontiguousarray(mesh, np.uint32), np.ascontiguousarray(material, np.uint32)
        self._ok(self.lib.scTickUploadRenderMeshes(self.ctx, first, len(h), h.ctypes.data_as(capi.U8P), _u(m), _u(t)),
                 "scTickUploadRenderMeshes")

    def upload_layers(self, first, group, mask):
        g, m = np.ascontiguousarray(group, np.uint32), np.ascontiguousarray(mask, np.uint32)
        self._ok(self.lib.scTickUploadLayers(self.ctx, first, len(g), _u(g), _u(m)), "scTickUploadLayers")

    def set_topology(self, parent):
        p = np.ascontiguousarray(parent, np.int32)
        self._ok(self.lib.scTickSetTopology(self.ctx, p.ctypes.data_as(capi.I32P), len(p)), "scTickSetTopology")

    def mark_dirty(self, first, count):
        self._ok(self.lib.scTickMarkDirty(self.ctx, first, count), "scTickMarkDirty")

    def mark_dirty_indices(self, idx):
        i = np.ascontiguousarray(idx, np.uint32)
        self._ok(self.lib.scTickMarkDirtyIndices(self.ctx, _u(i), len(i)), "scTickMarkDirtyIndices")

    def upload_world_matrices(self, first, m16):
        m = _c32(m16)
        self._ok(self.lib.scTickUploadWorldMatrices(self.ctx, first, len(m), _f(m)), "scTickUploadWorldMatrices")

    # ---- frame inputs ----
    def set_view_proj(self, vp):
        v = _c32(vp)
        self._ok(self.lib.scTickSetViewProj(self.ctx, _f(v)), "scTickSetViewProj")

    def set_frustum_planes(self, planes, valid=True):
        p = _c32(planes).reshape(24)
        self._ok(self.lib.scTickSetFrustumPlanes(self.ctx, _f(p), 1 if valid else 0), "scTickSetFrustumPlanes")

    def frustum_planes(self):
        p, v = np.zeros(24, np.float32), C.c_int()
        self._ok(self.lib.scTickGetFrustumPlanes(self.ctx, _f(p), C.byref(v)), "scTickGetFrustumPlanes")
        return p.reshape(6, 4), v.value

    def set_camera(self, cam):
        vp = camera_view_proj(cam)
        self.set_view_proj(vp)
        return vp

    def set_draw_budget(self, max_draws):
        """WorldStreamingBudgets::maxDrawsBudget for the next SC_TICK_DRAWS (0 = unlimited)"""
        self._ok(self.lib.scTickSetDrawBudget(self.ctx, int(max_draws)), "scTickSetDrawBudget")

    def set_freeze_culling(self, on):
        self._ok(self.lib.scTickSetFreezeCulling(self.ctx, 1 if on else 0), "scTickSetFreezeCulling")

    # ---- tick ----
    def run(self, flags=capi.XFORM | capi.CULL):
        self._ok(self.lib.scTickRun(self.ctx, flags), "scTickRun")

    def sync(self):
        self._ok(self.lib.scTickSynchronize(self.ctx), "scTickSynchronize")

    def nudge_roots_x(self, dx):
        self._ok(self.lib.scTickNudgeRootsX(self.ctx, float(dx)), "scTickNudgeRootsX")

    # ---- ray queries over this tick's boxes ----
    def set_ray_queries(self, origin, direction, max_dist, mask):
        o, dd = _c32(origin).reshape(-1, 3), _c32(direction).reshape(-1, 3)
        md, mk = _c32(max_dist).reshape(-1), np.ascontiguousarray(mask, np.uint32).reshape(-1)
        self._ok(self.lib.scTickSetRayQueries(self.ctx, len(o), _f(o), _f(dd), _f(md), _u(mk)), "scTickSetRayQueries")

    def ray_hits(self):
        """Structured array (hit, id, distance, position[3], normal[3], layer) of the last run's ray batch."""
        n = C.c_uint32()
        self._ok(self.lib.scTickReadRayHits(self.ctx, None, 0, C.byref(n)), "scTickReadRayHits")
        buf = (capi.RayHit * max(n.value, 1))()
        if n.value:
            self._ok(self.lib.scTickReadRayHits(self.ctx, buf, n.value, C.byref(n)), "scTickReadRayHits")
        dt = np.dtype([("hit", np.uint32), ("id", np.uint32), ("distance", np.float32), ("position", np.float32, 3),
                       ("normal", np.float32, 3), ("layer", np.uint32), ("pad", np.uint32, 2)])
        return np.frombuffer(buf, dtype=dt, count=n.value).copy()

    def occupied(self, pos, radius, mask):
        """isOccupiedWorld for a batch of points (sc_traffic_spawner.cpp:93-116); returns a uint8 array."""
        p, r = _c32(pos).reshape(-1, 3), _c32(radius).reshape(-1)
        m = np.ascontiguousarray(mask, np.uint32).reshape(-1)
        out = np.zeros(len(p), np.uint8)
        self._ok(self.lib.scTickQueryOccupied(self.ctx, len(p), _f(p), _f(r), _u(m), out.ctypes.data_as(capi.U8P)), "scTickQueryOccupied")
        return out

    def set_draw_sort_table(self, pipeline_of_material, mesh_count):
        """Material::pipelineId per material handle (0xFF = no such material) and the number of mesh handles:
        what the renderer's filter + sort of the draw list reads (sc_vk.cpp:1842-1864)."""
        p = np.ascontiguousarray(pipeline_of_material, np.uint8)
        self._ok(self.lib.scTickSetDrawSortTable(self.ctx, p.ctypes.data_as(capi.U8P), len(p), int(mesh_count)), "scTickSetDrawSortTable")

    # ---- sector residency (WorldPartition::pumpCompletedLoads / pumpUnloadQueue) ----
    def append_entities(self, pos, rot, scale, bmin=None, bmax=None, mesh=None, material=None, group=None, mask=None, parent=None):
        """Create len(pos) entities at the end of the dense order; returns the first one's dense index."""
        p, r, s = _c32(pos).reshape(-1, 3), _c32(rot).reshape(-1, 3), _c32(scale).reshape(-1, 3)
        k = len(p)
        opt_f = [None if a is None else _c32(a) for a in (bmin, bmax)]
        opt_u = [None if a is None else np.ascontiguousarray(a, np.uint32) for a in (mesh, material, group, mask)]
        par = None if parent is None else np.ascontiguousarray(parent, np.int32)
        first = C.c_uint32()
        self._ok(self.lib.scTickAppendEntities(
            self.ctx, k, _f(p), _f(r), _f(s), *[None if a is None else _f(a) for a in opt_f],
            *[None if a is None else _u(a) for a in opt_u],
            None if par is None else par.ctypes.data_as(capi.I32P), C.byref(first)), "scTickAppendEntities")
        self.n += k
        return first.value

    def remove_entities(self, idx):
        """Swap-remove the entities at these dense indices, in this order; returns (moved_from, moved_to)."""
        i = np.ascontiguousarray(idx, np.uint32)
        src, dst, m = np.zeros(len(i), np.uint32), np.zeros(len(i), np.uint32), C.c_uint32()
        self._ok(self.lib.scTickRemoveEntities(self.ctx, _u(i), len(i), _u(src), _u(dst), C.byref(m)), "scTickRemoveEntities")
        self.n -= len(i)
        return src[:m.value].copy(), dst[:m.value].copy()

    def activate_sector(self, sector, resolve_mesh=None, resolve_material=None):
        """Spawn a parsed .scsector (sectors.SectorData) as pumpCompletedLoads does (sc_world_partition.cpp:916-958):
        one root entity per instance with setLocal, RenderMesh handles from the resolvers (asset id -> handle; the
        reference resolves id 0 to handle 0, :746-749), unit-cube Bounds.  Returns the dense indices."""
        rm = resolve_mesh or (lambda a: 0)
        rt = resolve_material or (lambda a: 0)
        mesh = np.array([0 if int(a) == 0 else rm(int(a)) for a in sector.mesh_id], np.uint32)
        mat = np.array([0 if int(a) == 0 else rt(int(a)) for a in sector.material_id], np.uint32)
        first = self.append_entities(sector.pos, sector.rot, sector.scale, mesh=mesh, material=mat)
        return np.arange(first, first + len(mesh), dtype=np.uint32)

    # ---- upstream movers ----
    def upload_movers(self, first, kind, vel, lo, hi):
        k = np.ascontiguousarray(kind, np.uint8)
        v, a, b = _c32(vel), _c32(lo), _c32(hi)
        self._ok(self.lib.scTickUploadMovers(self.ctx, first, len(k), k.ctypes.data_as(capi.U8P), _f(v), _f(a), _f(b)), "scTickUploadMovers")

    def advance_movers(self, dt):
        self._ok(self.lib.scTickAdvanceMovers(self.ctx, float(dt)), "scTickAdvanceMovers")

    # ---- on-rails traffic ----
    def set_lane_graph(self, g):
        """g: lanes.LaneGraph (or anything with the same arrays)"""
        a = {k: np.ascontiguousarray(getattr(g, k), dt) for k, dt in (
            ("seg_start", np.float32), ("seg_dir", np.float32), ("seg_length", np.float32), ("seg_active", np.uint8),
            ("seg_end_node", np.uint32), ("seg_speed_limit", np.float32), ("node_pos", np.float32),
            ("node_conn_offset", np.uint32), ("node_conn", np.uint32))}
        lg = capi.LaneGraph()
        lg.segments, lg.nodes, lg.connections = len(a["seg_length"]), len(a["node_pos"]), len(a["node_conn"])
        lg.seg_start3, lg.seg_dir3, lg.seg_length = _f(a["seg_start"]), _f(a["seg_dir"]), _f(a["seg_length"])
        lg.seg_active = a["seg_active"].ctypes.data_as(capi.U8P)
        lg.seg_end_node, lg.seg_speed_limit = _u(a["seg_end_node"]), _f(a["seg_speed_limit"])
        lg.node_pos3, lg.node_conn_offset, lg.node_conn = _f(a["node_pos"]), _u(a["node_conn_offset"]), _u(a["node_conn"])
        self._ok(self.lib.scTickSetLaneGraph(self.ctx, C.byref(lg)), "scTickSetLaneGraph")

    def set_lane_active(self, segment_ids, active):
        ids = np.ascontiguousarray(segment_ids, np.uint32)
        self._ok(self.lib.scTickSetLaneActive(self.ctx, _u(ids), len(ids), 1 if active else 0), "scTickSetLaneActive")

    def upload_traffic_agents(self, first, is_agent, lane_id, lane_s, target_speed, mode, look_ahead=None):
        ia, md = np.ascontiguousarray(is_agent, np.uint8), np.ascontiguousarray(mode, np.uint8)
        ln, ls, sp = np.ascontiguousarray(lane_id, np.uint32), _c32(lane_s), _c32(target_speed)
        la = None if look_ahead is None else _c32(look_ahead)
        self._ok(self.lib.scTickUploadTrafficAgents(self.ctx, first, len(ia), ia.ctypes.data_as(capi.U8P), _u(ln), _f(ls), _f(sp),
                                                    md.ctypes.data_as(capi.U8P), None if la is None else _f(la)), "scTickUploadTrafficAgents")

    def traffic_agents(self):
        """(lane_id, lane_s, target_speed, mode) of every entity (meaningful where the entity is an agent)"""
        ln, ls, sp, md = np.zeros(self.n, np.uint32), np.zeros(self.n, np.float32), np.zeros(self.n, np.float32), np.zeros(self.n, np.uint8)
        self._ok(self.lib.scTickReadTrafficAgents(self.ctx, 0, self.n, _u(ln), _f(ls), _f(sp), md.ctypes.data_as(capi.U8P)), "scTickReadTrafficAgents")
        return ln, ls, sp, md

    def set_traffic_sensors(self, on=True, front_ray_length=20.0, safe_distance=10.0):
        """the traffic AI's obstacle ray per OnRails agent, cast in every run with BROADPHASE; its brake scales the next on-rails step"""
        self._ok(self.lib.scTickSetTrafficSensors(self.ctx, 1 if on else 0, float(front_ray_length), float(safe_distance)), "scTickSetTrafficSensors")

    def upload_traffic_sensors(self, first, front_ray_length, safe_distance):
        """per-agent TrafficSensors::frontRayLength / safeDistance for the entities first .. first + len - 1"""
        rl, sf = _c32(front_ray_length).reshape(-1), _c32(safe_distance).reshape(-1)
        assert len(rl) == len(sf)
        self._ok(self.lib.scTickUploadTrafficSensors(self.ctx, first, len(rl), _f(rl), _f(sf)), "scTickUploadTrafficSensors")

    def traffic_sensors(self):
        """(lastHitDistance, lastHitType) per entity as the last ray tick left them (0 none, 2 vehicle, 3 world)"""
        dist, typ = np.zeros(self.n, np.float32), np.zeros(self.n, np.uint8)
        self._ok(self.lib.scTickReadTrafficSensors(self.ctx, 0, self.n, _f(dist), typ.ctypes.data_as(capi.U8P)), "scTickReadTrafficSensors")
        return dist, typ

    def traffic_brakes(self):
        out = np.zeros(self.n, np.float32)
        self._ok(self.lib.scTickReadTrafficBrakes(self.ctx, 0, self.n, _f(out)), "scTickReadTrafficBrakes")
        return out

    def set_traffic_speed_multiplier(self, m):
        self._ok(self.lib.scTickSetTrafficSpeedMultiplier(self.ctx, float(m)), "scTickSetTrafficSpeedMultiplier")

    def select_traffic_tiers(self, player_pos, a_enter=50.0, a_exit=70.0, b_enter=110.0, b_exit=150.0, max_physics=24, max_kinematic=64):
        """TrafficLODSystem's tier selection; returns (physics, kinematic, on_rails) counts after the caps"""
        pp = _c32(player_pos)
        tp = capi.TierParams(a_enter, a_exit, b_enter, b_exit, max_physics, max_kinematic)
        out = capi.TierCounts()
        self._ok(self.lib.scTickSelectTrafficTiers(self.ctx, _f(pp), C.byref(tp), C.byref(out)), "scTickSelectTrafficTiers")
        return out.physics, out.kinematic, out.on_rails

    def select_traffic_despawns(self, player_pos, max_total):
        """TrafficLODSystem's total cap: dense indices of the vehicles to despawn, in the reference's order"""
        pp = _c32(player_pos)
        cnt = C.c_uint32()
        self._ok(self.lib.scTickSelectTrafficDespawns(self.ctx, _f(pp), int(max_total), None, 0, C.byref(cnt)), "scTickSelectTrafficDespawns")
        out = np.zeros(max(cnt.value, 1), np.uint32)
        if cnt.value:
            self._ok(self.lib.scTickSelectTrafficDespawns(self.ctx, _f(pp), int(max_total), _u(out), cnt.value, C.byref(cnt)), "scTickSelectTrafficDespawns")
        return out[:cnt.value].copy()

    def set_frame_producer(self, kind, param=0.0):
        """0 none, 1 nudge roots by `param`, 2 advance movers by dt=`param`: run() then is the whole frame"""
        self._ok(self.lib.scTickSetFrameProducer(self.ctx, int(kind), float(param)), "scTickSetFrameProducer")

    def mover_velocities(self):
        out = np.zeros((self.n, 2), np.float32)
        self._ok(self.lib.scTickReadMoverVelocities(self.ctx, 0, self.n, _f(out)), "scTickReadMoverVelocities")
        return out

    # ---- multi-GPU tiles ----
    def set_tile(self, rank, neighbour_mask):
        self._ok(self.lib.scTickSetTile(self.ctx, rank, neighbour_mask), "scTickSetTile")

    def set_tile_grid(self, tile_x, tile_z, tiles_x, tiles_z):
        self._ok(self.lib.scTickSetTileGrid(self.ctx, tile_x, tile_z, tiles_x, tiles_z), "scTickSetTileGrid")

    def set_border_capacity(self, records_per_ring_sector):
        """records per ring sector of a side (on average) a border message can carry; every tile of a world alike, before comm_init / binding"""
        self._ok(self.lib.scTickSetBorderCapacity(self.ctx, int(records_per_ring_sector)), "scTickSetBorderCapacity")

    def border_bytes(self, direction):
        return int(self.lib.scTickBorderBytes(self.ctx, direction))

    def bind_border(self, direction, send_ptr, recv_ptr):
        self._ok(self.lib.scTickBindBorderBuffers(self.ctx, direction, send_ptr, recv_ptr), "scTickBindBorderBuffers")

    def set_pairs_stream(self, hip_stream):
        """Pipelined tiles: merge + queries + pair search of a tick go to this stream (None / 0 switches it off)."""
        self._ok(self.lib.scTickSetPairsStream(self.ctx, C.c_void_p(hip_stream or None)), "scTickSetPairsStream")

    def bind_border_parity(self, parity, direction, send_ptr, recv_ptr):
        self._ok(self.lib.scTickBindBorderBuffersParity(self.ctx, parity, direction, C.c_void_p(send_ptr), C.c_void_p(recv_ptr)),
                 "scTickBindBorderBuffersParity")

    def border_buffer(self, parity, direction, recv=False):
        """device address of the message buffer bound for (tick parity, direction); 0 = none"""
        return int(self.lib.scTickGetBorderBuffer(self.ctx, parity, direction, 1 if recv else 0) or 0)

    def run_pairs(self):
        self._ok(self.lib.scTickRunPairs(self.ctx), "scTickRunPairs")

    # ---- library-owned exchange (RCCL communicator inside libsc_tick.so) ----
    def comm_init(self, unique_id, world_size, rank, peers=None):
        """unique_id: 128 bytes from capi.comm_unique_id() of rank 0, handed to every rank by the host's own channel.
        peers (optional): rank of the neighbour in each of the 8 directions, -1 = none (default: row-major tile order)."""
        if peers is not None:
            pr = np.ascontiguousarray(peers, np.int32)
            assert len(pr) == 8
            self._ok(self.lib.scTickCommSetPeers(self.ctx, pr.ctypes.data_as(capi.I32P)), "scTickCommSetPeers")
        uid = np.frombuffer(bytes(unique_id), np.uint8).copy()
        assert len(uid) == capi.COMM_ID_BYTES
        self._ok(self.lib.scTickCommInit(self.ctx, uid.ctypes.data_as(capi.U8P), int(world_size), int(rank)), "scTickCommInit")

    def comm_destroy(self):
        self._ok(self.lib.scTickCommDestroy(self.ctx), "scTickCommDestroy")

    def set_pipelined(self, on):
        """False / 0 = off, True / 1 = on (default depth 4), 2..4 = on with that many copies of the per-tick broadphase state"""
        self._ok(self.lib.scTickSetPipelined(self.ctx, int(on)), "scTickSetPipelined")

    def gather_visible_counts(self, max_ranks=128):
        """Every rank's visible count of the last tick (one all-gather over the library's communicator; every rank must call),
        this rank's offset in the global visible list and the list's length: (counts, offset, total).  SURVEY 8e, result assembly."""
        counts = np.zeros(max_ranks, np.uint32)
        off, tot = C.c_uint64(), C.c_uint64()
        self._ok(self.lib.scTickGatherVisibleCounts(self.ctx, counts.ctypes.data_as(capi.U32P), max_ranks, C.byref(off), C.byref(tot)), "scTickGatherVisibleCounts")
        ranks = max(self.comm_info()["world_size"], 1)
        return counts[:ranks].copy(), int(off.value), int(tot.value)

    def comm_info(self):
        """the exchange as the library sees it (communicator, peers, operations per group, host time per half of a step) as a dict"""
        ci = capi.CommInfo()
        self._ok(self.lib.scTickGetCommInfo(self.ctx, C.byref(ci)), "scTickGetCommInfo")
        return {k: (list(getattr(ci, k)) if k == "peer_rank" else getattr(ci, k)) for k, _ in capi.CommInfo._fields_}

    def set_world_layers(self, group, mask, known=True):
        """the tiled world's layer vocabulary: arrays (or ORs) of the group and mask words of every collider on ANY tile (scTickSetWorldLayers)"""
        g = int(np.bitwise_or.reduce(np.asarray(group, np.uint32).ravel())) if np.ndim(group) else int(group)
        m = int(np.bitwise_or.reduce(np.asarray(mask, np.uint32).ravel())) if np.ndim(mask) else int(mask)
        self._ok(self.lib.scTickSetWorldLayers(self.ctx, g, m, 1 if known else 0), "scTickSetWorldLayers")

    def bin_stats(self):
        """how the broadphase bins are filled: remembered slots, those written on every tick, whether the last tick could leave slots unwritten, learn ticks"""
        st = np.zeros(4, np.uint32)
        self._ok(self.lib.scTickGetBinStats(self.ctx, _u(st)), "scTickGetBinStats")
        return {"remembered_slots": int(st[0]), "written_every_tick": int(st[1]), "lazy_last_tick": bool(st[2] & 1),
                "unchanged_records_stay": bool(st[2] & 2), "pair_role_sweep_only": bool(st[2] & 4), "learn_ticks": int(st[3])}

    def learn_ticks(self):
        """learn ticks so far (host-side counter: no read-back)"""
        st = np.zeros(1, np.uint32)
        self._ok(self.lib.scTickGetLearnTicks(self.ctx, _u(st)), "scTickGetLearnTicks")
        return int(st[0])

    def reset_host_times(self):
        self._ok(self.lib.scTickResetHostTimes(self.ctx), "scTickResetHostTimes")

    def tile_step(self, flags):
        """One whole step of a tile: tick + pack, RCCL exchange, merge + pair search -- one call, nothing waits on the host."""
        self._ok(self.lib.scTickTileStep(self.ctx, flags), "scTickTileStep")

    def exchange_borders(self):
        self._ok(self.lib.scTickExchangeBorders(self.ctx), "scTickExchangeBorders")

    def set_stream(self, hip_stream, external=True):
        """external=True: run on the caller's hipStream_t (0 = the legacy default stream); False: own stream"""
        self._ok(self.lib.scTickSetStream(self.ctx, hip_stream, 1 if external else 0), "scTickSetStream")

    def set_profiling(self, period):
        """0 = off, n = record HIP events on every n-th tick"""
        self._ok(self.lib.scTickSetProfiling(self.ctx, int(period)), "scTickSetProfiling")

    def set_profiling_kernels(self, kernels=None):
        """time only these kernels (capi.K_* indices; None = all): a launch timed by events costs ~6 us of gap on its queue"""
        mask = 0 if kernels is None else sum(1 << int(k) for k in kernels)
        self._ok(self.lib.scTickSetProfilingKernels(self.ctx, mask), "scTickSetProfilingKernels")

    def set_graph_mode(self, on):
        self._ok(self.lib.scTickSetGraphMode(self.ctx, 1 if on else 0), "scTickSetGraphMode")

    def kernel_times_ms(self, kernel):
        cnt = C.c_uint32()
        self._ok(self.lib.scTickGetKernelTimes(self.ctx, kernel, None, 0, C.byref(cnt)), "scTickGetKernelTimes")
        out = np.zeros(cnt.value, np.float32)
        if cnt.value:
            self._ok(self.lib.scTickGetKernelTimes(self.ctx, kernel, _f(out), cnt.value, C.byref(cnt)), "scTickGetKernelTimes")
        return out

    # ---- per-frame read-back, overlapped with the next tick ----
    def set_frame_readback(self, max_visible, max_draws=0):
        self._ok(self.lib.scTickSetFrameReadback(self.ctx, int(max_visible), int(max_draws)), "scTickSetFrameReadback")

    def acquire_frame(self, frames_back=0, copy=True):
        """Frame `frames_back` (0 latest, 1 the one before): (frame struct, visible indices, draw items as a (n, 80) byte array).
        With copy=False the arrays are views of the pinned buffer (valid until the second run after this frame's)."""
        fr = capi.Frame()
        self._ok(self.lib.scTickAcquireFrame(self.ctx, int(frames_back), C.byref(fr)), "scTickAcquireFrame")
        nv, nd = fr.visible_in_buffer, fr.draws_in_buffer
        vis = np.ctypeslib.as_array(fr.visible_indices, shape=(nv,)) if nv else np.zeros(0, np.uint32)
        draws = (np.ctypeslib.as_array(C.cast(fr.draws, capi.U8P), shape=(nd * 80,)).reshape(nd, 80) if nd else np.zeros((0, 80), np.uint8))
        return fr, (vis.copy() if copy else vis), (draws.copy() if copy else draws)

    # ---- results ----
    def counts(self):
        c = capi.Counts()
        self._ok(self.lib.scTickGetCounts(self.ctx, C.byref(c)), "scTickGetCounts")
        return c

    def _index_list(self, fn, name):
        cnt = C.c_uint32()
        out = np.zeros(max(self.n, 1), np.uint32)
        self._ok(fn(self.ctx, _u(out), len(out), C.byref(cnt)), name)
        return out[:cnt.value].copy()

    def visible(self):
        return self._index_list(self.lib.scTickReadVisible, "scTickReadVisible")

    def culled(self):
        return self._index_list(self.lib.scTickReadCulled, "scTickReadCulled")

    def visibility_bits(self):
        nw = (self.n + 63) // 64
        w = np.zeros(max(nw, 1), np.uint64)
        self._ok(self.lib.scTickReadVisibilityBits(self.ctx, w.ctypes.data_as(capi.U64P), len(w)), "scTickReadVisibilityBits")
        bits = np.unpackbits(w[:nw].view(np.uint8), bitorder="little")[:self.n]
        return bits.astype(np.uint8)

    def world_matrices(self, first=0, count=None):
        count = self.n - first if count is None else count
        out = np.zeros((count, 16), np.float32)
        self._ok(self.lib.scTickReadWorldMatrices(self.ctx, first, count, _f(out)), "scTickReadWorldMatrices")
        return out

    def world_matrices_indexed(self, idx):
        i = np.ascontiguousarray(idx, np.uint32)
        out = np.zeros((len(i), 16), np.float32)
        self._ok(self.lib.scTickReadWorldMatricesIndexed(self.ctx, _u(i), len(i), _f(out)), "scTickReadWorldMatricesIndexed")
        return out

    def dirty(self):
        out = np.zeros(self.n, np.uint8)
        self._ok(self.lib.scTickReadDirty(self.ctx, 0, self.n, out.ctypes.data_as(capi.U8P)), "scTickReadDirty")
        return out

    def positions(self):
        out = np.zeros((self.n, 3), np.float32)
        self._ok(self.lib.scTickReadPositions(self.ctx, 0, self.n, _f(out)), "scTickReadPositions")
        return out

    def world_aabbs(self):
        mn, mx = np.zeros((self.n, 3), np.float32), np.zeros((self.n, 3), np.float32)
        self._ok(self.lib.scTickReadWorldAabbs(self.ctx, 0, self.n, _f(mn), _f(mx)), "scTickReadWorldAabbs")
        return mn, mx

    def pairs(self, cap=None):
        cnt = C.c_uint32()
        self._ok(self.lib.scTickReadPairs(self.ctx, None, 0, C.byref(cnt)), "scTickReadPairs")
        n = min(cnt.value, self.max_pairs)              # the list holds at most max_pairs; cnt is the number found
        n = n if cap is None else min(cap, n)
        out = np.full((max(n, 1), 2), 0xFFFFFFFF, np.uint32)
        if n:
            self._ok(self.lib.scTickReadPairs(self.ctx, _u(out), n, C.byref(cnt)), "scTickReadPairs")
        out = out[:n]
        out = out[out[:, 1] != 0xFFFFFFFF]              # a full shard segment keeps fewer than were found
        return out.copy(), cnt.value

    def draws(self):
        cnt = C.c_uint32()
        self._ok(self.lib.scTickReadDraws(self.ctx, None, 0, C.byref(cnt)), "scTickReadDraws")
        buf = (capi.DrawItem * max(cnt.value, 1))()
        if cnt.value:
            self._ok(self.lib.scTickReadDraws(self.ctx, buf, cnt.value, C.byref(cnt)), "scTickReadDraws")
        n = cnt.value
        a = np.frombuffer(buf, dtype=np.uint8, count=80 * n).reshape(n, 80) if n else np.zeros((0, 80), np.uint8)
        idx = a[:, 0:4].copy().view(np.uint32).ravel()
        mesh = a[:, 4:8].copy().view(np.uint32).ravel()
        mat = a[:, 8:12].copy().view(np.uint32).ravel()
        model = a[:, 16:80].copy().view(np.float32).reshape(n, 16)
        return idx, mesh, mat, model
