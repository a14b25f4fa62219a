"""The engine's own upstream mover on the device: on-rails traffic agents following the lane graph
(sc_traffic_ai.cpp:434-460, sc_traffic_lanes.cpp:291-352) and TrafficLODSystem's tier selection (sc_traffic_lod.cpp:323-417),
against the oracle's restatement on a laned config-5 world.  Lane state and positions are compared as bit patterns, world
matrices with IEEE equality (an agent's yaw only takes per-segment values whose sin / cos come from the host libm, so the
device matches without a trigonometric function of its own), visible lists element for element."""
import numpy as np
import pytest

from sc_gameengine_amd import capi, lanes, synth_world as sw
from sc_gameengine_amd.tick import WorldTick, camera_view_proj
from tests import worlds

pytestmark = pytest.mark.gpu
DT = 1.0 / 60.0


def laned_world(sx=24, sz=16, seed=3):
    w = sw.generate_config5(sx, sz, laned=True)
    rng = np.random.default_rng(seed)
    agents = np.flatnonzero(w.is_agent)
    # a few agents close to the end of their lane at the world's edge (they park on the dead end within the run), a few
    # already rolling, a few in the Physics / Kinematic tiers (not moved here)
    g = w.lane_graph
    pick = rng.choice(agents, 200, replace=False)
    w.agent_s[pick] = (g.seg_length[w.agent_lane[pick]] - rng.uniform(0.05, 3.0, 200)).astype(np.float32)
    p0 = (g.seg_start[w.agent_lane[pick]] + g.seg_dir[w.agent_lane[pick]] * w.agent_s[pick][:, None]).astype(np.float32)
    w.pos[pick, 0], w.pos[pick, 2] = p0[:, 0], p0[:, 2]
    fast = rng.choice(agents, 300, replace=False)
    w.agent_speed[fast] = rng.uniform(3.0, 20.0, 300).astype(np.float32)
    other = rng.choice(agents, 100, replace=False)
    w.agent_mode[other] = rng.integers(0, 2, 100).astype(np.uint8)
    return w


def oracle_side(oracle, w):
    ow = worlds.oracle_world(oracle, w, camera=False)
    ol = oracle.OracleLanes()
    ol.build_sectors(w.sector_of[::32, 0], w.sector_of[::32, 1])
    state = dict(lane=w.agent_lane.copy(), s=w.agent_s.copy(), speed=w.agent_speed.copy(), mode=w.agent_mode.copy(),
                 look=np.full(w.n, 12.0, np.float32), vel=w.mover_vel.copy())
    return ow, ol, state


def oracle_advance(ow, ol, w, st, dt, mult=1.0):
    ow.traffic_ai_onrails(ol, w.is_agent, st["lane"], st["s"], st["speed"], st["mode"], st["look"], dt, mult)
    ow.advance_movers(w.mover_kind, st["vel"], w.mover_lo, w.mover_hi, dt)


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def assert_agents_equal(t, w, st, ow):
    ln, ls, sp, md = t.traffic_agents()
    a = w.is_agent.astype(bool)
    assert np.array_equal(ln[a], st["lane"][a])
    assert np.array_equal(bits(ls[a]), bits(st["s"][a])) and np.array_equal(bits(sp[a]), bits(st["speed"][a]))
    assert np.array_equal(md[a], st["mode"][a])
    assert np.array_equal(bits(t.positions()), bits(ow.local_positions()[:w.n]))


def test_on_rails_agents_follow_their_lanes_60_ticks(oracle):
    w = laned_world()
    ow, ol, st = oracle_side(oracle, w)
    vp = camera_view_proj(w.camera)
    t = WorldTick.from_world(w, broadphase=True, max_pairs=1 << 19)
    t.set_view_proj(vp)
    start = w.pos.copy()
    for k in range(60):
        oracle_advance(ow, ol, w, st, DT)
        t.advance_movers(DT)
        ow.transform_system(); ow.culling_system(view_proj=vp)
        t.run(capi.FULL)
        if k % 6 == 5 or k < 3:
            assert_agents_equal(t, w, st, ow)
            assert np.array_equal(t.world_matrices(), ow.world_matrices()[:w.n]), f"tick {k}: world matrices differ"
            assert np.array_equal(t.visible(), ow.visible()), f"tick {k}: visible list differs"
            assert not t.dirty().any()
    a = w.is_agent.astype(bool) & (st["mode"] == 2)
    moved = np.abs(ow.local_positions()[:w.n] - start).max(axis=1)
    assert (moved[a] > 0).mean() > 0.95 and moved[a].max() > 6.0          # speeds approach 12 m/s: ~7 m in a second
    assert (moved[w.is_agent.astype(bool) & (st["mode"] != 2)] == 0).all()  # Physics / Kinematic agents are not the mover's
    g = w.lane_graph
    crossed = (st["lane"][a] != w.agent_lane[a]).sum()
    parked = (st["s"][a] == g.seg_length[st["lane"][a]]).sum()
    assert crossed > 100 and parked > 5                                    # lanes were crossed and dead ends reached
    # yaw only takes the segments' values
    yaw = ow.local_rotations()[:w.n][a, 1]
    allowed = np.float32([0.0, np.arctan2(np.float32(1), np.float32(0)), np.arctan2(np.float32(-1), np.float32(0)), np.arctan2(np.float32(0), np.float32(-1))])
    assert np.isin(yaw, allowed).all() and len(np.unique(yaw)) == 4
    t.close(); ow.close(); ol.close()


def test_traffic_as_fused_frame_producer_and_graph_replay(oracle):
    w = laned_world(16, 16, seed=5)
    ow, ol, st = oracle_side(oracle, w)
    vp = camera_view_proj(w.camera)
    for graph in (False, True):
        ow2, ol2, st2 = oracle_side(oracle, w)
        t = WorldTick.from_world(w, broadphase=True, max_pairs=1 << 19)
        t.set_view_proj(vp)
        t.set_frame_producer(2, DT)
        t.set_graph_mode(graph)
        t.advance_movers(DT)                                 # the first frame's producer, explicitly
        for k in range(12):
            oracle_advance(ow2, ol2, w, st2, DT)
            ow2.transform_system(); ow2.culling_system(view_proj=vp)
            t.run(capi.FULL | capi.PRODUCE_NEXT)             # tick, then the NEXT frame's producer inside the end-of-tick kernel
            assert np.array_equal(t.world_matrices(), ow2.world_matrices()[:w.n]), f"graph={graph} tick {k}"
            assert np.array_equal(t.visible(), ow2.visible())
        oracle_advance(ow2, ol2, w, st2, DT)                 # the device is one producer step ahead
        assert_agents_equal(t, w, st2, ow2)
        t.close(); ow2.close(); ol2.close()
    ow.close(); ol.close()


def test_inactive_lanes_speed_multiplier_and_invalid_lane(oracle):
    w = laned_world(12, 12, seed=7)
    agents = np.flatnonzero(w.is_agent)
    w.agent_lane[agents[:20]] = lanes.INVALID_LANE           # no lane: the agent takes the nearest active one first (sc_traffic_ai.cpp:264-272)
    ow, ol, st = oracle_side(oracle, w)
    t = WorldTick.from_world(w, broadphase=False)
    off = np.arange(40, 120, dtype=np.uint32)                # removeSector: these segments go inactive
    for s in off:
        ol.set_active(int(s), False)
    t.set_lane_active(off, False)
    t.set_traffic_speed_multiplier(1.7)
    for k in range(30):
        oracle_advance(ow, ol, w, st, DT, mult=1.7)
        t.advance_movers(DT)
        if k == 14:                                          # ... and back (buildProceduralForSector re-activates, :164-171)
            for s in off[:40]:
                ol.set_active(int(s), True)
            t.set_lane_active(off[:40], True)
    ow.transform_system()
    t.run(capi.XFORM)
    assert_agents_equal(t, w, st, ow)
    assert np.array_equal(t.world_matrices(), ow.world_matrices()[:w.n])
    a = w.is_agent.astype(bool)
    assert st["speed"][a].max() > 12.0                       # the multiplier raised the target speed above the limit
    assert (st["lane"][agents[:20]] != lanes.INVALID_LANE).all()          # queryNearestLane found every lost agent a lane ...
    assert (t.traffic_agents()[0][agents[:20]] != lanes.INVALID_LANE).all()   # ... on the device too (the same one: compared above)
    t.close(); ow.close(); ol.close()


def test_lost_agents_with_no_active_lane_stay_lost(oracle):
    """queryNearestLane over a graph without an active segment returns nothing: the agent keeps kInvalidLaneId and is skipped."""
    w = laned_world(8, 8, seed=13)
    agents = np.flatnonzero(w.is_agent)
    w.agent_lane[agents[::3]] = lanes.INVALID_LANE
    ow, ol, st = oracle_side(oracle, w)
    t = WorldTick.from_world(w, broadphase=False)
    every = np.arange(len(w.lane_graph.seg_length), dtype=np.uint32)
    for s in every:
        ol.set_active(int(s), False)
    t.set_lane_active(every, False)
    for k in range(3):
        oracle_advance(ow, ol, w, st, DT)
        t.advance_movers(DT)
    assert_agents_equal(t, w, st, ow)
    assert (t.traffic_agents()[0][agents[::3]] == lanes.INVALID_LANE).all()
    t.close(); ow.close(); ol.close()


@pytest.mark.parametrize("fused", [False, True])
def test_on_rails_agents_brake_for_what_their_front_ray_meets_60_ticks(oracle, fused):
    """The obstacle ray of TrafficAISystem (sc_traffic_ai.cpp:300-345) wired to the on-rails step (:436): every tick casts one
    ray per OnRails agent against the tick's boxes (own spec: world AABBs, the agent's own box excluded), the brake scales the
    desired speed of the step that follows.  The oracle does the same with a brute-force ray over every box.  Vehicles carry
    group 1 / mask all, so they brake for each other where the lanes bunch them up; brakes, lane state and positions are
    compared as bit patterns, matrices and visible lists as always.  fused: the step rides on the end-of-tick kernel."""
    w = laned_world(16, 12, seed=21)
    w.scale[w.is_agent.astype(bool), 1] = np.float32(2.0)    # SynthWorld's vehicles are 0.7 m tall at y = 0.35: the ray (0.6 m above the origin) would pass over them
    ow, ol, st = oracle_side(oracle, w)
    vp = camera_view_proj(w.camera)
    t = WorldTick.from_world(w, broadphase=True, max_pairs=1 << 20)
    t.set_view_proj(vp)
    t.set_traffic_sensors(True)                              # TrafficSensors defaults: 20 m ray, 10 m safe distance
    if fused:
        t.set_frame_producer(2, DT)
    a = w.is_agent.astype(bool)
    braked_ticks, slowest = 0, 1e9
    for k in range(60):
        ow.transform_system(); ow.culling_system(view_proj=vp)
        mn, mx = ow.world_aabbs()
        brake = ow.traffic_front_ray_brakes(mn[:w.n], mx[:w.n], w.group, w.mask, w.is_agent, st["mode"])
        ow.traffic_ai_onrails_braked(ol, w.is_agent, st["lane"], st["s"], st["speed"], st["mode"], st["look"], brake, DT)
        ow.advance_movers(w.mover_kind, st["vel"], w.mover_lo, w.mover_hi, DT)
        if fused:
            t.run(capi.FULL | capi.PRODUCE_NEXT)             # tick, rays, and the next frame's step in the end-of-tick kernel
        else:
            t.run(capi.FULL)                                 # tick + rays ...
        if k % 10 == 9 or k < 2:
            assert np.array_equal(t.visible(), ow.visible()), f"tick {k}"
            got = t.traffic_brakes()
            assert np.array_equal(bits(got[a]), bits(brake[a])), f"tick {k}: {(got[a] != brake[a]).sum()} brakes differ"
        if not fused:
            t.advance_movers(DT)                             # ... then the step that uses them
        braked_ticks += int((brake[a] > 0).sum() > 20)
        slowest = min(slowest, float(st["speed"][a & (st["mode"] == 2) & (brake > 0.5)].min(initial=1e9)))
        if k % 10 == 9:
            assert_agents_equal(t, w, st, ow)
    assert braked_ticks > 30 and slowest < 6.0               # agents really braked, hard enough to fall well below the 12 m/s limit
    ow.transform_system()
    t.set_frame_producer(0)
    t.run(capi.XFORM)
    assert np.array_equal(t.world_matrices(), ow.world_matrices()[:w.n])
    t.close(); ow.close(); ol.close()


def test_tier_selection_hysteresis_and_caps(oracle):
    w = laned_world(16, 16, seed=9)
    ow, ol, st = oracle_side(oracle, w)
    t = WorldTick.from_world(w, broadphase=False)
    a = w.is_agent.astype(bool)
    player = np.float32([500.0, 0.0, 510.0])
    for rnd, (caps, pl) in enumerate([((24, 64), player), ((24, 64), player + np.float32([30, 0, 10])), ((0, 0), player), ((5, 0), player), ((3, 7), player - np.float32([80, 0, 40]))]):
        want, counts = ow.traffic_lod_tiers(w.is_agent, st["mode"], pl, max_physics=caps[0], max_kinematic=caps[1])
        got_counts = t.select_traffic_tiers(pl, max_physics=caps[0], max_kinematic=caps[1])
        st["mode"][a] = want[a]                              # applyMode: the desired tier becomes the vehicle's mode
        md = t.traffic_agents()[3]
        assert np.array_equal(md[a], want[a]), f"round {rnd}"
        assert got_counts == counts and sum(counts) == int(a.sum())
        assert counts[0] > 0 and counts[1] > 0 and counts[2] > 0
        if caps[0]:
            assert counts[0] <= caps[0]
        if caps[1]:
            assert counts[1] <= caps[1]
        # agents promoted out of the OnRails tier stop being moved by the mover; the rest go on
        oracle_advance(ow, ol, w, st, DT)
        t.advance_movers(DT)
        assert_agents_equal(t, w, st, ow)
    t.close(); ow.close(); ol.close()


def test_total_cap_picks_the_farthest_onrails_first_then_kinematic_then_physics(oracle):
    """TrafficLODSystem's total cap (sc_traffic_lod.cpp:419-465) after a tier selection: the surplus over maxTrafficVehiclesTotal
    goes tier by tier -- OnRails, Kinematic, Physics --, the farthest first.  Caps that bite into the second and third bucket,
    a cap above the vehicle count, no cap; the list and its order equal the oracle's."""
    w = laned_world(12, 12, seed=17)
    ow, ol, st = oracle_side(oracle, w)
    t = WorldTick.from_world(w, broadphase=False)
    a = w.is_agent.astype(bool)
    player = np.float32([380.0, 0.0, 400.0])
    want_modes, counts = ow.traffic_lod_tiers(w.is_agent, st["mode"], player, max_physics=0, max_kinematic=0)
    t.select_traffic_tiers(player, max_physics=0, max_kinematic=0)
    st["mode"][a] = want_modes[a]
    total = int(a.sum())
    assert counts[0] > 3 and counts[1] > 10 and counts[2] > 100
    for max_total in (0, total + 5, total, total - 1, total - 50, counts[0] + counts[1] + 7, counts[0] + 4, 3, 1):
        want = ow.traffic_lod_despawns(w.is_agent, st["mode"], player, max_total)
        got = t.select_traffic_despawns(player, max_total)
        assert np.array_equal(got, want), f"max_total {max_total}: {len(got)} vs {len(want)}"
        if max_total and max_total < total:
            assert len(got) == total - max_total
            tiers = st["mode"][got]
            assert (np.diff((2 - tiers.astype(np.int32))) >= 0).all()      # OnRails (2) block, then Kinematic (1), then Physics (0)
    t.close(); ow.close(); ol.close()


def test_per_agent_traffic_sensors_and_what_the_ai_leaves_in_them_40_ticks(oracle):
    """TrafficSensors is a per-entity component (sc_traffic_common.h:46-53): the AI casts each agent's ray with ITS frontRayLength,
    brakes by ITS safeDistance (sc_traffic_ai.cpp:306-308), and writes lastHitDistance / lastHitType back (:339-345).  Here a third of
    the agents see 35 m and keep 18 m, a third 8 m and 3 m, the rest the defaults.  The rays meet vehicles (Vehicle) and
    pedestrians (dynamic bodies that are no vehicles: World).  Brakes, hit distances and hit kinds as bit patterns against the
    oracle on every checked tick."""
    w = laned_world(14, 12, seed=33)
    a = w.is_agent.astype(bool)
    w.scale[a, 1] = np.float32(2.0)
    is_vehicle = (a | (w.mover_kind == 1)).astype(np.uint8)
    ow, ol, st = oracle_side(oracle, w)
    vp = camera_view_proj(w.camera)
    t = WorldTick.from_world(w, broadphase=True, max_pairs=1 << 20)
    t.set_view_proj(vp)
    t.set_traffic_sensors(True)
    idx = np.flatnonzero(a)
    ray = np.full(w.n, 20.0, np.float32); safe = np.full(w.n, 10.0, np.float32)
    ray[idx[0::3]], safe[idx[0::3]] = 35.0, 18.0
    ray[idx[1::3]], safe[idx[1::3]] = 8.0, 3.0
    t.upload_traffic_sensors(0, ray, safe)
    seen_types, differ = set(), 0
    for k in range(40):
        ow.transform_system(); ow.culling_system(view_proj=vp)
        mn, mx = ow.world_aabbs()
        brake, dist, typ = ow.traffic_front_ray_sensors(mn[:w.n], mx[:w.n], w.group, w.mask, w.is_agent, st["mode"], is_vehicle, ray, safe)
        default_brake = ow.traffic_front_ray_brakes(mn[:w.n], mx[:w.n], w.group, w.mask, w.is_agent, st["mode"])
        differ += int((default_brake[a] != brake[a]).sum())
        ow.traffic_ai_onrails_braked(ol, w.is_agent, st["lane"], st["s"], st["speed"], st["mode"], st["look"], brake, DT)
        ow.advance_movers(w.mover_kind, st["vel"], w.mover_lo, w.mover_hi, DT)
        t.run(capi.FULL)
        if k % 8 == 7 or k < 2:
            onr = a & (st["mode"] == 2)
            gd, gt = t.traffic_sensors()
            assert np.array_equal(bits(t.traffic_brakes()[onr]), bits(brake[onr])), f"tick {k}"
            assert np.array_equal(bits(gd[onr]), bits(dist[onr])) and np.array_equal(gt[onr], typ[onr]), f"tick {k}"
            assert (dist[onr][typ[onr] == 0] == ray[onr][typ[onr] == 0]).all()          # no hit: the agent's own ray length
            seen_types |= set(np.unique(typ[onr]).tolist())
        t.advance_movers(DT)
        if k % 8 == 7:
            assert_agents_equal(t, w, st, ow)
    assert {0, 2, 3} <= seen_types and differ > 50             # misses, vehicles and pedestrians all occur, and the per-agent values really change brakes
    t.close(); ow.close(); ol.close()


def _split_laned(w, grid, S):
    """a laned config-5 world generated tile-major -> one SynthWorld per tile (movers, agents and the whole lane graph included)"""
    from sc_gameengine_amd import tiles
    n = w.n // (grid[0] * grid[1])
    parts = []
    for r in range(grid[0] * grid[1]):
        sl = slice(r * n, (r + 1) * n)
        tx, tz = tiles.tile_of(r, grid)
        parent = w.parent[sl].copy()
        parent[parent >= 0] -= r * n
        p = sw.SynthWorld(pos=w.pos[sl].copy(), rot=w.rot[sl].copy(), scale=w.scale[sl].copy(), parent=parent, bmin=w.bmin[sl], bmax=w.bmax[sl],
                          has_mesh=w.has_mesh[sl], has_bounds=w.has_bounds[sl], mesh=w.mesh[sl], material=w.material[sl],
                          group=w.group[sl], mask=w.mask[sl], sector_of=w.sector_of[sl],
                          origin=(w.origin[0] + tx * S[0], w.origin[1] + tz * S[1]), sectors=S, camera=w.camera,
                          mover_kind=w.mover_kind[sl], mover_vel=w.mover_vel[sl].copy(), mover_lo=w.mover_lo[sl], mover_hi=w.mover_hi[sl])
        p.is_agent, p.agent_lane, p.agent_s = w.is_agent[sl], w.agent_lane[sl].copy(), w.agent_s[sl].copy()
        p.agent_speed, p.agent_mode, p.lane_graph = w.agent_speed[sl].copy(), w.agent_mode[sl].copy(), w.lane_graph
        parts.append(p)
    return parts, n


@pytest.mark.parametrize("fused", [True, False])
def test_agents_brake_for_vehicles_on_the_neighbour_tile_60_ticks(oracle, fused):
    """The reference's obstacle ray sees the whole physics world (sc_traffic_ai.cpp:300-345 -> sc_physics.cpp:740-777).  On a tiled
    world the rays are therefore cast in the pair half, behind the border merge, and the border messages carry the neighbours'
    core-edge records (the halo section) into the ring bins: an agent within 20 m of the tile edge brakes for a vehicle on the
    other tile.  2 x 1 tiles on one GPU (device copies as the network), every lane crossing the shared edge busy; brakes of BOTH
    tiles against the WHOLE-world oracle's rays, bit for bit, on every tick; lane state and positions every ten ticks.
    fused: the next frame's step rides on the end-of-tick kernel of the tick half, i.e. it runs BEFORE this tick's rays -- the brake
    then acts one tick later (the reference's own ordering against Bullet's last step); the oracle is stepped the same way.
    Explicit step behind the pair half: no lag."""
    import torch
    from sc_gameengine_amd import tiles
    grid, S = (2, 1), (6, 8)
    w = sw.generate_config5(S[0] * grid[0], S[1] * grid[1], laned=True, tiles=grid)
    a = w.is_agent.astype(bool)
    w.scale[a, 1] = np.float32(2.0)                                         # (the ray runs 0.6 m above the agents' origins: see the one-tile test)
    # crowd the lanes near the shared edge (x = 6 * 64): agents there start rolling at once
    edge_x = 64.0 * S[0]
    near = a & (np.abs(w.pos[:, 0] - edge_x) < 40.0)
    w.agent_speed[near] = np.float32(9.0)
    parts, n = _split_laned(w, grid, S)
    ow, ol, st = oracle_side(oracle, w)
    vp = camera_view_proj(w.camera)
    ticks, bufs = [], []
    for r, p in enumerate(parts):
        t = WorldTick.from_world(p, broadphase=True, max_pairs=1 << 19)
        t.set_view_proj(vp)
        t.set_traffic_sensors(True)                                         # before the border buffers: the messages get their halo section
        if fused:
            t.set_frame_producer(2, DT)
        ticks.append(t)
        bufs.append(tiles.BorderBuffers(t, r, grid, "cuda"))
    assert ticks[0].border_bytes(4) > 64 * 32 * (S[1] + 2)                  # (the halo: a full bin per cell of the side)
    flags = capi.FULL | capi.SPLIT_PAIRS | (capi.PRODUCE_NEXT if fused else 0)
    prev = np.zeros(w.n, np.float32)
    cross, braked = 0, 0
    for k in range(60):
        ow.transform_system()
        mn, mx = ow.world_aabbs()
        now = ow.traffic_front_ray_brakes(mn[:w.n], mx[:w.n], w.group, w.mask, w.is_agent, st["mode"])
        ow.traffic_ai_onrails_braked(ol, w.is_agent, st["lane"], st["s"], st["speed"], st["mode"], st["look"], prev if fused else now, DT)
        ow.advance_movers(w.mover_kind, st["vel"], w.mover_lo, w.mover_hi, DT)
        for t in ticks:
            t.run(flags)
        for t in ticks:
            t.sync()
        for r, b in enumerate(bufs):
            for d, nb in tiles.neighbours(r, grid).items():
                bufs[nb].recv[7 - d].copy_(b.send[d])
        torch.cuda.synchronize()
        for t in ticks:
            t.run_pairs()
        got = np.concatenate([t.traffic_brakes() for t in ticks])
        onr = a & (st["mode"] == 2)
        assert np.array_equal(bits(got[onr]), bits(now[onr])), f"tick {k}: {(got[onr] != now[onr]).sum()} brakes differ from the whole world's"
        # brakes caused by a box of the OTHER tile: recompute with only the agent's own tile's boxes and see which differ
        if k % 15 == 0:
            for r in range(2):
                sl = slice(r * n, (r + 1) * n)
                own = ow.traffic_front_ray_brakes(np.where(np.arange(w.n)[:, None] // n == r, mn[:w.n], np.float32(np.inf)), np.where(np.arange(w.n)[:, None] // n == r, mx[:w.n], np.float32(-np.inf)),
                                                  w.group, w.mask, w.is_agent, st["mode"])
                cross += int((own[sl] != now[sl]).sum())
        braked += int((now[onr] > 0).sum())
        if not fused:
            for t in ticks:
                t.advance_movers(DT)
        prev = now
        if k % 10 == 9:
            for r, t in enumerate(ticks):
                sl = slice(r * n, (r + 1) * n)
                ln, ls, sp, md = t.traffic_agents()
                ar = a[sl]
                assert np.array_equal(ln[ar], st["lane"][sl][ar]) and np.array_equal(bits(ls[ar]), bits(st["s"][sl][ar])), f"tick {k} tile {r}"
                assert np.array_equal(bits(sp[ar]), bits(st["speed"][sl][ar])), f"tick {k} tile {r}"
                assert np.array_equal(bits(t.positions()), bits(ow.local_positions()[:w.n][sl])), f"tick {k} tile {r}"
            assert all(t.counts().border_lost == 0 for t in ticks)
    assert braked > 500 and cross > 0, (braked, cross)                      # agents braked, some of them for a box of the neighbour tile
    for t in ticks:
        t.close()
    ow.close(); ol.close()
