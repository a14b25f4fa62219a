"""Ray queries over the broadphase bins against the oracle's brute force over every world AABB.
Own spec (Bullet does the reference's ray tests and is not in the tree): direction handling and result fields after
PhysicsWorld::raycast (sc_physics.cpp:740-777), ray-box arithmetic = the reference's intersectRayAABB
(editor_core.cpp:438-470).  The world AABBs equal the oracle's as IEEE values, so every field must match exactly."""
import numpy as np
import pytest

from sc_gameengine_amd import capi, synth_world as sw
from sc_gameengine_amd.tick import WorldTick
from tests import worlds

pytestmark = pytest.mark.gpu
FLAGS = capi.XFORM | capi.BROADPHASE | capi.DENSE_AABBS | capi.RAYS


def compare(got, want, n_entities=None):
    assert len(got) == len(want)
    for f in ("hit", "id", "layer"):
        assert np.array_equal(got[f], want[f]), f
    for f in ("distance", "position", "normal"):
        a, b = got[f][got["hit"] == 1], want[f][want["hit"] == 1]
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f        # bit patterns
    miss = got["hit"] == 0
    assert (got["id"][miss] == 0xFFFFFFFF).all() and (got["normal"][miss] == np.float32([0, 1, 0])).all()


def random_rays(rng, k, spread, long_share=0.2):
    o = rng.uniform(-spread, spread, (k, 3)).astype(np.float32)
    o[:, 1] = rng.uniform(-3, 8, k)
    d = rng.normal(size=(k, 3)).astype(np.float32)
    d[:, 1] *= 0.15
    d *= rng.uniform(0.01, 50.0, (k, 1)).astype(np.float32)                    # not normalised on purpose
    md = np.where(rng.random(k) < long_share, rng.uniform(200, 900, k), rng.uniform(2, 60, k)).astype(np.float32)
    mask = rng.choice(np.array([1, 2, 3, 0xFFFFFFFF], np.uint32), k)
    return o, d, md, mask


def run_world(oracle, w, rays, ticks=1, nudge=None, graph=False):
    ow = worlds.oracle_world(oracle, w, camera=False)
    t = WorldTick.from_world(w, broadphase=True)
    t.set_ray_queries(*rays)
    if graph:
        t.set_graph_mode(True)
    for k in range(ticks):
        if nudge and k:
            ow.nudge_roots_x(nudge); t.nudge_roots_x(nudge)
        ow.transform_system()
        t.run(FLAGS)
        mn, mx = ow.world_aabbs()
        gmn, gmx = t.world_aabbs()
        assert np.array_equal(gmn, mn) and np.array_equal(gmx, mx)
        want = oracle.raycast_boxes(mn, mx, w.group, w.mask, *rays)
        got = t.ray_hits()
        compare(got, want)
        got_pairs, total = t.pairs()                                            # the pair search still sees full bins afterwards
        assert total == len(oracle.broadphase_bruteforce(mn, mx, w.group, w.mask))
    c = t.counts()
    t.close(); ow.close()
    return want, c


def test_random_world_random_rays(oracle):
    w = worlds.random_world(4000, seed=61, spread=220.0, max_depth=3)
    w.bmin[:30] *= 40.0; w.bmax[:30] *= 40.0                                    # big boxes: they are only in the big list
    rng = np.random.default_rng(62)
    o, d, md, mask = random_rays(rng, 3000, 260.0)
    # special cases: axis-parallel rays (the |dir| < 1e-6 branch), rays starting inside a box, no direction, far outside
    d[:200, 0] = 0.0
    d[200:400, 2] = 0.0
    d[400:500, [0, 2]] = 0.0
    inside = rng.choice(np.flatnonzero(w.has_bounds == 1), 300, replace=False)
    d[500:520] = 0.0
    d[520:530] = np.float32([1e-4, 0, 1e-4])                                    # |dir|^2 <= 1e-6: rejected like the reference
    md[530:540] = -1.0
    o[540:600] += np.float32([3000.0, 0, -2500.0])                              # outside the bin grid: big list only
    want, c = run_world(oracle, w, (o, d, md, mask), ticks=2, nudge=0.8)
    assert want["hit"].sum() > 300 and (want["hit"] == 0).sum() > 100
    assert c.big_boxes >= 30
    # rays from the centre of existing boxes hit at distance 0 with the default normal
    ow = worlds.oracle_world(oracle, w, camera=False); ow.transform_system()
    mn, mx = ow.world_aabbs(); ow.close()
    o2 = ((mn[inside] + mx[inside]) * 0.5).astype(np.float32)
    rays2 = (o2, d[:300].copy() + np.float32([0.3, 0.1, 0.2]), md[:300], np.full(300, 0xFFFFFFFF, np.uint32))
    want2, _ = run_world(oracle, w, rays2)
    started_inside = want2["distance"] == 0.0
    assert started_inside.sum() > 100 and (want2["normal"][started_inside] == np.float32([0, 1, 0])).all()


def test_traffic_front_rays_on_config5(oracle):
    """One ray per vehicle as the traffic AI casts it (sc_traffic_ai.cpp:303-319): 1.7 m ahead, 0.6 m up, 20 m, mask 1."""
    w = sw.generate_config5(8, 8)
    veh = np.flatnonzero(w.mover_kind == 1)
    yaw = w.rot[veh, 1]
    fwd = np.stack([np.sin(yaw), np.zeros_like(yaw), np.cos(yaw)], axis=1).astype(np.float32)
    o = (w.pos[veh] + fwd * np.float32(1.7) + np.float32([0, 0.6, 0])).astype(np.float32)
    rays = (o, fwd, np.full(len(veh), 20.0, np.float32), np.full(len(veh), 1, np.uint32))
    want, _ = run_world(oracle, w, rays, graph=True, ticks=2)
    assert len(veh) == 64 * 12 and 3 < want["hit"].sum() < len(veh)
    assert (want["layer"][want["hit"] == 1] == 1).all()                         # static props (group 2) are masked out


def test_rays_see_neighbour_boxes_that_reach_into_the_tile(oracle):
    """A tile answers for its own sectors: that includes the neighbours' boxes that straddle into them (they arrive
    with the border exchange; the queries run after the merge).  The rest of a ray belongs to the neighbour."""
    import torch
    from sc_gameengine_amd import tiles
    from tests.test_gpu_tiles import split_world
    grid, S = (2, 1), (6, 6)
    w = sw.generate(S[0] * grid[0], S[1], 15, tiles=grid)
    w.group[:], w.mask[:] = sw.GROUP_DYNAMIC, sw.MASK_ALL
    n = w.n // 2
    edge = 64.0 * S[0]
    # four root props of tile 1 moved onto the edge: unit boxes spanning x in [edge - 0.2, edge + 0.8]
    movers = np.flatnonzero((w.parent < 0) & (np.arange(w.n) % 16 != 0) & (np.arange(w.n) >= n))[:4]
    zs = np.float32([40.0, 110.0, 200.0, 300.0])
    w.pos[movers] = np.stack([np.full(4, edge + 0.3, np.float32), np.full(4, 200.0, np.float32), zs], axis=1)
    w.scale[movers] = 1.0; w.rot[movers] = 0.0
    w.bmin[movers], w.bmax[movers] = np.float32([-0.5] * 3), np.float32([0.5] * 3)
    parts, n = split_world(w, grid, S)
    ticks = [WorldTick.from_world(p, broadphase=True, max_pairs=1 << 16) for p in parts]
    bufs = [tiles.BorderBuffers(t, r, grid, "cuda") for r, t in enumerate(ticks)]
    o = np.stack([np.full(4, edge - 4.0, np.float32), np.full(4, 200.0, np.float32), zs], axis=1)    # 200 m up: above every prop
    d = np.float32([[2, 0, 0]] * 4)
    md, mk = np.full(4, 10.0, np.float32), np.full(4, 0xFFFFFFFF, np.uint32)
    ticks[0].set_ray_queries(o, d, md, mk)
    for t in ticks:
        t.run(capi.XFORM | capi.BROADPHASE | capi.SPLIT_PAIRS | (capi.RAYS if t is ticks[0] else 0))
    for t in ticks:
        t.sync()
    cnt = np.zeros(1, np.uint32)
    assert ticks[0].lib.scTickReadRayHits(ticks[0].ctx, None, 0, cnt.ctypes.data_as(capi.U32P)) == 0   # not before scTickRunPairs
    for r, b in enumerate(bufs):
        for dd, nb in tiles.neighbours(r, grid).items():
            bufs[nb].recv[7 - dd].copy_(b.send[dd])
    torch.cuda.synchronize()
    for t in ticks:
        t.run_pairs()
    hits = ticks[0].ray_hits()
    ow = worlds.oracle_world(oracle, w, camera=False); ow.transform_system()
    mn, mx = ow.world_aabbs(); ow.close()
    want = oracle.raycast_boxes(mn, mx, w.group, w.mask, o, d, md, mk)
    assert (hits["hit"] == 1).all() and np.array_equal(want["id"], movers.astype(np.uint32))
    assert ((hits["id"] >> 24) == 1).all()                                       # the boxes hit belong to rank 1
    assert np.array_equal(tiles.global_pair_ids(hits["id"].reshape(-1, 1), n).ravel(), want["id"].astype(np.uint64))
    for f in ("distance", "position", "normal"):
        assert np.array_equal(hits[f].view(np.uint32), want[f].view(np.uint32)), f
    for t in ticks:
        t.close()


def test_ray_api_errors():
    w = worlds.random_world(200, seed=63)
    t = WorldTick.from_world(w, broadphase=True)
    assert t.lib.scTickRun(t.ctx, capi.XFORM | capi.RAYS) == 0
    assert b"SC_TICK_BROADPHASE" in t.lib.scTickGetLastError(t.ctx)
    t.run(capi.XFORM | capi.BROADPHASE)
    n = np.zeros(1, np.uint32)
    assert t.lib.scTickReadRayHits(t.ctx, None, 0, n.ctypes.data_as(capi.U32P)) == 0
    t.run(capi.XFORM | capi.BROADPHASE | capi.RAYS)                               # an empty batch is fine
    assert len(t.ray_hits()) == 0
    assert t.lib.scTickSetRayQueries(t.ctx, 3, None, None, None, None) == 0
    assert t.lib.scTickSetRayQueries(None, 0, None, None, None, None) == 0
    t.close()


def test_occupancy_queries_match_is_occupied_world(oracle):
    """scTickQueryOccupied against the oracle's isOccupiedWorld (sc_traffic_spawner.cpp:93-116) on a config-5 world whose
    vehicles keep moving: points on, near and far from agents, radii around the decision boundary, strict '<'."""
    w = sw.generate_config5(8, 8)
    ow = worlds.oracle_world(oracle, w, camera=False)
    t = WorldTick.from_world(w, broadphase=True)
    rng = np.random.default_rng(71)
    veh = np.flatnonzero(w.mover_kind == 1)
    is_agent = (w.mover_kind == 1).astype(np.uint8)                 # vehicles: group 1; peds share the group, so give them another
    w.group[w.mover_kind == 2] = 4
    t.upload_layers(0, w.group, w.mask)
    vel = w.mover_vel.copy()
    for rnd in range(3):
        ow.advance_movers(w.mover_kind, vel, w.mover_lo, w.mover_hi, 1.0 / 60.0)
        t.advance_movers(1.0 / 60.0)
        pos_now = t.positions()
        k = 200
        pts = rng.uniform(0, 512, (k, 3)).astype(np.float32)
        near = rng.choice(veh, 120, replace=False)
        pts[:120] = pos_now[near] + rng.uniform(-3, 3, (120, 3)).astype(np.float32)
        radius = rng.uniform(0.5, 6.0, k).astype(np.float32)
        # exactly on the boundary: distance == radius must NOT block
        pts[0] = pos_now[near[0]] + np.float32([3.0, 0.0, 4.0]); radius[0] = 5.0
        pts[1] = pos_now[near[1]]; radius[1] = 0.0
        mask = np.full(k, 1, np.uint32)
        got = t.occupied(pts, radius, mask)
        want = np.array([ow.is_occupied(is_agent, pts[j], radius[j]) for j in range(k)], np.uint8)
        assert np.array_equal(got, want)
        assert 20 < want.sum() < k and want[1] == 0
        # the group mask selects the agents: nothing has group 8; peds (group 4) answer for themselves
        assert t.occupied(pts, radius, np.full(k, 8, np.uint32)).sum() == 0
    peds = np.flatnonzero(w.mover_kind == 2)
    p = t.positions()[peds[:5]]
    assert (t.occupied(p, np.full(5, 0.1, np.float32), np.full(5, 4, np.uint32)) == 1).all()
    assert t.lib.scTickQueryOccupied(t.ctx, 300, None, None, None, None) == 0
    t.close(); ow.close()


def test_ray_queries_against_the_reference_intersectRayAABB_cases():
    """The reference's own answers (tests/golden/ray_aabb_ref.npz, group b: sc::editor::intersectRayAABB compiled unmodified,
    oracle/make_golden_rays.py) through scTickSetRayQueries: case i is one box in sector i of a row of 512 sectors and one ray
    near it, with a far limit of 64 m that keeps every ray away from the other cases' boxes (tests/test_oracle_pinned.py shows
    that the limit does not change an answer).  The device normalises the raw direction itself; the golden direction is that
    normalisation in float32.  Hit flags and ids equal, entry distances equal as bit patterns."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ray_aabb_ref.npz"))
    mn, mx = g["b_min"], g["b_max"]
    n = len(mn)
    zeros = np.zeros((n, 3), np.float32)
    w = sw.SynthWorld(pos=zeros.copy(), rot=zeros.copy(), scale=np.ones((n, 3), np.float32), parent=np.full(n, -1, np.int32),
                      bmin=mn.copy(), bmax=mx.copy(), has_mesh=np.ones(n, np.uint8), has_bounds=np.ones(n, np.uint8),
                      mesh=np.zeros(n, np.uint32), material=np.zeros(n, np.uint32),
                      group=np.ones(n, np.uint32), mask=np.full(n, 0xFFFFFFFF, np.uint32),
                      sector_of=np.zeros((n, 2), np.int32), origin=(0, 0), sectors=(n, 1))
    t = WorldTick.from_world(w, broadphase=True)
    t.set_ray_queries(g["b_origin"], g["b_dir_raw"], np.full(n, 64.0, np.float32), np.full(n, 0xFFFFFFFF, np.uint32))
    for flags in (FLAGS, FLAGS):                      # the learn tick, then a tick on remembered (ordered) slots
        t.run(flags)
        gmn, gmx = t.world_aabbs()
        assert np.array_equal(gmn.view(np.uint32), mn.view(np.uint32)) and np.array_equal(gmx.view(np.uint32), mx.view(np.uint32))
        got = t.ray_hits()
        assert np.array_equal(got["hit"], g["b_hit"].astype(np.uint32))
        hit = got["hit"] == 1
        assert np.array_equal(got["id"][hit], np.flatnonzero(hit).astype(np.uint32))
        assert np.array_equal(got["distance"][hit].view(np.uint32), g["b_t"][hit].view(np.uint32))
    assert 300 < hit.sum() < n
    t.close()
