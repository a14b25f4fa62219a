"""CPU checks of the step before the path: the product's host-side lane-graph builder against the oracle's
restatement of TrafficLaneGraph, lane walking (advanceAlongLane) across sectors / onto dead ends / over inactive
segments, and the on-rails advance + tier selection of the oracle on a laned config-5 world.  PARITY UNPINNED: the
reference has no test or fixture for its traffic code (and it needs Vulkan headers to compile)."""
import numpy as np

from sc_gameengine_amd import lanes, synth_world as sw
from tests import worlds


def test_lane_builder_equals_oracle(oracle):
    w = sw.generate_config5(12, 9, origin=(-3, 5), laned=True)
    g = w.lane_graph
    ol = oracle.OracleLanes()
    segs = ol.build_sectors(w.sector_of[::32, 0], w.sector_of[::32, 1])
    e = ol.export()
    assert np.array_equal(segs, g.sector_segments)
    for k in ("seg_start", "seg_dir", "seg_length", "seg_end_node", "seg_speed_limit", "node_pos", "node_conn_offset", "node_conn", "seg_active"):
        a, b = np.ascontiguousarray(getattr(g, k)), np.ascontiguousarray(getattr(e, k))
        assert a.shape == b.shape and a.tobytes() == b.tobytes(), k
    # neighbouring sectors share their edge nodes: 4 segments per sector, but fewer than 8 nodes
    assert g.segments == 4 * 12 * 9 and g.nodes < 8 * 12 * 9
    assert (g.seg_length == np.float32(64.0)).all()
    ol.close()


def test_advance_along_lane(oracle):
    ol = oracle.OracleLanes()
    cx, cz = np.meshgrid(np.arange(3), np.arange(2))
    segs = ol.build_sectors(cx.ravel(), cz.ravel())                   # 3 x 2 sectors, row-major
    plus_x = segs[:, 0]                                               # +x lanes of sectors (0,0) (1,0) (2,0) (0,1) ...
    ok, lane, s, pos, dr = ol.advance(plus_x[0], 60.0, 10.0)          # crosses into the next sector's +x lane
    assert ok and lane == plus_x[1] and s == np.float32(6.0) and pos[0] == np.float32(70.0) and tuple(dr) == (1.0, 0.0, 0.0)
    ok, lane, s, pos, _ = ol.advance(plus_x[0], 60.0, 140.0)          # two crossings, then the world's edge: parks on the end node
    assert ok and lane == plus_x[2] and s == np.float32(64.0) and pos[0] == np.float32(192.0)
    ok, lane, s, pos, _ = ol.advance(plus_x[2], 64.0, 5.0)            # already parked: stays
    assert ok and lane == plus_x[2] and s == np.float32(64.0) and pos[0] == np.float32(192.0)
    ol.set_active(plus_x[1], False)
    ok, lane, s, pos, _ = ol.advance(plus_x[0], 60.0, 10.0)           # the next lane is inactive: no connection, parks at the node
    assert ok and lane == plus_x[0] and s == np.float32(64.0)
    assert not ol.advance(plus_x[1], 1.0, 1.0)[0]                     # starting ON an inactive lane fails
    assert not ol.advance(lanes.INVALID_LANE, 0.0, 1.0)[0]
    ol.close()


def test_oracle_on_rails_and_tiers(oracle):
    w = sw.generate_config5(8, 8, laned=True)
    ow = worlds.oracle_world(oracle, w, camera=False)
    ol = oracle.OracleLanes()
    ol.build_sectors(w.sector_of[::32, 0], w.sector_of[::32, 1])
    lane, s, speed, mode = w.agent_lane.copy(), w.agent_s.copy(), w.agent_speed.copy(), w.agent_mode.copy()
    look = np.full(w.n, 12.0, np.float32)
    a = w.is_agent.astype(bool)
    p0 = ow.local_positions().copy()
    for _ in range(120):
        ow.traffic_ai_onrails(ol, w.is_agent, lane, s, speed, mode, look, 1.0 / 60.0)
    p1 = ow.local_positions()
    assert np.array_equal(p0[~a], p1[~a]) and ow.dirty()[a].all()
    assert np.allclose(speed[a].max(), 12.0, atol=0.2)                # smoothExp converges on the speed limit
    d = np.abs(p1[a] - p0[a])
    assert (d[:, 1] == 0).all() and ((d[:, 0] == 0) | (d[:, 2] == 0)).all()     # lanes are axis-parallel, y is kept
    assert d.max() > 15.0
    # pos == lane start + dir * s for every agent that moved
    g = w.lane_graph
    want = (g.seg_start[lane[a]] + g.seg_dir[lane[a]] * s[a][:, None]).astype(np.float32)
    moved = d.max(axis=1) > 0
    assert np.array_equal(want[moved][:, [0, 2]], p1[a][moved][:, [0, 2]])
    # tiers: hysteresis keeps a Kinematic vehicle Kinematic between tierBEnter and tierBExit, an OnRails one OnRails
    player = p1[a][0]
    des, counts = ow.traffic_lod_tiers(w.is_agent, mode, player, max_physics=0, max_kinematic=0)
    dist = np.sqrt(((p1[:, [0, 2]] - player[[0, 2]]) ** 2).sum(axis=1))
    band = a & (dist > 115.0) & (dist < 145.0)
    assert band.sum() > 3 and (des[band] == 2).all()
    des2, _ = ow.traffic_lod_tiers(w.is_agent, np.where(band, 1, mode).astype(np.uint8), player, max_physics=0, max_kinematic=0)
    assert (des2[band] == 1).all()
    assert (des[a & (dist < 49.0)] == 0).all() and sum(counts) == a.sum()
    _, capped = ow.traffic_lod_tiers(w.is_agent, mode, player, max_physics=2, max_kinematic=3)
    assert capped[0] == 2 and capped[1] == 3 and sum(capped) == a.sum()
    ow.close(); ol.close()


def test_total_cap_order_against_a_plain_python_restatement(oracle):
    """orc_traffic_lod_despawns (sc_traffic_lod.cpp:419-465) against the same rule written with Python's stable sort."""
    import numpy as np
    from sc_gameengine_amd import synth_world as sw
    from tests import worlds
    w = sw.generate_config5(6, 6, laned=True)
    rng = np.random.default_rng(4)
    mode = w.agent_mode.copy()
    agents = np.flatnonzero(w.is_agent)
    mode[rng.choice(agents, 40, replace=False)] = 1
    mode[rng.choice(agents, 15, replace=False)] = 0
    ow = worlds.oracle_world(oracle, w, camera=False)
    player = np.float32([150.0, 0.0, 170.0])
    d = np.sqrt(((w.pos[agents, 0] - player[0]) ** 2 + (w.pos[agents, 2] - player[2]) ** 2).astype(np.float32)).astype(np.float32)
    for max_total in (0, len(agents), len(agents) - 10, 60, 20, 2):
        got = ow.traffic_lod_despawns(w.is_agent, mode, player, max_total)
        want = []
        if max_total and len(agents) > max_total:
            left = len(agents) - max_total
            for tier in (2, 1, 0):
                bucket = [k for k in range(len(agents)) if mode[agents[k]] == tier]
                bucket.sort(key=lambda k: -float(d[k]))                      # stable: equal distances keep pool order
                take = bucket[:left]
                want += [int(agents[k]) for k in take]
                left -= len(take)
        assert list(got) == want, f"max_total {max_total}"
    ow.close()
