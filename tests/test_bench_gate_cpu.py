"""bench.py's per-rank pair gate at N > 1, checked on the CPU against an independent construction:
the WHOLE tiled world generated at once (tile-major, as the ranks own it), the oracle's pair set of it, each pair handed to
the rank that owns the sector holding the low corner of the intersection.  Every rank's gate -- which rebuilds only its tile
plus the one-sector ring around it -- must accept exactly that list, and reject a list with a pair missing or added."""
import argparse

import numpy as np
import pytest

import bench
from sc_gameengine_amd import synth_world as sw


def park_vehicles_on_sector_edges(w):
    """config 5 only: in every sector vehicle 17 stops 0.3 m short of the sector's +x edge and vehicle 18 0.3 m behind its -x
    edge, both on the sector's centre line -- so they overlap across EVERY sector boundary in x, tile boundaries included.
    Keyed on sector coordinates and the index inside the sector: the same edit on the whole world and on a rank's rebuilt one."""
    if w.mover_kind is None:
        return
    per = 32
    k = np.arange(w.n) % per
    cx, cz = w.sector_of[:, 0].astype(np.float32), w.sector_of[:, 1].astype(np.float32)
    for kk, off in ((17, np.float32(64.0 - 0.3)), (18, np.float32(0.3))):
        sel = k == kk
        w.pos[sel, 0] = cx[sel] * np.float32(64.0) + off
        w.pos[sel, 2] = cz[sel] * np.float32(64.0) + np.float32(32.0)
        w.mover_vel[sel] = 0.0


@pytest.mark.parametrize("workload,grid", [("config5", (2, 2)), ("config5", (4, 2)), ("config3dyn", (2, 1))])
def test_rank_gate_accepts_the_whole_world_pairs_split_by_ownership(oracle, workload, grid):
    S, steps = 8, 7
    args = argparse.Namespace(sectors=S, workload=workload)
    tx, tz = grid
    SX, SZ = (S // 2, S) if workload == "config5" else (S, S)
    if workload == "config5":
        w = sw.generate_config5(SX * tx, SZ * tz, tiles=grid)
        park_vehicles_on_sector_edges(w)
    else:
        w = sw.generate(SX * tx, SZ * tz, bench.PROPS, hierarchy=True, tiles=grid)
        dyn = (np.arange(w.n) % 16) == 4
        w.group[dyn], w.mask[dyn] = sw.GROUP_DYNAMIC, sw.MASK_ALL
    n_tile = w.n // (tx * tz)
    ow = oracle.OracleWorld.from_arrays(w.pos, w.rot, w.scale, w.parent, w.bmin, w.bmax, has_mesh=w.has_mesh, has_bounds=w.has_bounds)
    vel = None if w.mover_kind is None else w.mover_vel.copy()
    for _ in range(steps):
        if vel is None:
            ow.nudge_roots_x(0.01)
        else:
            ow.advance_movers(w.mover_kind, vel, w.mover_lo, w.mover_hi, 1.0 / 60.0)
    ow.transform_system()
    mn, mx = ow.world_aabbs()
    want = oracle.broadphase_grid(mn[:w.n], mx[:w.n], w.group, w.mask, 64.0)
    ow.close()
    assert len(want) > (20 if workload == "config5" else 0)
    inv = np.float32(1.0) / np.float32(64.0)
    lx = np.maximum(mn[want[:, 0], 0], mn[want[:, 1], 0]); lz = np.maximum(mn[want[:, 0], 2], mn[want[:, 1], 2])
    sx = np.clip(np.floor(lx * inv).astype(np.int64), 0, tx * SX - 1); sz = np.clip(np.floor(lz * inv).astype(np.int64), 0, tz * SZ - 1)
    owner = (sz // SZ) * tx + (sx // SX)
    gid = ((np.arange(w.n, dtype=np.int64) // n_tile) << 24) | (np.arange(w.n, dtype=np.int64) % n_tile)
    crossing_total = 0
    for rank in range(tx * tz):
        sel = owner == rank
        a, b = gid[want[sel, 0]], gid[want[sel, 1]]
        got = np.stack([np.minimum(a, b), np.maximum(a, b)], axis=1).astype(np.uint32)
        res = bench.tile_pair_gate(args, rank, grid, steps, got, len(got), mutate=park_vehicles_on_sector_edges)
        assert res["pairs_equal"] and res["pairs"] == len(got), f"rank {rank}: {res}"
        crossing_total += res["pairs_with_a_neighbours_box"]
        if len(got):
            assert not bench.tile_pair_gate(args, rank, grid, steps, got[1:], len(got) - 1, mutate=park_vehicles_on_sector_edges)["pairs_equal"]      # a pair missing
        extra = np.concatenate([got, np.uint32([[0, 0xFFFFFF]])]) if len(got) else np.uint32([[0, 0xFFFFFF]])
        assert not bench.tile_pair_gate(args, rank, grid, steps, extra, len(extra), mutate=park_vehicles_on_sector_edges)["pairs_equal"]               # a pair too many
    if workload == "config5":
        assert crossing_total > 10         # pairs that involve a neighbour tile's box really occur and were matched
