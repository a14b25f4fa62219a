"""Multi-tile broadphase on ONE GPU: several contexts, each owning a tile, exchange their border
messages by device copies (what RCCL send/recv does between GPUs).  The union of the tiles' pair
lists must be the oracle's pair set of the whole world -- no pair lost at an edge, none reported twice."""
import numpy as np
import pytest

from sc_gameengine_amd import capi, synth_world as sw, tiles
from sc_gameengine_amd.tick import WorldTick
from tests import worlds

pytestmark = pytest.mark.gpu


def split_world(w, grid, sectors_per_tile):
    """SynthWorld generated tile-major: rank r owns the contiguous slice r*n..(r+1)*n."""
    n = w.n // (grid[0] * grid[1])
    out = []
    for r in range(grid[0] * grid[1]):
        sl = slice(r * n, (r + 1) * n)
        tx, tz = tiles.tile_of(r, grid)
        parent = w.parent[sl].copy()
        parent[parent >= 0] -= r * n
        out.append(sw.SynthWorld(pos=w.pos[sl].copy(), rot=w.rot[sl], scale=w.scale[sl], parent=parent, bmin=w.bmin[sl], bmax=w.bmax[sl],
                                 has_mesh=w.has_mesh[sl], has_bounds=w.has_bounds[sl], mesh=w.mesh[sl], material=w.material[sl],
                                 group=w.group[sl], mask=w.mask[sl], sector_of=w.sector_of[sl],
                                 origin=(w.origin[0] + tx * sectors_per_tile[0], w.origin[1] + tz * sectors_per_tile[1]),
                                 sectors=sectors_per_tile, camera=w.camera))
    return out, n


@pytest.mark.parametrize("grid", [(2, 1), (2, 2), (4, 2)])
def test_tiled_pairs_equal_whole_world_pairs(oracle, grid):
    import torch
    S = (6, 6)
    w = sw.generate(S[0] * grid[0], S[1] * grid[1], 15, tiles=grid)
    rng = np.random.default_rng(7)
    dyn = rng.random(w.n) < 0.3
    w.group[dyn], w.mask[dyn] = sw.GROUP_DYNAMIC, sw.MASK_ALL
    # push many root props right onto tile edges so boxes straddle them (and corners)
    roots = np.flatnonzero((w.parent < 0) & (np.arange(w.n) % 16 != 0))
    edge = rng.choice(roots, len(roots) // 8, replace=False)
    w.pos[edge, 0] = (np.round(w.pos[edge, 0] / (64.0 * S[0])) * 64.0 * S[0] + rng.uniform(-1.0, 1.0, len(edge))).astype(np.float32)
    edge2 = rng.choice(roots, len(roots) // 8, replace=False)
    w.pos[edge2, 2] = (np.round(w.pos[edge2, 2] / (64.0 * S[1])) * 64.0 * S[1] + rng.uniform(-1.0, 1.0, len(edge2))).astype(np.float32)

    parts, n = split_world(w, grid, S)
    ow = worlds.oracle_world(oracle, w, camera=False)
    ticks = [WorldTick.from_world(p, broadphase=True) for p in parts]
    bufs = [tiles.BorderBuffers(t, r, grid, "cuda") for r, t in enumerate(ticks)]
    flags = capi.XFORM | capi.BROADPHASE | capi.SPLIT_PAIRS
    for step in range(3):
        if step:
            ow.nudge_roots_x(0.9)
            for t in ticks:
                t.nudge_roots_x(0.9)
        ow.transform_system()
        mn, mx = ow.world_aabbs()
        want = oracle.broadphase_grid(mn, mx, w.group, w.mask, 16.0)
        for t in ticks:
            t.run(flags)
        for t in ticks:
            t.sync()
        for r, b in enumerate(bufs):                                  # the "network": send[d] -> neighbour's recv[7-d]
            for d, nb in tiles.neighbours(r, grid).items():
                bufs[nb].recv[7 - d].copy_(b.send[d])
        torch.cuda.synchronize()
        got = []
        for t in ticks:
            t.run_pairs()
        for t in ticks:
            p, total = t.pairs()
            assert total == len(p)
            c = t.counts()
            assert c.big_boxes == 0 and c.bin_overflow == 0
            got.append(tiles.global_pair_ids(p, n))
        got = np.concatenate(got).astype(np.uint64)
        lo, hi = np.minimum(got[:, 0], got[:, 1]), np.maximum(got[:, 0], got[:, 1])
        key = np.sort(lo << np.uint64(32) | hi)
        wkey = want[:, 0].astype(np.uint64) << np.uint64(32) | want[:, 1].astype(np.uint64)
        assert len(key) == len(np.unique(key)), "a pair was reported twice"
        assert np.array_equal(key, wkey), f"{len(key)} pairs vs {len(wkey)} expected"
        assert len(wkey) > 50
        crossing = (want[:, 0] // n) != (want[:, 1] // n)
        assert crossing.sum() > 5                                      # pairs that span two tiles really occur
    for t in ticks:
        t.close()
    ow.close()


def test_context_runs_on_an_external_stream(oracle):
    """scTickSetStream: the tick is stream-ordered with the caller's own work on that stream (what the
    RCCL exchange relies on) -- including the legacy default stream, whose handle is NULL."""
    import torch
    w = sw.generate(8, 8, 15)
    ow = worlds.oracle_world(oracle, w, camera=False)
    ow.transform_system()
    for stream in (torch.cuda.Stream(), None):
        t = WorldTick.from_world(w, broadphase=False)
        t.set_stream(stream.cuda_stream if stream is not None else 0, external=True)
        t.run(capi.XFORM)
        (stream.synchronize() if stream is not None else torch.cuda.synchronize())
        assert np.array_equal(t.world_matrices(), ow.world_matrices())
        t.set_stream(0, external=False)                # back to the context's own stream
        t.mark_dirty(0, w.n); t.run(capi.XFORM)
        assert np.array_equal(t.world_matrices(), ow.world_matrices())
        t.close()
    ow.close()
