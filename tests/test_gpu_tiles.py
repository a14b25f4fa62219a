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
            # (no big boxes in this world; the sector overflow list may see a few records: props crowd onto the shared edge from
            #  both tiles, and with remembered slots a box that moved on leaves a null record in its old slot until the next learn tick)
            assert c.big_boxes == 0 and c.bin_overflow < 16 and c.border_lost == 0
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


def run_tiles(oracle, w, grid, S, steps=2, nudge=0.9, border_capacity=None, max_pairs=1 << 16):
    """Whole-world pair set from the oracle vs the union of the tiles' pair lists; returns the tiles' counts."""
    import torch
    parts, n = split_world(w, grid, S)
    ow = worlds.oracle_world(oracle, w, camera=False)
    ticks = [WorldTick.from_world(p, broadphase=True, max_pairs=max_pairs) for p in parts]
    if border_capacity:
        for t in ticks:
            t.set_border_capacity(border_capacity)
    bufs = [tiles.BorderBuffers(t, r, grid, "cuda") for r, t in enumerate(ticks)]
    flags = capi.XFORM | capi.BROADPHASE | capi.SPLIT_PAIRS
    counts = []
    for step in range(steps):
        if step:
            ow.nudge_roots_x(nudge)
            for t in ticks:
                t.nudge_roots_x(nudge)
        ow.transform_system()
        mn, mx = ow.world_aabbs()
        want = oracle.broadphase_grid(mn, mx, w.group, w.mask, 16.0)
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
        got, counts = [], []
        for t in ticks:
            p, total = t.pairs()
            assert total == len(p)
            counts.append(t.counts())
            got.append(tiles.global_pair_ids(p, n))
        got = np.concatenate(got).astype(np.uint64)
        lo, hi = np.minimum(got[:, 0], got[:, 1]), np.maximum(got[:, 0], got[:, 1])
        key = np.sort(lo << np.uint64(32) | hi)
        wkey = want[:, 0].astype(np.uint64) << np.uint64(32) | want[:, 1].astype(np.uint64)
        assert len(key) == len(np.unique(key)), "a pair was reported twice"
        missing, extra = np.setdiff1d(wkey, key), np.setdiff1d(key, wkey)
        assert len(missing) == 0 and len(extra) == 0, (f"step {step}: {len(missing)} pairs missing, {len(extra)} unexpected of {len(wkey)}; "
                                                       f"border_lost per tile {[int(c.border_lost) for c in counts]}")
    for t in ticks:
        t.close()
    ow.close()
    return counts, want, n


@pytest.mark.parametrize("grid", [(2, 2), (4, 2)])
def test_big_boxes_cross_tile_borders(oracle, grid):
    """Boxes the bins cannot hold -- wider than 2x2 sectors, outside the world, or pushed out of a full bin -- travel
    in the border messages' big-box section; every pair with them is found once, whichever tiles the two boxes live in."""
    S = (6, 6)
    w = sw.generate(S[0] * grid[0], S[1] * grid[1], 15, tiles=grid)
    rng = np.random.default_rng(17)
    dyn = rng.random(w.n) < 0.3
    w.group[dyn], w.mask[dyn] = sw.GROUP_DYNAMIC, sw.MASK_ALL
    roots = np.flatnonzero((w.parent < 0) & (np.arange(w.n) % 16 != 0))
    per_tile = w.n // (grid[0] * grid[1])
    tile_x, tile_z = (roots // per_tile) % grid[0], (roots // per_tile) // grid[0]      # an entity lives with its own tile
    TW, TH = 64.0 * S[0], 64.0 * S[1]
    W = TW * grid[0]
    # wide slabs (130-190 m) sitting on the edges and corners of their own tile, colliding with everything
    sel = rng.choice(len(roots), 40, replace=False)
    wide = roots[sel]
    w.pos[wide, 0] = ((tile_x[sel] + rng.integers(0, 2, 40)) * TW + rng.uniform(-40, 40, 40)).astype(np.float32)
    w.pos[wide, 2] = ((tile_z[sel] + rng.integers(0, 2, 40)) * TH + rng.uniform(-40, 40, 40)).astype(np.float32)
    w.scale[wide] = np.float32([1.0, 1.0, 1.0])
    w.rot[wide] = 0.0
    half = rng.uniform(65, 95, (40, 3)).astype(np.float32); half[:, 1] = 2.0
    w.bmin[wide], w.bmax[wide] = -half, half
    w.group[wide], w.mask[wide] = sw.GROUP_DYNAMIC, sw.MASK_ALL
    # a crowd in the sector at the corner where tiles (0,0) (1,0) (0,1) (1,1) meet, drawn from those four tiles:
    # the owner's bin overflows on arrival of the neighbours' copies
    near = np.flatnonzero((tile_x <= 1) & (tile_z <= 1) & ~np.isin(roots, wide))
    crowd = roots[rng.choice(near, 44, replace=False)]        # about half of them drag a child and a grandchild along: ~90 boxes
    w.pos[crowd, 0] = (TW - rng.uniform(0.2, 20.0, 44)).astype(np.float32)
    w.pos[crowd, 2] = (TH - rng.uniform(0.2, 20.0, 44)).astype(np.float32)
    w.group[crowd], w.mask[crowd] = sw.GROUP_DYNAMIC, sw.MASK_ALL
    # and a few boxes beyond the world's edge, next to their own (outer) tile
    outer = np.flatnonzero(((tile_x == 0) | (tile_x == grid[0] - 1)) & ~np.isin(roots, wide) & ~np.isin(roots, crowd))
    pick = rng.choice(outer, 12, replace=False)
    out = roots[pick]
    w.pos[out, 0] = (np.where(tile_x[pick] == 0, -150.0, W + 150.0) + rng.uniform(-30, 30, 12)).astype(np.float32)
    w.group[out], w.mask[out] = sw.GROUP_DYNAMIC, sw.MASK_ALL

    counts, want, n = run_tiles(oracle, w, grid, S)
    assert sum(c.big_boxes for c in counts) >= 40
    assert all(c.border_lost == 0 for c in counts)
    involved = np.isin(want[:, 0], wide) | np.isin(want[:, 1], wide)
    crossing = (want[:, 0] // n) != (want[:, 1] // n)
    assert (involved & crossing).sum() > 20                     # big boxes really pair across tiles


@pytest.mark.parametrize("grid,K", [((2, 1), 95), ((2, 2), 199)])
def test_crowded_sectors_on_tile_edges_lose_nothing(oracle, grid, K):
    """The engine's own density (200 entities per sector, src/sandbox/src/main.cpp:92-99) on tile edges: every bin overflows,
    ring sectors hold far more than a bin.  Since round 3 a ring sector's overflow records cross the border with its bin
    (round 2 counted them in border_lost and the pairs were missing): with the messages sized for the world
    (scTickSetBorderCapacity) the tiled pair set equals the whole world's and nothing is reported lost."""
    S = (3, 3)
    w = sw.generate(S[0] * grid[0], S[1] * grid[1], K, tiles=grid)
    rng = np.random.default_rng(31 + K)
    dyn = rng.random(w.n) < 0.3
    w.group[dyn], w.mask[dyn] = sw.GROUP_DYNAMIC, sw.MASK_ALL
    roots = np.flatnonzero((w.parent < 0) & (np.arange(w.n) % (K + 1) != 0))
    TW, TH = 64.0 * S[0], 64.0 * S[1]
    e = rng.choice(roots, len(roots) // 3, replace=False)
    w.pos[e, 0] = (np.round(w.pos[e, 0] / TW) * TW + rng.uniform(-1.5, 1.5, len(e))).astype(np.float32)
    e = rng.choice(roots, len(roots) // 3, replace=False)
    w.pos[e, 2] = (np.round(w.pos[e, 2] / TH) * TH + rng.uniform(-1.5, 1.5, len(e))).astype(np.float32)
    counts, want, n = run_tiles(oracle, w, grid, S, steps=3, nudge=0.4, border_capacity=384, max_pairs=1 << 21)
    assert all(c.border_lost == 0 and c.pairs_truncated == 0 for c in counts)
    assert sum(c.bin_overflow for c in counts) > 500                 # the overflow lists really carry records
    crossing = (want[:, 0] // n) != (want[:, 1] // n)
    assert crossing.sum() > 200


@pytest.mark.parametrize("capacity", [1, 5, 16])
def test_a_long_border_with_fewer_than_four_fixed_slots_per_cell(oracle, capacity):
    """Round 4's message layout gives every ring cell up to four FIXED record slots and packs the rest into a shared spill-over area.
    On a 600-sector border with a capacity of ONE record per cell on average a cell owns one fixed slot (borderFixedSlots: 1 088
    records for 602 cells) and every cell that holds more goes through the spill-over path (486 records); with 5 or 16 the cells own
    four and the spill-over area is rarely needed: same pair set, nothing lost."""
    S, grid = (600, 2), (1, 2)
    w = sw.generate(S[0] * grid[0], S[1] * grid[1], 4, tiles=grid, ground=False)      # (ground slabs alone would put two records into every ring cell)
    rng = np.random.default_rng(41)
    dyn = rng.random(w.n) < 0.5
    w.group[dyn], w.mask[dyn] = sw.GROUP_DYNAMIC, sw.MASK_ALL
    roots = np.flatnonzero(w.parent < 0)
    TH = 64.0 * S[1]
    e = rng.choice(roots, len(roots) // 4, replace=False)          # boxes onto the 600-sector edge: most cells hold 0-2 records, some more
    w.pos[e, 2] = (np.round(w.pos[e, 2] / TH) * TH + rng.uniform(-1.0, 1.0, len(e))).astype(np.float32)
    counts, want, n = run_tiles(oracle, w, grid, S, steps=3, nudge=0.4, border_capacity=capacity, max_pairs=1 << 20)
    assert all(c.pairs_truncated == 0 for c in counts)
    assert all(c.border_lost == 0 for c in counts)
    crossing = (want[:, 0] // n) != (want[:, 1] // n)
    assert crossing.sum() > 5, crossing.sum()


def test_border_capacity_is_validated():
    w = sw.generate(4, 4, 15)
    t = WorldTick.from_world(w, broadphase=True)
    for bad in (0, 1089):
        with pytest.raises(capi.ScTickError, match="border capacity"):
            t.set_border_capacity(bad)
    before = t.border_bytes(3)
    t.set_border_capacity(256)
    assert t.border_bytes(3) > before
    t.close()


def test_big_box_reaching_past_the_neighbours_is_counted(oracle):
    import torch
    grid, S = (4, 2), (6, 6)
    w = sw.generate(S[0] * grid[0], S[1] * grid[1], 15, tiles=grid)
    w.bmin[1] = np.float32([-2000.0, -1.0, -10.0]); w.bmax[1] = np.float32([2000.0, 1.0, 10.0])      # spans all four tile columns
    w.scale[1] = 1.0; w.rot[1] = 0.0
    parts, n = split_world(w, grid, S)
    ticks = [WorldTick.from_world(p, broadphase=True) for p in parts]
    bufs = [tiles.BorderBuffers(t, r, grid, "cuda") for r, t in enumerate(ticks)]
    for t in ticks:
        t.run(capi.XFORM | capi.BROADPHASE | capi.SPLIT_PAIRS)
    for t in ticks:
        t.sync()
    for r, b in enumerate(bufs):
        for d, nb in tiles.neighbours(r, grid).items():
            bufs[nb].recv[7 - d].copy_(b.send[d])
    torch.cuda.synchronize()
    for t in ticks:
        t.run_pairs()
    lost = [t.counts().border_lost for t in ticks]
    assert lost[0] >= 1 and sum(lost[1:]) == 0
    for t in ticks:
        t.close()


@pytest.mark.parametrize("grid", [(2, 2), (4, 2)])
def test_pipelined_tiles_overlap_pair_search_with_the_next_tick(oracle, grid):
    """scTickSetPairsStream: tick t's merge + pair search run on a second stream while tick t+1's fused kernel runs; bins,
    big list, spill list and border messages are double-buffered by parity.  Several ticks are issued back to back
    without a host synchronisation, then the last tick's pair set must be the whole world's -- a race between the two
    halves would leave stale or missing records behind."""
    import torch
    S = (6, 6)
    w = sw.generate(S[0] * grid[0], S[1] * grid[1], 15, tiles=grid)
    rng = np.random.default_rng(27)
    dyn = rng.random(w.n) < 0.35
    w.group[dyn], w.mask[dyn] = sw.GROUP_DYNAMIC, sw.MASK_ALL
    roots = np.flatnonzero((w.parent < 0) & (np.arange(w.n) % 16 != 0))
    edge = rng.choice(roots, len(roots) // 6, replace=False)
    w.pos[edge, 0] = (np.round(w.pos[edge, 0] / (64.0 * S[0])) * 64.0 * S[0] + rng.uniform(-1.0, 1.0, len(edge))).astype(np.float32)
    big = rng.choice(np.setdiff1d(roots, edge), 12, replace=False)
    w.bmin[big] *= 150.0; w.bmax[big] *= 150.0                       # ~150 m boxes: big list, travelling in the messages
    w.group[big], w.mask[big] = sw.GROUP_DYNAMIC, sw.MASK_ALL
    parts, n = split_world(w, grid, S)
    ow = worlds.oracle_world(oracle, w, camera=False)
    ticks, bufs, s1, s2 = [], [], [], []
    for r, p in enumerate(parts):
        t = WorldTick.from_world(p, broadphase=True, max_pairs=1 << 16)
        a, b = torch.cuda.Stream(), torch.cuda.Stream()
        t.set_stream(a.cuda_stream, external=True)
        t.set_pairs_stream(b.cuda_stream)
        t.set_frame_producer(1, 0.7)
        ticks.append(t); s1.append(a); s2.append(b)
        bufs.append(tiles.BorderBuffers(t, r, grid, "cuda", pipelined=True))
    flags = capi.XFORM | capi.BROADPHASE | capi.SPLIT_PAIRS | capi.PRODUCE_NEXT
    for t in ticks:
        t.nudge_roots_x(0.7)
    steps = 6
    for step in range(steps):
        q = step % len(bufs[0].sets)
        for t in ticks:
            t.run(flags)                                               # tick stream: fused kernel, end-of-tick kernel, pack
        for r, b in enumerate(bufs):                                   # the "network", on the receivers' pairs streams
            for d, nb in tiles.neighbours(r, grid).items():
                s2[nb].wait_stream(s1[r])
                with torch.cuda.stream(s2[nb]):
                    bufs[nb].sets[q][3][7 - d].copy_(b.sets[q][2][d], non_blocking=True)
        for t in ticks:
            t.run_pairs()                                              # pairs stream; nothing waits on the host
    for _ in range(steps):
        ow.nudge_roots_x(0.7)
    ow.transform_system()
    mn, mx = ow.world_aabbs()
    want = oracle.broadphase_grid(mn, mx, w.group, w.mask, 16.0)
    got = []
    for t in ticks:
        p, total = t.pairs()
        c = t.counts()
        assert total == len(p) and c.border_lost == 0 and c.pairs_truncated == 0
        got.append(tiles.global_pair_ids(p, n))
    got = np.concatenate(got).astype(np.uint64)
    lo, hi = np.minimum(got[:, 0], got[:, 1]), np.maximum(got[:, 0], got[:, 1])
    key = np.sort(lo << np.uint64(32) | hi)
    wkey = want[:, 0].astype(np.uint64) << np.uint64(32) | want[:, 1].astype(np.uint64)
    assert len(key) == len(np.unique(key))
    assert np.array_equal(key, wkey), f"{len(np.setdiff1d(wkey, key))} missing, {len(np.setdiff1d(key, wkey))} unexpected of {len(wkey)}"
    assert sum(t.counts().big_boxes for t in ticks) >= 12 and len(wkey) > 200
    torch.cuda.synchronize()
    for t in ticks:
        t.close()
    ow.close()


@pytest.mark.parametrize("declare", [True, False])
def test_pipelined_tiles_leave_unwritten_what_nothing_in_the_world_can_meet(oracle, declare):
    """scTickSetWorldLayers: with the world's layer vocabulary declared, a PIPELINED tile leaves the bins unwritten whose own
    records can meet nothing the world contains (here: static props 2/1 in a world whose other bodies are 4/8 and 8/4, which
    only meet each other) -- nobody will ever read them, so nothing has to be rebuilt either.  Bodies of the two meeting kinds
    drift through prop-only sectors and across tile borders, big boxes of one kind travel in the messages: the pair set is the
    whole world's, with the declaration and without it (then every record is written)."""
    import torch
    grid, S = (2, 2), (6, 6)
    w = sw.generate(S[0] * grid[0], S[1] * grid[1], 15, tiles=grid)
    rng = np.random.default_rng(31)
    roots = np.flatnonzero((w.parent < 0) & (np.arange(w.n) % 16 != 0))
    kind = rng.choice(roots, len(roots) // 5, replace=False)
    a, b = kind[: len(kind) // 2], kind[len(kind) // 2:]
    w.group[a], w.mask[a] = 4, 8
    w.group[b], w.mask[b] = 8, 4
    # the meeting kinds in clusters, so that there are pairs: every b next to some a
    b = b[: len(a)]
    w.pos[b] = w.pos[a[: len(b)]] + rng.uniform(-1.5, 1.5, (len(b), 3)).astype(np.float32) * np.float32([1, 0.2, 1])
    edge = rng.choice(a, len(a) // 4, replace=False)
    w.pos[edge, 0] = (np.round(w.pos[edge, 0] / (64.0 * S[0])) * 64.0 * S[0] + rng.uniform(-1.0, 1.0, len(edge))).astype(np.float32)
    big = rng.choice(a, 6, replace=False)
    w.bmin[big] *= 150.0; w.bmax[big] *= 150.0
    parts, n = split_world(w, grid, S)
    ow = worlds.oracle_world(oracle, w, camera=False)
    ticks, bufs, s1, s2 = [], [], [], []
    for r, p in enumerate(parts):
        t = WorldTick.from_world(p, broadphase=True, max_pairs=1 << 16)
        x, y = torch.cuda.Stream(), torch.cuda.Stream()
        t.set_stream(x.cuda_stream, external=True)
        t.set_pairs_stream(y.cuda_stream)
        if declare:
            t.set_world_layers(w.group, w.mask)
        t.set_frame_producer(1, 1.3)
        ticks.append(t); s1.append(x); s2.append(y)
        bufs.append(tiles.BorderBuffers(t, r, grid, "cuda", pipelined=True))
    flags = capi.XFORM | capi.BROADPHASE | capi.SPLIT_PAIRS | capi.PRODUCE_NEXT
    for t in ticks:
        t.nudge_roots_x(1.3)
    steps = 9
    for step in range(steps):
        q = step % len(bufs[0].sets)
        for t in ticks:
            t.run(flags)
        for r, bb in enumerate(bufs):
            for d, nb in tiles.neighbours(r, grid).items():
                s2[nb].wait_stream(s1[r])
                with torch.cuda.stream(s2[nb]):
                    bufs[nb].sets[q][3][7 - d].copy_(bb.sets[q][2][d], non_blocking=True)
        for t in ticks:
            t.run_pairs()
    for _ in range(steps):
        ow.nudge_roots_x(1.3)
    ow.transform_system()
    mn, mx = ow.world_aabbs()
    want = oracle.broadphase_grid(mn, mx, w.group, w.mask, 16.0)
    got = []
    for t in ticks:
        p, total = t.pairs()
        c = t.counts()
        assert total == len(p) and c.border_lost == 0 and c.pairs_truncated == 0
        got.append(tiles.global_pair_ids(p, n))
        bs = t.bin_stats()
        assert bs["lazy_last_tick"] == declare
        if declare:
            assert 0 < bs["written_every_tick"] < bs["remembered_slots"] * 3 // 4      # prop-only bins stay unwritten
    got = np.concatenate(got).astype(np.uint64)
    lo, hi = np.minimum(got[:, 0], got[:, 1]), np.maximum(got[:, 0], got[:, 1])
    key = np.sort(lo << np.uint64(32) | hi)
    wkey = want[:, 0].astype(np.uint64) << np.uint64(32) | want[:, 1].astype(np.uint64)
    assert len(key) == len(np.unique(key))
    assert np.array_equal(key, wkey), f"{len(np.setdiff1d(wkey, key))} missing, {len(np.setdiff1d(key, wkey))} unexpected of {len(wkey)}"
    assert len(wkey) > 100
    torch.cuda.synchronize()


@pytest.mark.parametrize("grid", [(2, 2), (4, 2)])
def test_global_visible_list_is_the_concatenation_of_the_tiles_lists(oracle, grid):
    """SURVEY 8e, result assembly (north_star: "bit-exact visibility lists at 1, 2, 4 and 8 GPUs"): with the frustum shared, the
    tiles' visible lists, each shifted by rank * entities-per-tile and placed at the offset its rank gets from the all-gathered
    counts, are the WHOLE world's visible list in the reference's order -- the oracle's CullingSystem over the whole pool
    (sc_world_partition.cpp:1273-1280); tile-major creation is what makes the concatenation the serial compaction's order."""
    from sc_gameengine_amd.tick import camera_view_proj
    S = (6, 6)
    w = sw.generate(S[0] * grid[0], S[1] * grid[1], 15, tiles=grid)
    w.has_mesh[::7] = 0                                             # some entities are no candidates; some have no bounds (always visible)
    w.has_bounds[5::11] = 0
    cam = sw.default_camera(float(S[0] * grid[0]) * 64.0)
    cam["pos"][2] = np.float32(float(S[1] * grid[1]) * 64.0 / 2)
    w.camera = cam
    vp = camera_view_proj(cam)
    parts, n = split_world(w, grid, S)
    ow = worlds.oracle_world(oracle, w, camera=False)
    ticks = [WorldTick.from_world(p, broadphase=False) for p in parts]
    for t in ticks:
        t.set_view_proj(vp)
    for step in range(3):
        if step:
            ow.nudge_roots_x(1.7)
            for t in ticks:
                t.nudge_roots_x(1.7)
        ow.transform_system()
        ow.culling_system(view_proj=vp)
        want = ow.visible().astype(np.uint64)
        for t in ticks:
            t.run(capi.XFORM | capi.CULL)
        lists = [t.visible() for t in ticks]
        counts = [len(v) for v in lists]
        whole = np.full(sum(counts), np.uint64(0xFFFFFFFFFFFFFFFF))
        for r, t in enumerate(ticks):
            c1, off1, tot1 = t.gather_visible_counts()             # no communicator: the tile reports itself as the world
            assert (c1.tolist(), off1, tot1) == ([counts[r]], 0, counts[r])
            off, total = tiles.visible_offsets(counts, r)
            assert total == len(want)
            whole[off:off + counts[r]] = lists[r].astype(np.uint64) + np.uint64(r * n)
        assert np.array_equal(whole, want)
        assert min(counts) >= 0 and sum(1 for c in counts if c) >= 2 and 50 < len(want) < w.n      # several tiles really contribute
    for t in ticks:
        t.close()
    ow.close()


def test_world_layer_vocabulary_is_a_checked_contract(oracle):
    """scTickSetWorldLayers is what lets a pipelined tile leave bins unwritten for good, so a collider outside the declared words
    would miss pairs silently.  It is checked instead: (1) uploading or appending layers outside the vocabulary fails, (2) a
    vocabulary that does not cover what the tile already holds fails, (3) records and big boxes that ARRIVE from a neighbour
    outside the vocabulary are counted in ScTickCounts::vocabulary_violations."""
    import torch
    from sc_gameengine_amd.capi import ScTickError
    grid, S = (2, 1), (4, 4)
    w = sw.generate(S[0] * grid[0], S[1] * grid[1], 15, tiles=grid)              # static props only: group 2 / mask 1
    rng = np.random.default_rng(5)
    roots = np.flatnonzero((w.parent < 0) & (np.arange(w.n) % 16 != 0))
    edge = rng.choice(roots, len(roots) // 6, replace=False)                   # props straddling the shared edge: border records both ways
    w.pos[edge, 0] = (64.0 * S[0] + rng.uniform(-1.0, 1.0, len(edge))).astype(np.float32)
    parts, n = split_world(w, grid, S)
    ticks = [WorldTick.from_world(p, broadphase=True, capacity=n + 8) for p in parts]
    bufs = [tiles.BorderBuffers(t, r, grid, "cuda") for r, t in enumerate(ticks)]
    a, b = ticks
    # (2) a vocabulary narrower than the tile's own layers is refused
    with pytest.raises(ScTickError, match="does not cover"):
        a.set_world_layers(4, 8)
    a.set_world_layers(2, 1)
    # (1) a dynamic body (group 1 / mask all) in a world declared static-only is refused, on upload and on append
    with pytest.raises(ScTickError, match="vocabulary"):
        a.upload_layers(3, np.array([1], np.uint32), np.array([0xFFFFFFFF], np.uint32))
    with pytest.raises(ScTickError, match="vocabulary"):
        a.append_entities(np.zeros((1, 3), np.float32), np.zeros((1, 3), np.float32), np.ones((1, 3), np.float32),
                          group=np.array([4], np.uint32), mask=np.array([8], np.uint32))
    assert a.counts().entities == n                                             # (the refused append left nothing behind)
    a.upload_layers(3, np.array([2], np.uint32), np.array([1], np.uint32))      # inside the vocabulary: fine
    # (3) the neighbour never declared anything and carries 4/8 bodies on the shared edge: its border records break tile a's contract
    k = edge[edge >= n][:5] - n
    assert len(k) == 5
    b.upload_layers(int(k[0]), np.array([4], np.uint32), np.array([8], np.uint32))
    flags = capi.XFORM | capi.BROADPHASE | capi.SPLIT_PAIRS
    seen = []
    for step in range(2):
        for t in ticks:
            t.run(flags)
        for t in ticks:
            t.sync()
        for r, bb in enumerate(bufs):
            for d, nb in tiles.neighbours(r, grid).items():
                bufs[nb].recv[7 - d].copy_(bb.send[d])
        torch.cuda.synchronize()
        for t in ticks:
            t.run_pairs()
        seen.append((a.counts().vocabulary_violations, b.counts().vocabulary_violations))
        if step == 0:
            b.upload_layers(int(k[0]), np.array([2], np.uint32), np.array([1], np.uint32))      # the neighbour mends its ways
    assert seen[0][0] >= 1 and seen[0][1] == 0, seen        # counted on the tile whose contract was broken; b declared nothing
    assert seen[1] == (0, 0), seen                          # per tick: gone once the record is inside the vocabulary again
    for t in ticks:
        t.close()
