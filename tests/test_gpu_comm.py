"""The library-owned exchange (scTickCommInit / scTickTileStep) on ONE GPU: a one-rank RCCL communicator whose eight
neighbours are all the rank itself.  What a tile sends in direction d comes back as what it receives from direction 7-d,
through real ncclSend / ncclRecv groups on the streams the tick uses.  A twin context does the same loop-back with device
copies; both must report the same pairs, tick after tick, in the in-order and in the pipelined flow -- so the transport
(buffers, group, posting order, stream order against the kernels around it) is what is being compared."""
import os

import numpy as np
import pytest

from sc_gameengine_amd import capi, synth_world as sw, tiles
from sc_gameengine_amd.tick import WorldTick, camera_view_proj

pytestmark = pytest.mark.gpu

GRID, RANK = (3, 3), 4                 # the centre tile of a 3x3 world: neighbours in all eight directions


def centre_tile_world(S=8, seed=9):
    w = sw.generate(S, S, 15, origin=(S, S))
    rng = np.random.default_rng(seed)
    dyn = rng.random(w.n) < 0.4
    w.group[dyn], w.mask[dyn] = sw.GROUP_DYNAMIC, sw.MASK_ALL
    roots = np.flatnonzero((w.parent < 0) & (np.arange(w.n) % 16 != 0))
    # boxes straddling all four edges of the tile, so every one of the eight messages carries records
    lo, hi = 64.0 * S, 128.0 * S
    for axis, k in ((0, 0), (0, 1), (2, 2), (2, 3)):
        sel = rng.choice(roots, len(roots) // 12, replace=False)
        w.pos[sel, axis] = ((lo if k % 2 == 0 else hi) + rng.uniform(-1.0, 1.0, len(sel))).astype(np.float32)
    big = rng.choice(roots, 6, replace=False)
    w.bmin[big] *= 120.0; w.bmax[big] *= 120.0          # big boxes travel in the messages' big section
    w.group[big], w.mask[big] = sw.GROUP_DYNAMIC, sw.MASK_ALL
    return w


def keys(t):
    p, total = t.pairs()
    assert total == len(p)
    return np.sort(p[:, 0].astype(np.uint64) << np.uint64(32) | p[:, 1].astype(np.uint64))


@pytest.mark.timeout(300)
@pytest.mark.parametrize("pipelined,graph", [(False, False), (True, False), (2, False), (4, False), (False, True), (True, True), (2, True)])
def test_rccl_loopback_equals_device_copy_loopback(pipelined, graph):
    """graph=True: the in-order tile step of A -- fused kernel, compaction + pack, the RCCL group, merge, pair search -- is
    captured once per tick parity and replayed with one hipGraphLaunch (BASELINE config 5: "hipGraph-captured frame").
    pipelined = 2 / 4: that many copies of the per-tick broadphase state on A's side (True = the default, 4).
    pipelined AND graph: each half of a step -- the tick on its stream; RCCL group, merge and pair search on theirs -- is a
    graph of its own, two hipGraphLaunch per step with the ordering events between them."""
    import torch
    w = centre_tile_world()
    vp = camera_view_proj(w.camera)
    flags = capi.FULL | capi.PRODUCE_NEXT

    # A: the library's own exchange
    a = WorldTick.from_world(w, broadphase=True, max_pairs=1 << 17)
    a.set_view_proj(vp)
    a.set_tile(RANK, 0xFF); a.set_tile_grid(1, 1, 3, 3)
    a.comm_init(capi.comm_unique_id(), 1, 0, peers=[0] * 8)
    a.set_pipelined(pipelined)
    a.set_graph_mode(graph)
    # B: the twin, messages moved by device copies
    b = WorldTick.from_world(w, broadphase=True, max_pairs=1 << 17)
    b.set_view_proj(vp)
    s2 = torch.cuda.Stream()
    if pipelined:
        b.set_pairs_stream(s2.cuda_stream)
    bufs = tiles.BorderBuffers(b, RANK, GRID, "cuda", pipelined=pipelined)
    nudge = float(os.environ.get("SC_TICK_TWIN_NUDGE", "0.6"))      # (with many steps take a small one: the boxes must stay around the tile)
    for t in (a, b):
        t.set_frame_producer(1, nudge)
        t.nudge_roots_x(nudge)

    seen = 0
    steps = int(os.environ.get("SC_TICK_TWIN_STEPS", "6"))          # (a long run now and then: SC_TICK_TWIN_STEPS=300)
    for step in range(steps):
        a.tile_step(flags)
        b.run(flags | capi.SPLIT_PAIRS)
        q = (step % len(bufs.sets)) if pipelined else 0
        send, recv = bufs.sets[q][2], bufs.sets[q][3]
        if pipelined:
            with torch.cuda.stream(s2):                     # ordered behind the pack by the library's event
                for d in range(8):
                    recv[7 - d].copy_(send[d], non_blocking=True)
        else:
            b.sync()
            for d in range(8):
                recv[7 - d].copy_(send[d])
            torch.cuda.synchronize()
        b.run_pairs()
        if step >= steps - 3:                               # reading results joins the streams: do it on the last steps only
            ka, kb = keys(a), keys(b)
            assert np.array_equal(ka, kb), f"step {step}: {len(ka)} vs {len(kb)} pairs"
            assert np.array_equal(a.visible(), b.visible())
            # result assembly: the visible counts all-gathered over the library's own communicator (here: one rank)
            counts, off, total = a.gather_visible_counts()
            assert counts.tolist() == [len(a.visible())] and off == 0 and total == len(a.visible())
            assert tiles.global_visible(a, 0, w.n)[1] == total
            ca, cb = a.counts(), b.counts()
            assert (ca.big_boxes, ca.border_lost, ca.bin_overflow) == (cb.big_boxes, cb.border_lost, cb.bin_overflow)
            assert ca.big_boxes >= 6 and len(ka) > 100
            seen += 1
    assert seen == 3
    a.close(); b.close()


def test_tile_with_neighbours_and_no_communicator_fails_loudly():
    w = sw.generate(4, 4, 15)
    t = WorldTick.from_world(w, broadphase=True)
    t.set_tile(0, 0); t.set_tile_grid(0, 0, 2, 1)
    with pytest.raises(capi.ScTickError, match="no communicator"):
        t.tile_step(capi.FULL)
    t.close()


def test_single_tile_step_is_a_plain_tick(oracle):
    from tests import worlds
    w = sw.generate(8, 8, 15)
    ow = worlds.oracle_world(oracle, w)
    vp = camera_view_proj(w.camera)
    t = WorldTick.from_world(w, broadphase=True)
    t.set_view_proj(vp)
    tiles.setup_tile(t, 0, (1, 1))
    ow.transform_system(); ow.culling_system(view_proj=vp)
    t.tile_step(capi.FULL)
    assert np.array_equal(t.visible(), ow.visible())
    t.close(); ow.close()


@pytest.mark.parametrize("graph", [False, True])
def test_a_lone_pipelined_tile_steps_without_a_communicator(oracle, graph):
    """No neighbours, but the pipelined flow is on (a host that configures every tile alike): the step runs the tick on its
    stream and the pair half on the pairs stream -- with graph replay each as one graph -- and equals the in-order tick."""
    w = sw.generate(12, 12, 15)
    dyn = (np.arange(w.n) % 5) == 2
    w.group[dyn], w.mask[dyn] = sw.GROUP_DYNAMIC, sw.MASK_ALL
    vp = camera_view_proj(w.camera)
    flags = capi.FULL | capi.PRODUCE_NEXT
    a = WorldTick.from_world(w, broadphase=True); b = WorldTick.from_world(w, broadphase=True)
    for t in (a, b):
        t.set_view_proj(vp); t.set_frame_producer(1, 0.4); t.nudge_roots_x(0.4)
    a.set_pipelined(True); a.set_graph_mode(graph)
    for step in range(8):
        a.tile_step(flags); b.run(flags)
        if step >= 5:
            assert np.array_equal(keys(a), keys(b)) and len(keys(a)) > 50
            assert np.array_equal(a.visible(), b.visible())
    a.close(); b.close()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("depth", [3, 4])
def test_pipelined_loopback_on_a_tile_without_a_corner_neighbour(depth):
    """A tile of a 2x1 / 1x2 world has no (-1,-1) neighbour.  Round 2's scTickSetPairsStream took "direction 0 unbound" for
    "bound without a parity" and overwrote parities 2 and 3 with parity 1's buffers there -- three ticks in flight shared one
    message set.  Here: neighbours in directions 3 and 4 only (both the rank itself), comm first and pipelining second as
    tiles.setup_tile does, many steps against the device-copy twin, and the parities' buffers must all differ."""
    import torch
    w = centre_tile_world(seed=11)
    vp = camera_view_proj(w.camera)
    flags = capi.FULL | capi.PRODUCE_NEXT
    MASK = (1 << 3) | (1 << 4)
    peers = [-1, -1, -1, 0, 0, -1, -1, -1]

    a = WorldTick.from_world(w, broadphase=True, max_pairs=1 << 17)
    a.set_view_proj(vp)
    a.set_tile_grid(1, 0, 3, 1)                       # the middle tile of a 3x1 world: mask = directions 3 and 4
    a.comm_init(capi.comm_unique_id(), 1, 0, peers=peers)
    a.set_pipelined(depth)
    for d in (3, 4):
        for recv in (False, True):
            ptrs = [a.border_buffer(q, d, recv) for q in range(depth)]
            assert all(ptrs) and len(set(ptrs)) == depth, f"direction {d}: parities share a buffer: {ptrs}"
    for d in (0, 1, 2, 5, 6, 7):
        assert a.border_buffer(1, d) == 0

    b = WorldTick.from_world(w, broadphase=True, max_pairs=1 << 17)
    b.set_view_proj(vp)
    b.set_tile_grid(1, 0, 3, 1)
    s2 = torch.cuda.Stream()
    b.set_pairs_stream(s2.cuda_stream)
    words = {d: b.border_bytes(d) // 4 for d in (3, 4)}
    sets = []
    for q in range(3):                                # the caller-run twin rotates through the default three copies
        send = {d: torch.zeros(words[d], dtype=torch.int32, device="cuda") for d in (3, 4)}
        recv = {d: torch.zeros(words[d], dtype=torch.int32, device="cuda") for d in (3, 4)}
        for d in (3, 4):
            b.bind_border_parity(q, d, send[d].data_ptr(), recv[d].data_ptr())
        sets.append((send, recv))
    for t in (a, b):
        t.set_frame_producer(1, 0.05)
        t.nudge_roots_x(0.05)
    steps = 40
    for step in range(steps):
        a.tile_step(flags)
        b.run(flags | capi.SPLIT_PAIRS)
        send, recv = sets[step % 3]
        with torch.cuda.stream(s2):
            for d in (3, 4):
                recv[7 - d].copy_(send[d], non_blocking=True)
        b.run_pairs()
        if step % 8 == 7 or step >= steps - 2:
            ka, kb = keys(a), keys(b)
            assert np.array_equal(ka, kb), f"step {step}: {len(ka)} vs {len(kb)} pairs"
            assert len(ka) > 50
    a.close(); b.close()


def test_pipeline_depth_is_validated():
    w = sw.generate(4, 4, 15)
    t = WorldTick.from_world(w, broadphase=True)
    for bad in (-1, -7, 5):
        with pytest.raises(capi.ScTickError, match="pipeline depth"):
            t.set_pipelined(bad)
    t.close()
