"""N>1 host logic on CPU: tile <-> rank maps, and the border exchange as real point-to-point
operations between processes (gloo, world_size 2 and 4).  send[d] of a rank must arrive as recv[7-d] of
the neighbour in direction d and nowhere else."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sc_gameengine_amd import tiles


def test_tile_maps():
    assert tiles.tile_grid(1) == (1, 1) and tiles.tile_grid(2) == (2, 1) and tiles.tile_grid(4) == (2, 2) and tiles.tile_grid(8) == (4, 2)
    assert tiles.neighbours(0, (1, 1)) == {} and tiles.neighbour_mask(0, (1, 1)) == 0
    assert tiles.neighbours(0, (2, 1)) == {4: 1} and tiles.neighbours(1, (2, 1)) == {3: 0}
    assert tiles.neighbours(0, (2, 2)) == {4: 1, 6: 2, 7: 3}
    assert tiles.neighbours(3, (2, 2)) == {0: 0, 1: 1, 3: 2}
    nb = tiles.neighbours(5, (4, 2))                       # tile (1,1) of a 4x2 grid
    assert nb == {0: 0, 1: 1, 2: 2, 3: 4, 4: 6}
    for ws in (2, 4, 8):
        g = tiles.tile_grid(ws)
        for r in range(ws):
            for d, n in tiles.neighbours(r, g).items():
                assert tiles.neighbours(n, g)[7 - d] == r          # symmetric: I am my neighbour's opposite direction
            assert bin(tiles.neighbour_mask(r, g)).count("1") == len(tiles.neighbours(r, g))
    p = np.array([[5 | (1 << 24), 7 | (3 << 24)]], np.uint32)
    assert tiles.global_pair_ids(p, 1000).tolist() == [[1005, 3007]]


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        grid = tiles.tile_grid(world)
        nb = tiles.neighbours(rank, grid)
        send = {d: torch.full((64,), rank * 16 + d, dtype=torch.int32) for d in nb}
        recv = {d: torch.full((64,), -1, dtype=torch.int32) for d in nb}
        for rounds in range(3):                              # repeated exchanges stay matched
            tiles.exchange(send, recv, rank, grid)
            for d, n in nb.items():
                want = n * 16 + (7 - d)
                assert bool((recv[d] == want).all()), f"rank {rank} dir {d}: got {recv[d][0].item()} want {want}"
        t = torch.tensor([float(rank)])
        dist.all_reduce(t, op=dist.ReduceOp.MAX)             # the bench's max-over-ranks timing reduction
        assert t.item() == world - 1
        out.put((rank, "ok"))
    except Exception as e:                                   # noqa: BLE001
        out.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_border_exchange_over_gloo(world):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    results = [out.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(r, "ok") for r in range(world)], results


def _rendezvous_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        calls = []

        def make_id():                                       # stands in for capi.comm_unique_id (needs a GPU): rank 0 only
            calls.append(rank)
            return bytes((7 * i + 3) % 256 for i in range(128))

        uid = tiles.rendezvous_unique_id(rank, make_id)
        assert uid == bytes((7 * i + 3) % 256 for i in range(128))
        assert calls == ([0] if rank == 0 else [])
        try:
            tiles.rendezvous_unique_id(rank, lambda: b"short")
            bad = False
        except ValueError:
            bad = True
        assert bad                                           # a malformed id is refused on every rank
        out.put((rank, "ok"))
    except Exception as e:                                   # noqa: BLE001
        out.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_communicator_id_rendezvous_over_gloo():
    """bench.py's only N>1 host duty besides launching: rank 0's 128-byte RCCL id reaches every rank unchanged."""
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_rendezvous_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    results = [out.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(r, "ok") for r in range(world)], results


def test_rendezvous_without_a_process_group_is_local():
    assert tiles.rendezvous_unique_id(0, lambda: bytes(128)) == bytes(128)


class _FakeTick:
    """what tiles.global_visible needs of a context, without a GPU: a visible list and no communicator"""
    def __init__(self, vis):
        self._vis = np.asarray(vis, np.uint32)

    def visible(self):
        return self._vis

    def comm_info(self):
        return {"has_communicator": 0, "world_size": 0}


def _visible_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_per = 1000
        rng = np.random.default_rng(7)                        # every rank draws the same world: rank r's tile sees lists[r]
        lists = [np.sort(rng.choice(n_per, size=int(rng.integers(0, 40)), replace=False)).astype(np.uint32) for _ in range(world)]
        off, total, ids = tiles.global_visible(_FakeTick(lists[rank]), rank, n_per)
        want = np.concatenate([lists[r].astype(np.uint64) + r * n_per for r in range(world)])
        assert total == len(want) and off == sum(len(lists[r]) for r in range(rank))
        assert np.array_equal(want[off:off + len(ids)], ids)          # the rank's slice sits at its offset of the whole-world list
        # rank 0 assembles the whole list from the slices, each placed at its own offset
        box = [None] * world
        dist.all_gather_object(box, (off, ids.tolist()))
        if rank == 0:
            whole = np.zeros(total, np.uint64)
            for o, part in box:
                whole[o:o + len(part)] = part
            assert np.array_equal(whole, want)
        out.put((rank, "ok"))
    except Exception as e:                                   # noqa: BLE001
        out.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_global_visible_list_assembly_over_gloo(world):
    """SURVEY 8e, result assembly: all-gather of the ranks' visible counts -> offsets -> every rank's slice, with global dense
    indices, at its own offset of the whole-world list (tile-major order = the reference's compaction order)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_visible_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = [out.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res


def test_visible_offsets():
    assert tiles.visible_offsets([5, 0, 7, 2], 0) == (0, 14)
    assert tiles.visible_offsets([5, 0, 7, 2], 2) == (5, 14)
    assert tiles.visible_offsets([5, 0, 7, 2], 3) == (12, 14)
