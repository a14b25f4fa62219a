"""Tile sharding of the world across GPUs (one process per GPU) and the border-AABB exchange.

The only data-path exchange of the tick is the broadphase's border boxes:
neighbour send/recv over RCCL (torch.distributed backend "nccl" on ROCm), no all-reduce.  Transform
and culling are embarrassingly parallel per tile.  Directions d = 0..7 are (dx, dz) =
(-1,-1) (0,-1) (1,-1) (-1,0) (1,0) (-1,1) (0,1) (1,1); the opposite of d is 7 - d.
"""
import numpy as np

DIRS = [(-1, -1), (0, -1), (1, -1), (-1, 0), (1, 0), (-1, 1), (0, 1), (1, 1)]
TILE_GRIDS = {1: (1, 1), 2: (2, 1), 4: (2, 2), 8: (4, 2)}


def tile_grid(world_size):
    return TILE_GRIDS.get(world_size, (world_size, 1))


def tile_of(rank, grid):
    return rank % grid[0], rank // grid[0]


def neighbours(rank, grid):
    """{direction: neighbour rank} for the tiles that exist around `rank` (row-major tile order)."""
    tx, tz = tile_of(rank, grid)
    out = {}
    for d, (dx, dz) in enumerate(DIRS):
        nx, nz = tx + dx, tz + dz
        if 0 <= nx < grid[0] and 0 <= nz < grid[1]:
            out[d] = nz * grid[0] + nx
    return out


def neighbour_mask(rank, grid):
    m = 0
    for d in neighbours(rank, grid):
        m |= 1 << d
    return m


def exchange(send, recv, rank, grid, group=None):
    """send[d] of this rank -> recv[7-d] of the neighbour in direction d, for every existing neighbour.

    Tensors may live on the GPU (backend nccl = RCCL over xGMI) or on the CPU (gloo, tests).  One
    batched group of point-to-point operations; returns after the receives are ordered on the current
    stream (nccl) or complete (gloo)."""
    import torch.distributed as dist
    nb = neighbours(rank, grid)
    if not nb:
        return
    staged = None
    if dist.get_backend(group) == "gloo" and next(iter(send.values())).is_cuda:
        # rehearsal transport (no RCCL, e.g. several ranks on one GPU): stage through host memory
        staged = recv
        send = {d: send[d].cpu() for d in nb}
        recv = {d: staged[d].cpu() for d in nb}
    ops = []
    for d in sorted(nb):
        ops.append(dist.P2POp(dist.isend, send[d], nb[d], group=group, tag=d))
        # what arrives from the neighbour in direction d left it as its direction 7-d message
        ops.append(dist.P2POp(dist.irecv, recv[d], nb[d], group=group, tag=7 - d))
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    if staged is not None:
        for d in nb:
            staged[d].copy_(recv[d])


class BorderBuffers:
    """Device message buffers of one tile, owned here (torch tensors) and bound into the context.

    All outgoing messages live in ONE tensor, ordered by destination rank, and all incoming ones in another, ordered by
    source rank, so that on RCCL the whole exchange is a single `all_to_all_single` with per-rank split sizes (zero for
    ranks that are not neighbours): one torch operation per step instead of two per neighbour -- the host side of a
    batched isend/irecv group costs about 25 us per operation, 200 us for eight neighbours, four ticks' worth.
    send[d] / recv[d] are views of those tensors per direction (what the single-GPU tile tests copy between contexts)."""

    PIPE_DEPTH = 3          # copies a pipelined tile with a CALLER-run exchange rotates through (scTickSetPairsStream's depth)

    def __init__(self, tick, rank, grid, device, world_size=None, pipelined=False):
        import torch
        self.rank, self.grid = rank, grid
        tick.set_tile(rank, neighbour_mask(rank, grid))
        tx, tz = tile_of(rank, grid)
        tick.set_tile_grid(tx, tz, grid[0], grid[1])      # big boxes travel in the border messages too
        nb = neighbours(rank, grid)
        ranks = grid[0] * grid[1] if world_size is None else world_size
        words = {d: tick.border_bytes(d) // 4 for d in nb}
        self.splits = [0] * ranks                          # the same both ways: the message from the neighbour in direction d
        for d, r in nb.items():                            # is its direction 7-d message, which has the same side length
            self.splits[r] = words[d]
        total = max(sum(self.splits), 1)
        offset, at = {}, 0
        for r in range(ranks):
            offset[r] = at
            at += self.splits[r]
        # pipelined tiles keep one set per tick parity (t mod depth): later ticks pack their messages while tick t's are still in flight
        self.sets = []
        for q in range(self.PIPE_DEPTH if pipelined else 1):
            send_all = torch.zeros(total, dtype=torch.int32, device=device)
            recv_all = torch.zeros(total, dtype=torch.int32, device=device)
            send = {d: send_all[offset[r]:offset[r] + words[d]] for d, r in nb.items()}
            recv = {d: recv_all[offset[r]:offset[r] + words[d]] for d, r in nb.items()}
            self.sets.append((send_all, recv_all, send, recv))
            for d in nb:
                if pipelined:
                    tick.bind_border_parity(q, d, send[d].data_ptr(), recv[d].data_ptr())
                else:
                    tick.bind_border(d, send[d].data_ptr(), recv[d].data_ptr())
        self.send_all, self.recv_all, self.send, self.recv = self.sets[0]

    def exchange(self, group=None, parity=0):
        """Per-step exchange: one all-to-all with split sizes on RCCL; point-to-point staging elsewhere (gloo has no
        all-to-all; it is only used for rehearsals and CPU tests).  parity selects the buffer set of a pipelined tile."""
        import torch.distributed as dist
        if not neighbours(self.rank, self.grid):
            return
        send_all, recv_all, send, recv = self.sets[parity if len(self.sets) > 1 else 0]
        if dist.get_backend(group) != "nccl":
            exchange(send, recv, self.rank, self.grid, group)
            return
        dist.all_to_all_single(recv_all, send_all, self.splits, self.splits, group=group)


def global_pair_ids(pairs, entities_per_rank):
    """rank << 24 | dense index  ->  global dense index (tile-major creation order)."""
    p = np.asarray(pairs, np.uint64)
    return ((p >> np.uint64(24)) & np.uint64(0x7F)) * np.uint64(entities_per_rank) + (p & np.uint64(0xFFFFFF))


def visible_offsets(counts, rank):
    """Result assembly: where rank `rank`'s visible list starts in the global list, and the global list's length, from
    every rank's visible count.  Tiles are created in rank order and entities tile-major, so the concatenation of the tiles' lists
    in rank order is the reference's own order (its serial compaction over the whole pool, sc_world_partition.cpp:1273-1280)."""
    c = np.asarray(counts, np.uint64)
    return int(c[:rank].sum()), int(c.sum())


def global_visible(tick, rank, entities_per_rank, group=None):
    """This rank's slice of the global visible list: (offset, total, global dense indices).  The one collective is the all-gather of
    the visible counts: over the library's own communicator when the context has one (scTickGatherVisibleCounts), else over the
    host's control plane (torch.distributed, any backend; e.g. the one-GPU rehearsal), else the tile is the world."""
    vis = np.asarray(tick.visible(), np.uint64)
    ids = vis + np.uint64(rank) * np.uint64(entities_per_rank)
    if tick.comm_info()["has_communicator"]:
        _, off, total = tick.gather_visible_counts()
        return off, total, ids
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        import torch
        dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
        mine = torch.tensor([len(vis)], dtype=torch.int64, device=dev)
        box = [torch.zeros_like(mine) for _ in range(dist.get_world_size(group))]
        dist.all_gather(box, mine, group=group)
        off, total = visible_offsets([int(b.item()) for b in box], rank)
        return off, total, ids
    return 0, len(ids), ids


def rendezvous_unique_id(rank, make_id, group=None):
    """The one thing the host's own channel has to carry: rank 0's 128-byte communicator id, to every rank.
    Here the channel is torch.distributed (any backend: the payload is a CPU object broadcast); `make_id` is called on
    rank 0 only (capi.comm_unique_id).  Every rank returns the same bytes."""
    import torch.distributed as dist
    box = [make_id() if rank == 0 else None]
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast_object_list(box, src=0, group=group)
    uid = bytes(box[0])
    if len(uid) != 128:
        raise ValueError(f"communicator id must be 128 bytes, got {len(uid)}")
    return uid


def setup_tile(tick, rank, grid, unique_id=None, pipelined=True):
    """Place a context in the tile grid and, when it has neighbours, give it its communicator (library-owned exchange)."""
    tick.set_tile(rank, neighbour_mask(rank, grid))
    tx, tz = tile_of(rank, grid)
    tick.set_tile_grid(tx, tz, grid[0], grid[1])
    if grid[0] * grid[1] > 1:
        if unique_id is None:
            raise ValueError("a tile with neighbours needs the communicator id")
        tick.comm_init(unique_id, grid[0] * grid[1], rank)
        tick.set_pipelined(pipelined)
