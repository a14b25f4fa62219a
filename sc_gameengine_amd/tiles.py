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
    """Device message buffers of one tile, owned here (torch tensors) and bound into the context."""

    def __init__(self, tick, rank, grid, device):
        import torch
        self.rank, self.grid = rank, grid
        self.send, self.recv = {}, {}
        tick.set_tile(rank, neighbour_mask(rank, grid))
        tx, tz = tile_of(rank, grid)
        tick.set_tile_grid(tx, tz, grid[0], grid[1])      # big boxes travel in the border messages too
        for d in neighbours(rank, grid):
            nbytes = tick.border_bytes(d)
            self.send[d] = torch.zeros(nbytes // 4, dtype=torch.int32, device=device)
            self.recv[d] = torch.zeros(nbytes // 4, dtype=torch.int32, device=device)
            tick.bind_border(d, self.send[d].data_ptr(), self.recv[d].data_ptr())

        self._ops = None

    def exchange(self, group=None):
        """Per-step exchange.  On RCCL the batched op list is built once and reused (the tensors are
        persistent), which keeps the per-step host cost to one batch_isend_irecv call."""
        import torch.distributed as dist
        nb = neighbours(self.rank, self.grid)
        if not nb:
            return
        if dist.get_backend(group) != "nccl":
            exchange(self.send, self.recv, self.rank, self.grid, group)
            return
        if self._ops is None:
            self._ops = []
            for d in sorted(nb):
                self._ops.append(dist.P2POp(dist.isend, self.send[d], nb[d], group=group))
                self._ops.append(dist.P2POp(dist.irecv, self.recv[d], nb[d], group=group))
        for w in dist.batch_isend_irecv(self._ops):
            w.wait()


def global_pair_ids(pairs, entities_per_rank):
    """rank << 24 | dense index  ->  global dense index (tile-major creation order)."""
    p = np.asarray(pairs, np.uint64)
    return ((p >> np.uint64(24)) & np.uint64(0x7F)) * np.uint64(entities_per_rank) + (p & np.uint64(0xFFFFFF))
