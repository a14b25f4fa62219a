#!/usr/bin/env python3
"""What a 1-GPU box can check of the RCCL path bench.py uses at N > 1: communicator creation through
torch.distributed ("nccl" IS RCCL on ROCm) with device_id, a batched isend/irecv group (to the own rank) on a side
stream that the tick context has adopted, stream-ordered with tick kernels, plus the all-reduce and barrier calls."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch, torch.distributed as dist
from sc_gameengine_amd import capi, synth_world as sw
from sc_gameengine_amd.tick import WorldTick, camera_view_proj

torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
w = sw.generate(16, 16, 15)
t = WorldTick.from_world(w, broadphase=True)
t.set_view_proj(camera_view_proj(w.camera))
stream = torch.cuda.Stream(device=0)
torch.cuda.set_stream(stream)
t.set_stream(stream.cuda_stream, external=True)
t.set_tile(0, 0); t.set_tile_grid(0, 0, 1, 1)
nbytes = t.border_bytes(4)
send = torch.arange(nbytes // 4, dtype=torch.int32, device="cuda")
recv = torch.zeros_like(send)
ops = [dist.P2POp(dist.isend, send, 0), dist.P2POp(dist.irecv, recv, 0)]
t0 = time.perf_counter()
for step in range(50):
    t.run(capi.FULL)
    for wk in dist.batch_isend_irecv(ops):
        wk.wait()
    t.run(capi.XFORM | capi.CULL)
stream.synchronize()
dt = (time.perf_counter() - t0) / 50
assert torch.equal(send, recv)
flag = torch.tensor([1], dtype=torch.int32, device="cuda")
dist.all_reduce(flag, op=dist.ReduceOp.MIN)
dist.barrier()
torch.cuda.synchronize()
# the pipelined flow bench.py uses at N > 1: torch's current stream is the PAIRS stream (the RCCL operation goes there),
# the tick keeps the context's own stream; one all_to_all_single per step
w2 = sw.generate(32, 32, 15)
w2.group[::3], w2.mask[::3] = sw.GROUP_DYNAMIC, sw.MASK_ALL
ref = WorldTick.from_world(w2, broadphase=True, max_pairs=1 << 18)
ref.set_frame_producer(1, 0.5); ref.nudge_roots_x(0.5)
for _ in range(12):
    ref.run(capi.FULL | capi.PRODUCE_NEXT)
want_pairs = ref.counts().pairs
p2 = WorldTick.from_world(w2, broadphase=True, max_pairs=1 << 18)
ps = torch.cuda.Stream(device=0); torch.cuda.set_stream(ps)
p2.set_pairs_stream(ps.cuda_stream)
p2.set_tile(0, 0); p2.set_tile_grid(0, 0, 1, 1)
p2.set_frame_producer(1, 0.5); p2.nudge_roots_x(0.5)
a2a_in = torch.arange(33000, dtype=torch.int32, device="cuda"); a2a_out = torch.zeros_like(a2a_in)
t0 = time.perf_counter()
for _ in range(12):
    p2.run(capi.FULL | capi.SPLIT_PAIRS | capi.PRODUCE_NEXT)
    dist.all_to_all_single(a2a_out, a2a_in, [33000], [33000])
    p2.run_pairs()
p2.sync(); torch.cuda.synchronize()
dt2 = (time.perf_counter() - t0) / 12
got_pairs = p2.counts().pairs
assert torch.equal(a2a_in, a2a_out) and got_pairs == want_pairs and want_pairs > 100, (got_pairs, want_pairs)
print(f"pipelined flow with an all-to-all on the pairs stream ok: {got_pairs} pairs after 12 ticks, {dt2 * 1e6:.1f} us per step")
print(f"rccl self test ok: {nbytes} B message, {dt * 1e6:.1f} us per step (2 ticks + 1 send/recv group), visible {t.counts().visible}")
dist.destroy_process_group()
