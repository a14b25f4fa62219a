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
print(f"rccl self test ok: {nbytes} B message, {dt * 1e6:.1f} us per step (2 ticks + 1 send/recv group), visible {t.counts().visible}")
dist.destroy_process_group()
