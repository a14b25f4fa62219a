#!/usr/bin/env python3
"""Host and device cost of the border exchange's transport on a 1-GPU box (self-send): a batched isend/irecv group with
1 / 3 / 8 pairs against one all_to_all_single, message sizes of the border exchange."""
import os, sys, time
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
stream = torch.cuda.Stream(device=0); torch.cuda.set_stream(stream)

def measure(name, call):
    for _ in range(20): call()
    stream.synchronize()
    n = 200
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    host = 0.0
    e0.record()
    for _ in range(n):
        h0 = time.perf_counter(); call(); host += time.perf_counter() - h0
    e1.record(); stream.synchronize()
    print(f"{name}: host {host / n * 1e6:.1f} us per call, device {e0.elapsed_time(e1) / n * 1e3:.1f} us per call back to back")

for nops, words in ((1, 33000), (3, 33000), (8, 33000)):
    send = [torch.arange(words, dtype=torch.int32, device="cuda") for _ in range(nops)]
    recv = [torch.zeros(words, dtype=torch.int32, device="cuda") for _ in range(nops)]
    ops = []
    for k in range(nops):
        ops += [dist.P2POp(dist.isend, send[k], 0, tag=k), dist.P2POp(dist.irecv, recv[k], 0, tag=k)]
    def p2p():
        for wk in dist.batch_isend_irecv(ops): wk.wait()
    measure(f"batch_isend_irecv, {nops} pairs x {words * 4} B", p2p)
for words in (33000, 8 * 33000):
    a = torch.arange(words, dtype=torch.int32, device="cuda"); b = torch.zeros_like(a)
    measure(f"all_to_all_single, {words * 4} B", lambda: dist.all_to_all_single(b, a, [words], [words]))
    assert torch.equal(a, b)
dist.destroy_process_group()
