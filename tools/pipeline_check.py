#!/usr/bin/env python3
"""On-device effect of pipelining a tile's step (no network: the neighbours' messages stay empty): one 1M-entity tile set up
as tile (0,0) of a 2x2 world, split flow, with and without a pairs stream.  An artificial delay kernel on the pairs stream
stands in for the exchange latency."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sc_gameengine_amd import capi, synth_world as sw, tiles
from sc_gameengine_amd.tick import WorldTick, camera_view_proj

w = sw.config("config3")
out = {}
only = sys.argv[1] if len(sys.argv) > 1 else None          # e.g. "40p": one case (for a kernel trace)
for delay_us in (0, 40):
    for pipelined in (False, True):
        if only and only != f"{delay_us}{'p' if pipelined else 'i'}":
            continue
        t = WorldTick.from_world(w, broadphase=True)
        t.set_view_proj(camera_view_proj(w.camera))
        s2 = torch.cuda.Stream()
        torch.cuda.set_stream(s2)                       # torch's current stream: where the exchange is issued
        if pipelined:
            t.set_pairs_stream(s2.cuda_stream)          # the tick itself stays on the context's own stream
        else:
            t.set_stream(s2.cuda_stream, external=True)
        b = tiles.BorderBuffers(t, 0, (2, 2), "cuda", pipelined=pipelined)
        t.set_frame_producer(1, 0.01); t.nudge_roots_x(0.01)
        flags = capi.FULL | capi.SPLIT_PAIRS | capi.PRODUCE_NEXT
        spin = torch.zeros(1, device="cuda")
        def fake_exchange():
            if delay_us:
                torch.cuda._sleep(int(delay_us * 2100))          # ~cycles at 2.1 GHz
        def step():
            t.run(flags)
            fake_exchange()
            t.run_pairs()
        for _ in range(30): step()
        t.sync(); torch.cuda.synchronize()
        n = 300 if not only else 20
        t0 = time.perf_counter()
        for _ in range(n): step()
        issued = time.perf_counter() - t0
        t.sync(); torch.cuda.synchronize()
        key = f"delay{delay_us}us_{'pipelined' if pipelined else 'in_order'}"
        out[key + "_us_per_step"] = round((time.perf_counter() - t0) / n * 1e6, 2)
        out[key + "_host_issue_us"] = round(issued / n * 1e6, 2)
        c = t.counts(); assert c.visible > 0
        t.close()
print(json.dumps(out))
