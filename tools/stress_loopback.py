#!/usr/bin/env python3
"""The library-owned exchange (loop-back RCCL, scTickTileStep) against a device-copy twin over random centre tiles and
random flows (in order, pipelined depth 2-4, graph replay or not): same pairs, visible lists and counts after every step of
the last three.  python tools/stress_loopback.py [--seeds 16]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sc_gameengine_amd import capi, tiles                       # noqa: E402
from sc_gameengine_amd.tick import WorldTick, camera_view_proj   # noqa: E402
from tests.test_gpu_comm import centre_tile_world, keys, RANK, GRID   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seeds", type=int, default=16)
args = ap.parse_args()
import torch                                                      # noqa: E402

flags = capi.FULL | capi.PRODUCE_NEXT
bad = 0
for seed in range(args.seeds):
    rng = np.random.default_rng(7000 + seed)
    S = int(rng.integers(3, 11))
    pipelined = [False, True, 2, 3, 4][int(rng.integers(0, 5))]
    graph = bool(rng.integers(0, 2))
    steps = int(rng.integers(5, 14))
    nudge = float(rng.choice([0.05, 0.6, 1.7]))
    w = centre_tile_world(S, seed=seed)
    vp = camera_view_proj(w.camera)
    a = WorldTick.from_world(w, broadphase=True, max_pairs=1 << 18)
    a.set_view_proj(vp); a.set_tile(RANK, 0xFF); a.set_tile_grid(1, 1, 3, 3)
    a.comm_init(capi.comm_unique_id(), 1, 0, peers=[0] * 8)
    a.set_pipelined(pipelined); a.set_graph_mode(graph)
    b = WorldTick.from_world(w, broadphase=True, max_pairs=1 << 18)
    b.set_view_proj(vp)
    s2 = torch.cuda.Stream()
    if pipelined:
        b.set_pairs_stream(s2.cuda_stream)
    bufs = tiles.BorderBuffers(b, RANK, GRID, "cuda", pipelined=bool(pipelined))
    for t in (a, b):
        t.set_frame_producer(1, nudge); t.nudge_roots_x(nudge)
    why = None
    for step in range(steps):
        a.tile_step(flags)
        b.run(flags | capi.SPLIT_PAIRS)
        q = (step % len(bufs.sets)) if pipelined else 0
        send, recv = bufs.sets[q][2], bufs.sets[q][3]
        if pipelined:
            with torch.cuda.stream(s2):
                for d in range(8):
                    recv[7 - d].copy_(send[d], non_blocking=True)
        else:
            b.sync()
            for d in range(8):
                recv[7 - d].copy_(send[d])
            torch.cuda.synchronize()
        b.run_pairs()
        if step >= steps - 3:
            ka, kb = keys(a), keys(b)
            ca, cb = a.counts(), b.counts()
            if not np.array_equal(ka, kb):
                why = f"step {step}: {len(ka)} vs {len(kb)} pairs"
            elif not np.array_equal(a.visible(), b.visible()):
                why = f"step {step}: visible lists differ"
            elif (ca.big_boxes, ca.border_lost, ca.bin_overflow) != (cb.big_boxes, cb.border_lost, cb.bin_overflow):
                why = f"step {step}: counts differ"
            if why:
                break
    if why:
        bad += 1
        print(f"seed {seed}: S {S} pipelined {pipelined} graph {graph} steps {steps} nudge {nudge}: {why}", flush=True)
    a.close(); b.close()
print(f"{args.seeds - bad} of {args.seeds} loop-back runs equal their twin")
sys.exit(1 if bad else 0)
