#!/usr/bin/env python3
"""What ONE GPU can show of a tile's step at N > 1: a 1M-entity tile set up as the centre of a 3x3 world (eight neighbours),
its exchange a real RCCL group issued by the library on a one-rank communicator whose peers are all the rank itself
(loop-back: the messages come back as the opposite directions').  Reports, per flow (in order / pipelined):
  us_per_step      wall time per step, device-bound
  host_issue_us    host time to ISSUE one step (scTickTileStep: tick + pack, ncclSend/ncclRecv group, merge + pair search)
and for comparison the old Python-driven step (scTickRun + torch all_to_all_single + scTickRunPairs) when --torch is given."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sc_gameengine_amd import capi, synth_world as sw, tiles          # noqa: E402
from sc_gameengine_amd.tick import WorldTick, camera_view_proj         # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=400)
ap.add_argument("--workload", default="config3")
ap.add_argument("--graph", type=int, default=0)
ap.add_argument("--row", type=int, default=0, help="1: centre tile of a 3x1 world (two neighbours, 4 operations per group) instead of 3x3 (eight, 16)")
ap.add_argument("--only", type=int, default=-1, help="run one flow only: 0 in order, 2..4 pipelined with that depth")
ap.add_argument("--border", type=int, default=0, help="scTickSetBorderCapacity: records per ring sector a message holds on average (0 = the default, 16)")
ap.add_argument("--vocab", type=int, default=1, help="1: declare the world's layer vocabulary (scTickSetWorldLayers), 0: do not")
args = ap.parse_args()

S = 256
w = sw.generate(S, S, 15, origin=(S, S))
w.camera = sw.default_camera(3.0 * S * 64.0)          # the middle of the 3x3 world = the middle of this tile
if args.workload == "config3dyn":
    dyn = (np.arange(w.n) % 16) == 4
    w.group[dyn], w.mask[dyn] = sw.GROUP_DYNAMIC, sw.MASK_ALL
out = {"world": f"{w.n} entities, centre tile of a 3x1 grid, loop-back RCCL (2 sends + 2 receives per step)" if args.row else
                f"{w.n} entities, centre tile of a 3x3 grid, loop-back RCCL (8 sends + 8 receives per step)", "graph_replay": bool(args.graph)}
flags = capi.FULL | capi.PRODUCE_NEXT
for pipelined in (0, 2, 3, 4):
    if args.only >= 0 and pipelined != args.only:
        continue
    t = WorldTick.from_world(w, broadphase=True)
    t.set_view_proj(camera_view_proj(w.camera))
    if args.border:
        t.set_border_capacity(args.border)
    if args.row:
        t.set_tile(1, 0b00011000); t.set_tile_grid(1, 0, 3, 1)
        t.comm_init(capi.comm_unique_id(), 1, 0, peers=[-1, -1, -1, 0, 0, -1, -1, -1])
    else:
        t.set_tile(4, 0xFF); t.set_tile_grid(1, 1, 3, 3)
        t.comm_init(capi.comm_unique_id(), 1, 0, peers=[0] * 8)
    t.set_pipelined(pipelined)
    if args.vocab:
        t.set_world_layers(w.group, w.mask)
    t.set_frame_producer(1, 0.01); t.nudge_roots_x(0.01)
    if args.graph:
        t.set_graph_mode(True)
    for _ in range(30):
        t.tile_step(flags)
    t.sync()
    t.reset_host_times()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        t.tile_step(flags)
    issued = time.perf_counter() - t0
    t.sync()
    total = time.perf_counter() - t0
    key = f"pipelined_depth{pipelined}" if pipelined else "in_order"
    out[key + "_us_per_step"] = round(total / args.steps * 1e6, 2)
    out[key + "_host_issue_us"] = round(issued / args.steps * 1e6, 2)
    ci = t.comm_info()                                     # the library's own clock around each half of the step (scTickGetCommInfo)
    out[key + "_host_tick_half_us"] = round(ci["host_tick_half_us"], 2)
    out[key + "_host_pair_half_us"] = round(ci["host_pair_half_us"], 2)
    # host issue alone, with the device idle between steps (the queue never fills up: what the call itself costs)
    lone = []
    for _ in range(50):
        t.sync()
        t1 = time.perf_counter()
        t.tile_step(flags)
        lone.append(time.perf_counter() - t1)
    t.sync()
    out[key + "_host_issue_idle_queue_us"] = round(float(np.median(lone)) * 1e6, 2)
    # the tick stream's own kernels under this flow (every 8th step sampled: events are not free)
    t.set_profiling(8)
    for _ in range(160):
        t.tile_step(flags)
    t.sync()
    kx, kc = t.kernel_times_ms(capi.K_XFORM_CULL), t.kernel_times_ms(capi.K_COMPACT)
    t.set_profiling(0)
    out[key + "_k_xform_cull_us"] = round(float(np.median(kx)) * 1e3, 2) if len(kx) else None
    out[key + "_k_compact_pack_us"] = round(float(np.median(kc)) * 1e3, 2) if len(kc) else None
    c = t.counts()
    out[key + "_visible"] = int(c.visible)
    out[key + "_border_lost"] = int(c.border_lost)
    t.close()
print(json.dumps(out))
