#!/usr/bin/env python3
"""bench.py -- entities/sec of one world tick on MI355X (BASELINE.json metric).

A step = the upstream producer (every root's localPos.x += 0.01, marked dirty: SynthWorld dirty
regime (ii)) followed by one pass of the hot path over the resident world: TransformSystem +
CullingSystem (+ broadphase once built), through the C ABI.  Workload at N=1: SynthWorld v1
config 3 (256x256 sectors, 15 props + ground slab each = 1 048 576 entities, depths 0/1/2).
With --gpus N the world is N tiles of that size (weak scaling), one process per GPU.

Prints ONE JSON line (rank 0).  The oracle is used only for the cpu_baseline leg.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
TILE_SECTORS = 256
PROPS = 15


def algorithmic_bytes_per_entity(child_frac, stages):
    """Per-launch algorithmic bytes of the dominant kernel k_xform_cull:
    xform 88 + 48*C/N; cull +24 (bounds; the 4*V/N index list is written by k_compact);
    broadphase +32 (the AABB record written into its sector bin; bounds are read once for both).
    SURVEY 8d's further 96 B/entity of broadphase traffic (dense AABB array, sort scatter) do not
    exist in this design -- boxes are binned directly -- and are NOT counted anywhere."""
    b = 88.0 + 48.0 * child_frac
    if "cull" in stages or "broadphase" in stages:
        b += 24.0
    if "broadphase" in stages:
        b += 32.0
    return b


def pmc_traffic(stages, entities_per_gpu, workload="config3"):
    """HBM bytes per k_xform_cull launch from the committed rocprofv3 PMC passes (tools/pmc_session.sh ->
    profiles/pmc_traffic.json): FETCH_SIZE and WRITE_SIZE collected in separate passes and calibrated on
    known-byte copy kernels of the same access widths (FETCH_SIZE under-counts 2x on gfx950).  None when
    no profile of this exact workload is committed."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        d = json.load(open(path))
    except Exception:
        return None
    cfg = d.get("bench_config", {})
    if sorted(cfg.get("stages", [])) != sorted(stages) or cfg.get("entities_per_gpu") != entities_per_gpu or cfg.get("workload", "config3") != workload:
        return None
    return d.get("kernels", {}).get("k_xform_cull", {}).get("hbm_bytes_per_launch")


def copy_ceiling_gbs(torch, device):
    """Measured device copy rate (read + write bytes per second) of a 1 GiB buffer, the practical HBM
    ceiling SURVEY 8d asks to report next to the vendor peak."""
    n = 1 << 28
    a = torch.empty(n, dtype=torch.float32, device=device)
    b = torch.empty_like(a)
    a.fill_(1.0)
    for _ in range(3):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    del a, b
    return 2.0 * n * 4 / (ms * 1e-3) / 1e9


def cpu_baseline(world, ticks=20, warm=3, workers=None):
    """Reference-faithful CPU tick (oracle port), timed on this host: Transform + Camera + Culling,
    hardware_concurrency()-1 workers as the sandbox does (src/sandbox/src/main.cpp:52-54)."""
    from oracle import oracle_py as oracle
    oracle.build()
    hw = os.cpu_count() or 1
    workers = max(hw - 1, 1) if workers is None else workers
    oracle.lib().orc_jobs_init(workers)
    ow = oracle.OracleWorld.from_arrays(world.pos, world.rot, world.scale, world.parent, world.bmin, world.bmax,
                                        has_mesh=world.has_mesh, has_bounds=world.has_bounds)
    ow.add_camera_entity(world.camera["pos"], world.camera["rot"], aspect=world.camera["aspect"])
    times = []
    vel = None if world.mover_kind is None else world.mover_vel.copy()
    for k in range(warm + ticks):
        if vel is None:
            ow.nudge_roots_x(0.01)
        else:
            ow.advance_movers(world.mover_kind, vel, world.mover_lo, world.mover_hi, 1.0 / 60.0)
        t0 = time.perf_counter()
        ow.tick()
        dt = time.perf_counter() - t0
        if k >= warm:
            times.append(dt)
    vis = len(ow.visible())
    ow.close()
    oracle.lib().orc_jobs_init(0)
    med = float(np.median(times))
    return {"value": world.n / med, "unit": "entities/s", "cores": workers + 1, "kind": "port",
            "sample": f"same world ({world.n} entities), {warm} warm-up + {ticks} timed ticks, "
                      f"{'all roots nudged' if vel is None else 'movers advanced'} each tick, "
                      f"median tick {med * 1e3:.1f} ms (xform+camera+cull), host cpus {hw}",
            "visible": vis}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--sectors", type=int, default=TILE_SECTORS, help="tile side in sectors (default 256 = 1M entities per GPU)")
    ap.add_argument("--stages", default="auto", help="comma list of xform,cull,broadphase (auto = all that are built)")
    ap.add_argument("--graph", type=int, default=0, help="replay the frame from a hipGraph")
    ap.add_argument("--workload", default="config3", choices=["config3", "config5"],
                    help="config3: 16 static entities per sector, every root nudged each step (the metric's config); "
                         "config5: 16 static + 12 vehicles + 4 peds per sector, agents advanced on device each step")
    ap.add_argument("--sample", type=int, default=8, help="record HIP events on every n-th step (1 = all)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pipeline", type=int, default=1, help="N>1: pair-search half of tick t on a second stream under the fused kernel of tick t+1")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank uses GPU 0 (needs --backend gloo)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from sc_gameengine_amd import capi, synth_world as sw, tiles
    from sc_gameengine_amd.tick import WorldTick, camera_view_proj

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world_size != args.gpus:
        if world_size == 1 and args.gpus > 1:
            sys.exit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world_size > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    grid = tiles.tile_grid(world_size)
    tx, tz = grid
    S = args.sectors
    origin = ((rank % tx) * S, (rank // tx) * S)
    if args.workload == "config5":
        SX, SZ = S // 2, S                       # 128 x 256 sectors x 32 entities = the same 1 048 576 per GPU
        origin = ((rank % tx) * SX, (rank // tx) * SZ)
        w = sw.generate_config5(SX, SZ, origin=origin)
    else:
        SX, SZ = S, S
        w = sw.generate(S, S, PROPS, hierarchy=True, origin=origin)
    cam = sw.default_camera(float(tx * SX) * 64.0)
    cam["pos"][2] = np.float32(float(tz * SZ) * 64.0 / 2)
    w.camera = cam

    built = ["xform", "cull"] + (["broadphase"] if capi.HAVE_PAIR_SEARCH else [])
    stages = built if args.stages == "auto" else args.stages.split(",")
    flags = 0
    if "xform" in stages:
        flags |= capi.XFORM
    if "cull" in stages:
        flags |= capi.CULL
    if "broadphase" in stages:
        flags |= capi.BROADPHASE

    t = WorldTick.from_world(w, device=local_rank, broadphase=("broadphase" in stages))
    t.set_view_proj(camera_view_proj(cam))
    t.set_graph_mode(bool(args.graph))

    # N > 1: the broadphase's border boxes are the one exchange on the path.  The context runs on
    # torch's current stream so the RCCL send/recv group is stream-ordered with the kernels around it
    # (no host synchronisation inside a step).
    borders = None
    tick_stream = pairs_stream = None
    if world_size > 1 and (flags & capi.BROADPHASE):
        if args.pipeline:
            # the exchange, the merge and the pair search of tick t run on a second stream under the fused kernel of tick t+1
            # (bins and messages are double-buffered by tick parity; the library orders the halves with events).  That
            # second stream is torch's current stream, where the RCCL operation goes; the tick keeps the context's own.
            pairs_stream = torch.cuda.Stream(device=local_rank)
            torch.cuda.set_stream(pairs_stream)
            t.set_pairs_stream(pairs_stream.cuda_stream)
        else:
            tick_stream = torch.cuda.Stream(device=local_rank)
            torch.cuda.set_stream(tick_stream)        # torch's current stream for everything below, RCCL ops included
            t.set_stream(tick_stream.cuda_stream, external=True)
        borders = tiles.BorderBuffers(t, rank, grid, torch.device("cuda", local_rank), pipelined=bool(args.pipeline))
    tick_parity = [0]

    # The frame producer (config 3: every root nudged; config 5: vehicles and peds advanced) is part of every step.  It
    # runs fused into the end-of-tick kernel as the producer of the NEXT frame (SC_TICK_PRODUCE_NEXT): same work per
    # step, one launch fewer.  The first frame's producer runs once, explicitly, before the first step.
    kind, param = (2, 1.0 / 60.0) if args.workload == "config5" else (1, 0.01)
    t.set_frame_producer(kind, param)
    if kind == 2:
        t.advance_movers(param)
    else:
        t.nudge_roots_x(param)
    flags |= capi.PRODUCE_NEXT

    def step():
        if borders is None:
            t.run(flags)
            return
        t.run(flags | capi.SPLIT_PAIRS)           # ... bins filled, border messages packed, next frame produced
        borders.exchange(parity=tick_parity[0])   # neighbour messages over RCCL (xGMI), one all-to-all, on the current stream
        if pairs_stream is not None:
            tick_parity[0] ^= 1
        t.run_pairs()                             # merge what arrived, pair search (on the pairs stream when pipelined)

    def fence():
        t.sync()
        torch.cuda.synchronize()
        if world_size > 1:
            dist.barrier()
            torch.cuda.synchronize()

    exchange_note = None
    if borders is not None:
        # one guarded step first: if the RCCL point-to-point group cannot run on this node, every rank learns it
        # (min-reduce of a flag) and the broadphase stage is dropped -- and reported as dropped -- instead of
        # crashing the whole scaling measurement
        ok = 1
        try:
            step()
            fence()
        except Exception as e:                                  # noqa: BLE001
            ok = 0
            exchange_note = f"border exchange failed ({type(e).__name__}: {e}); broadphase stage dropped"
            print(exchange_note, file=sys.stderr)
        flag = torch.tensor([ok], dtype=torch.int32, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            exchange_note = exchange_note or "border exchange failed on another rank; broadphase stage dropped"
            borders = None
            stages = [x for x in stages if x != "broadphase"]
            flags &= ~capi.BROADPHASE
            if pairs_stream is not None:
                t.set_pairs_stream(0)
                pairs_stream = None
            t.set_stream(0, external=False)
    for _ in range(args.warmup):
        step()
    fence()
    t.set_profiling(args.sample)          # HIP events on every n-th tick, inside the timed region
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    k1 = t.kernel_times_ms(capi.K_XFORM_CULL)
    k2 = t.kernel_times_ms(capi.K_COMPACT)
    kn = t.kernel_times_ms(capi.K_NUDGE)
    kp = t.kernel_times_ms(capi.K_PAIRS)
    t.set_profiling(0)
    counts = t.counts()

    if world_size > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        n_total = w.n * world_size
        child_frac = float((w.parent >= 0).mean())
        bpe = algorithmic_bytes_per_entity(child_frac, stages)
        if args.workload == "config5":
            # only the movers (half the world, all roots) are rebuilt: they read 40 B of locals and write 48 B;
            # the clean half re-reads its stored matrix (48 B) for the sphere test / AABB instead
            bpe = 0.5 * 88.0 + 0.5 * 48.0 + (24.0 if ("cull" in stages or "broadphase" in stages) else 0.0) + (32.0 if "broadphase" in stages else 0.0)
        k1_ms = float(np.mean(k1)) if len(k1) else float("nan")
        achieved = (w.n * bpe) / (k1_ms * 1e-3) / 1e9 if len(k1) else None
        out = {
            "metric": "entities/sec world-tick (xform+broadphase+cull), 1M-entity world",
            "value": n_total * args.steps / elapsed,
            "unit": "entities/s",
            "n_gpus": world_size,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": (f"SynthWorld v1 config 3 per GPU: {SX}x{SZ} sectors x (15 props + ground) = {w.n} entities, "
                             f"depths 0/1/2, every root nudged +0.01 m in x and marked dirty each step") if args.workload == "config3" else
                            (f"SynthWorld v1 config 5 per GPU: {SX}x{SZ} sectors x (ground + 15 props + 12 vehicles + 4 peds) = {w.n} "
                             f"entities, vehicles and peds advanced on device each step (dt 1/60), props static"),
                "stages": stages,
                "producer": "per step, fused into the end-of-tick kernel as the next frame's producer",
                "tiles": f"{tx}x{tz}",
                "entities_total": n_total,
                "visible": int(counts.visible),
                "pairs": int(counts.pairs),
                "graph": bool(args.graph),
                "pipelined": bool(pairs_stream is not None),
                "exchange": (exchange_note or ("border AABBs to <=8 neighbour tiles per step, one RCCL all-to-all with split sizes" if borders is not None else "none")),
                "resident": "device SoA authoritative; no per-step host transfer",
                "backend": args.backend if world_size > 1 else None,
                "rehearsal_same_device": bool(args.same_device),
            },
            "roofline": {
                "bound": "hbm", "kernel": "k_xform_cull",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                "traffic": pmc_traffic(stages, w.n, args.workload),
                "bytes_per_entity": bpe, "avg_launch_ms": k1_ms, "launches_timed": int(len(k1)),
                "other_kernels_ms": {"k_compact": float(np.mean(k2)) if len(k2) else None,
                                     "frame producer": float(np.mean(kn)) if len(kn) else "fused into the end-of-tick kernel (SC_TICK_PRODUCE_NEXT)",
                                     "k_compact_pairs (one launch, with the next frame's producer)" if not len(k2) else "k_pairs": float(np.mean(kp)) if len(kp) else None},
            },
        }
        if world_size == 1:
            ceiling = copy_ceiling_gbs(torch, torch.device("cuda", local_rank))
            out["roofline"]["copy_ceiling"] = ceiling
            out["roofline"]["frac_of_copy_ceiling"] = (achieved / ceiling) if achieved else None
            # resident mode's per-frame read-back: the visible list and the matrices of the visible entities
            for _ in range(2):                      # the first call allocates scratch; report the warm one
                t0 = time.perf_counter()
                vis = t.visible()
                mats = t.world_matrices_indexed(vis[:65536])
                out["config"]["readback_ms_visible_list_and_matrices"] = (time.perf_counter() - t0) * 1e3
            del mats
        if world_size == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w)
            one = cpu_baseline(w, ticks=5, warm=1, workers=0)
            out["cpu_baseline"]["single_thread_value"] = one["value"]
            out["cpu_baseline"]["single_thread_sample"] = one["sample"]
        print(json.dumps(out))
    t.close()
    if world_size > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
