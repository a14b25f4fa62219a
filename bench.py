#!/usr/bin/env python3
"""bench.py -- entities/sec of one world tick on MI355X (BASELINE.json metric).

A step = one pass of the hot path over the resident world -- TransformSystem + CullingSystem + AABB broadphase,
through the C ABI -- followed by the upstream producer of the NEXT frame (config 3: every root's localPos.x += 0.01,
marked dirty: SynthWorld dirty regime (ii); config 5: vehicles and peds advanced), which rides on the end-of-tick kernel.
Workload at N=1: SynthWorld v1 config 3 (256x256 sectors, 15 props + ground slab each = 1 048 576 entities, depths
0/1/2).  With --gpus N the world is N tiles of that size (weak scaling), one process per GPU; the only exchange on
the path, the broadphase's border boxes, is issued by the library itself (its own RCCL communicator: one group of
ncclSend / ncclRecv per step).  This script's N>1 duties are the rendezvous of the communicator id and the timing.

Parity gate IN THE RUN: after the timed region the oracle (liboracle.so, CPU) is brought to the same frame
-- the same number of producer steps on the same world -- and the visible list, every world matrix and the pair set are
compared with what the GPU holds.  A mismatch exits non-zero; no number is printed.  The oracle is used for that and
for the cpu_baseline leg only, never inside the timed region.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (vendor peak); the copy ceiling is measured below
PROFILE_ROUND = "r04"          # profiles/<round>/pmc_traffic_<workload>.json: the PMC passes the `traffic` fields are read from
TILE_SECTORS = 256
PROPS = 15


def records_written_frac(t):
    """(share of the broadphase records of REBUILT entities the fused kernel writes on an ordinary tick, whether records of
    entities that were not rebuilt stay unwritten) from scTickGetBinStats: (1, False) unless the tick may leave the slots of
    bins that admit no pair unwritten and unchanged records alone ("Records nobody reads")"""
    bs = t.bin_stats()
    if not bs["lazy_last_tick"] or not bs["remembered_slots"]:
        return 1.0, bs["unchanged_records_stay"]
    return bs["written_every_tick"] / bs["remembered_slots"], bs["unchanged_records_stay"]


def algorithmic_bytes_per_entity(child_frac, stages, dirty_frac=1.0, records_frac=1.0, clean_stay=False):
    """Per-launch algorithmic bytes of the dominant kernel k_xform_cull:
    xform 88 + 48*C/N; cull +24 (bounds; the 4*V/N index list is written by the end-of-tick kernel);
    broadphase +32 x records_frac (the AABB record written into its sector bin -- only the share of records that IS written
    is counted: bins that admit no pair are left unwritten, records_written_frac(); bounds are read once for both).
    SURVEY 8d's further 96 B/entity of broadphase traffic (dense AABB array, sort scatter) do not
    exist in this design -- boxes are binned directly -- and are NOT counted anywhere.
    dirty_frac < 1 (config 5): a clean entity re-reads its stored matrix (48 B) instead of 40 B in / 48 B out."""
    b = dirty_frac * (88.0 + 48.0 * child_frac) + (1.0 - dirty_frac) * 48.0
    if "cull" in stages or "broadphase" in stages:
        b += 24.0
    if "broadphase" in stages:
        b += 32.0 * records_frac * (dirty_frac if clean_stay else 1.0)      # (a record that did not change is not rewritten)
    return b


def end_of_tick_bytes(n, roots, visible, sectors, records_read, pairs, stages, producer_kind):
    """Algorithmic bytes of one launch of the end-of-tick kernel (k_compact_pairs): compaction role = visibility words in
    (N/8), ordered list out (4 V), dirty words in and out (N/4); the next frame's producer = link word in (4 N) + the moved
    positions in and out (config 3: x of every root, 8 B; config 5: x, z and six mover words of every entity, see
    DESIGN section 5); pair role = bin counters and layer summaries in and out (16 B per sector), the records of the bins
    that can hold a pair (32 B each) and the pairs out (8 P)."""
    b = 0.0
    if "cull" in stages:
        b += n / 8.0 + 4.0 * visible
    b += n / 4.0
    if producer_kind == 1:
        b += 4.0 * n + 8.0 * roots
    elif producer_kind == 2:
        b += (4.0 + 24.0 + 8.0) * n + 8.0 * roots
    if "broadphase" in stages:
        b += 16.0 * sectors + 32.0 * records_read + 8.0 * pairs
    return b


def pmc_traffic(stages, entities_per_gpu, workload="config3", kernel="k_xform_cull"):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (tools/pmc_session.sh -> profiles/pmc_traffic.json):
    FETCH_SIZE and WRITE_SIZE collected in separate passes and calibrated on known-byte copy kernels of the same
    access widths (FETCH_SIZE under-counts 2x on gfx950).  None when no profile of this exact workload is committed."""
    for path in (os.path.join(ROOT, "profiles", PROFILE_ROUND, f"pmc_traffic_{workload}.json"),
                 os.path.join(ROOT, "profiles", "r03", f"pmc_traffic_{workload}.json"),
                 os.path.join(ROOT, "profiles", "r02", f"pmc_traffic_{workload}.json"),
                 os.path.join(ROOT, "profiles", "pmc_traffic.json")):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        cfg = d.get("bench_config", {})
        if sorted(cfg.get("stages", [])) != sorted(stages) or cfg.get("entities_per_gpu") != entities_per_gpu or cfg.get("workload", "config3") != workload:
            continue
        v = d.get("kernels", {}).get(kernel, {}).get("hbm_bytes_per_launch")
        if v is not None:
            PMC_SOURCES[kernel] = os.path.relpath(path, ROOT)
            return v
    return None


PMC_SOURCES = {}      # kernel -> the committed profile file its `traffic` figure was replayed from (it is NOT measured by this run)


def copy_ceiling_gbs(torch, device):
    """Device copy rate (read + write bytes per second) of a 1 GiB buffer with torch's D2D copy on THIS box (boxes differ:
    4.8-5.5 TB/s seen).  A reference point, NOT the ceiling: a float4-per-thread copy reaches 6.1 TB/s on the same boxes
    and this kernel's own traffic as plain streams 5.7-6.9 TB/s (profiles/r02/micro_bw.log, micro_soa_mix.log)."""
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


def make_world(args, rank, grid):
    from sc_gameengine_amd import synth_world as sw
    tx, tz = grid
    S = args.sectors
    if args.workload == "config5":
        SX, SZ = S // 2, S                       # 128 x 256 sectors x 32 entities = the same 1 048 576 per GPU
        origin = ((rank % tx) * SX, (rank // tx) * SZ)
        w = sw.generate_config5(SX, SZ, origin=origin)
    else:
        SX, SZ = S, S
        origin = ((rank % tx) * S, (rank // tx) * S)
        w = sw.generate(S, S, PROPS, hierarchy=True, origin=origin)
        if args.workload == "config3dyn":        # one prop per sector is a dynamic body: every bin holds an admissible pair partner
            dyn = (np.arange(w.n) % 16) == 4
            w.group[dyn], w.mask[dyn] = sw.GROUP_DYNAMIC, sw.MASK_ALL
    cam = sw.default_camera(float(tx * SX) * 64.0)
    cam["pos"][2] = np.float32(float(tz * SZ) * 64.0 / 2)
    w.camera = cam
    return w, SX, SZ


def host_cpu_model():
    """the host CPU's model name ("print the host CPU model and core count")"""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


class OracleLeg:
    """The checker: liboracle.so on the same world.  Used after the timed region only."""

    def __init__(self, world):
        from oracle import oracle_py as oracle
        oracle.build()
        self.oracle = oracle
        self.world = world
        self.ow = oracle.OracleWorld.from_arrays(world.pos, world.rot, world.scale, world.parent, world.bmin, world.bmax,
                                                 has_mesh=world.has_mesh, has_bounds=world.has_bounds)
        self.has_camera = False
        self.vel = None if world.mover_kind is None else world.mover_vel.copy()

    def produce(self):
        if self.vel is None:
            self.ow.nudge_roots_x(0.01)
        else:
            w = self.world
            self.ow.advance_movers(w.mover_kind, self.vel, w.mover_lo, w.mover_hi, 1.0 / 60.0)

    def parity(self, t, producer_steps, stages, view_proj):
        """Bring the oracle to the frame the GPU ticked last (`producer_steps` producer applications, each exactly the
        device's arithmetic, applied one after the other) and compare.  viewProj is an INPUT of the path on both sides
        (CameraSystem stays on the host): the oracle culls with the matrix the GPU was given."""
        w, ow = self.world, self.ow
        for _ in range(producer_steps):
            self.produce()
        ow.transform_system()
        ow.culling_system(view_proj=view_proj)
        out = {"ticks": producer_steps}
        if "cull" in stages:
            gv, cv = t.visible(), ow.visible()
            out["visible_equal"] = bool(np.array_equal(gv, cv))
            out["visible"] = int(len(cv))
        gm, cm = t.world_matrices(), ow.world_matrices()[:w.n]
        out["matrices_equal"] = bool(np.array_equal(gm, cm))    # IEEE equality, element for element (+0 == -0)
        out["matrices_compared"] = int(w.n)
        if "broadphase" in stages:
            mn, mx = ow.world_aabbs()
            want = self.oracle.broadphase_grid(mn[:w.n], mx[:w.n], w.group, w.mask, 64.0)      # the oracle's search on the oracle's boxes
            got, total = t.pairs()
            key = np.sort(got[:, 0].astype(np.uint64) << np.uint64(32) | got[:, 1].astype(np.uint64))
            wkey = want[:, 0].astype(np.uint64) << np.uint64(32) | want[:, 1].astype(np.uint64)
            out["pairs_equal"] = bool(total == len(want) and np.array_equal(key, wkey))
            out["pairs"] = int(len(want))
        out["ok"] = all(v for k, v in out.items() if k.endswith("_equal"))
        return out

    def baseline(self, ticks=20, warm=3, workers=None):
        """Reference-faithful CPU tick (oracle port), timed on this host: Transform + Camera + Culling,
        hardware_concurrency()-1 workers as the sandbox does (src/sandbox/src/main.cpp:52-54)."""
        hw = os.cpu_count() or 1
        workers = max(hw - 1, 1) if workers is None else workers
        if not self.has_camera:                                 # the timed tick is Transform + Camera + Culling, as the sandbox registers them
            cam = self.world.camera
            self.ow.add_camera_entity(cam["pos"], cam["rot"], aspect=cam["aspect"])
            self.has_camera = True
        self.oracle.lib().orc_jobs_init(workers)
        times = []
        for k in range(warm + ticks):
            self.produce()
            t0 = time.perf_counter()
            self.ow.tick()
            dt = time.perf_counter() - t0
            if k >= warm:
                times.append(dt)
        self.oracle.lib().orc_jobs_init(0)
        med = float(np.median(times))
        n = self.world.n
        return {"value": n / med, "unit": "entities/s", "cores": workers + 1, "kind": "port", "host_cpu": host_cpu_model(),
                "threading": "JobSystem::Dispatch restated: per-worker 1024-slot rings, round-robin enqueue, inline when every ring is full, stealing workers, helping waiter (sc_jobs.cpp:247-372)" if workers else "no job system: groups run in order on the caller",
                "sample": f"same world ({n} entities), {warm} warm-up + {ticks} timed ticks, "
                          f"{'all roots nudged' if self.vel is None else 'movers advanced'} each tick, "
                          f"median tick {med * 1e3:.1f} ms (xform+camera+cull), host cpus {hw}"}

    def close(self):
        self.ow.close()


def tile_pair_gate(args, rank, grid, producer_steps, got_pairs, got_total, mutate=None):
    """The pair gate of ONE rank of a tiled world.  SynthWorld is deterministic per sector, so a rank
    can rebuild, on its own, its tile PLUS the one-sector ring of its neighbours' sectors around it (clipped to the world),
    bring that world to the frame the GPU ticked last with the oracle, run the oracle's grid search on the oracle's boxes,
    and keep the pairs this rank has to report: those whose intersection's low corner lies in a sector it owns (the rule
    the tiles use).  Ids are rank << 24 | dense index on both sides.  `got_pairs`: what the GPU listed.
    mutate (tests only): applied to the rebuilt world before anything else -- a deterministic edit keyed on sector coordinates."""
    from oracle import oracle_py as oracle
    from sc_gameengine_amd import synth_world as sw
    oracle.build()
    tx, tz = grid
    S = args.sectors
    SX, SZ = (S // 2, S) if args.workload == "config5" else (S, S)
    ox, oz = (rank % tx) * SX, (rank // tx) * SZ
    x0, x1 = max(ox - 1, 0), min(ox + SX + 1, tx * SX)
    z0, z1 = max(oz - 1, 0), min(oz + SZ + 1, tz * SZ)
    ex, ez = x1 - x0, z1 - z0
    if args.workload == "config5":
        w = sw.generate_config5(ex, ez, origin=(x0, z0))
    else:
        w = sw.generate(ex, ez, PROPS, hierarchy=True, origin=(x0, z0))
        if args.workload == "config3dyn":
            per16 = (np.arange(w.n) % 16) == 4
            w.group[per16], w.mask[per16] = sw.GROUP_DYNAMIC, sw.MASK_ALL
    if mutate is not None:
        mutate(w)
    per = w.n // (ex * ez)
    ow = oracle.OracleWorld.from_arrays(w.pos, w.rot, w.scale, w.parent, w.bmin, w.bmax, has_mesh=w.has_mesh, has_bounds=w.has_bounds)
    vel = None if w.mover_kind is None else w.mover_vel.copy()
    for _ in range(producer_steps):
        if vel is None:
            ow.nudge_roots_x(0.01)
        else:
            ow.advance_movers(w.mover_kind, vel, w.mover_lo, w.mover_hi, 1.0 / 60.0)
    ow.transform_system()
    mn, mx = ow.world_aabbs()
    mn, mx = mn[:w.n], mx[:w.n]
    want = oracle.broadphase_grid(mn, mx, w.group, w.mask, 64.0)
    ow.close()
    # who reports a pair: the tile that owns the sector holding the low corner of the two boxes' intersection
    inv = np.float32(1.0) / np.float32(64.0)
    lx = np.maximum(mn[want[:, 0], 0], mn[want[:, 1], 0]); lz = np.maximum(mn[want[:, 0], 2], mn[want[:, 1], 2])
    sx = np.clip(np.floor(lx * inv).astype(np.int64), 0, tx * SX - 1); sz = np.clip(np.floor(lz * inv).astype(np.int64), 0, tz * SZ - 1)
    mine = ((sz // SZ) * tx + (sx // SX)) == rank
    # extended-world index -> rank << 24 | dense index inside that rank's tile (sectors row-major inside a tile, `per` entities each)
    e = np.arange(w.n, dtype=np.int64)
    gx, gz = w.sector_of[:, 0].astype(np.int64), w.sector_of[:, 1].astype(np.int64)
    owner = (gz // SZ) * tx + (gx // SX)
    dense = ((gz % SZ) * SX + (gx % SX)) * per + (e % per)
    gid = (owner << 24) | dense
    a, b = gid[want[mine, 0]], gid[want[mine, 1]]
    wkey = np.sort(np.minimum(a, b).astype(np.uint64) << np.uint64(32) | np.maximum(a, b).astype(np.uint64))
    gkey = np.sort(got_pairs[:, 0].astype(np.uint64) << np.uint64(32) | got_pairs[:, 1].astype(np.uint64))
    crossing = int(((owner[want[mine, 0]] != rank) | (owner[want[mine, 1]] != rank)).sum())
    return {"pairs_equal": bool(got_total == len(wkey) and np.array_equal(gkey, wkey)), "pairs": int(len(wkey)),
            "pairs_with_a_neighbours_box": crossing, "ring_world_entities": int(w.n)}


def end_of_tick_line(w, workload, SX, SZ, counts, kp_ms, stages, kind):
    """achieved rate of the end-of-tick kernel from its algorithmic bytes (end_of_tick_bytes) and its dispatch timestamps"""
    roots = int((w.parent < 0).sum())
    sectors = (SX + 2) * (SZ + 2)
    # records the pair role has to read: every record of a bin that holds an admissible pair partner (config 3: none)
    per_sector = 32 if workload == "config5" else 16
    records_read = 0 if workload == "config3" else int(round((per_sector + 3.0) / per_sector * w.n))   # + the ground slab's three extra copies
    eot_ms = float(np.mean(kp_ms)) if len(kp_ms) else None
    eot_bytes = end_of_tick_bytes(w.n, roots if kind == 1 else int((w.mover_kind > 0).sum()), int(counts.visible), sectors,
                                  records_read, int(counts.pairs), stages, kind)
    eot_achieved = (eot_bytes / (eot_ms * 1e-3) / 1e9) if eot_ms else None
    return eot_ms, eot_bytes, eot_achieved


DEVICE_WAKE_MS = float(os.environ.get("SC_BENCH_WAKE_MS", "20"))


def device_wake(torch, device_index):
    """Neutral device activity (fills of a 64 MB buffer, DEVICE_WAKE_MS long) right ahead of the W warm-up steps.  A run of W + K = 25 steps
    lasts a millisecond and starts on a device that idled for seconds while the world was built on the CPU: its first steps run at idle
    clocks (first fused kernel 30.7-31.4 us against 28.5 in steady state; 20 steps 40.9-42.4 us per step without this, 40.1-40.5 with it,
    200 steps unchanged).  Not a step, not inside the timed region, and stated in the line (config.device_wake_ms); SC_BENCH_WAKE_MS=0 turns it off."""
    if DEVICE_WAKE_MS <= 0:
        return
    buf = torch.empty(64 << 20, dtype=torch.uint8, device=f"cuda:{device_index}")
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < DEVICE_WAKE_MS:
        buf.zero_()
        torch.cuda.synchronize()
    del buf


def secondary_leg(args, workload, device, steps=20, warmup=5):
    """The workloads whose pair search really searches, under the driver's clock: the same tick on a
    config3dyn / config 5 world of the same size -- `steps` timed steps bracketed by synchronisation, then as many with every
    launch timed by its dispatch timestamps, then the parity gate INCLUDING the pair set.  One GPU only."""
    import torch
    from sc_gameengine_amd import capi
    from sc_gameengine_amd.tick import WorldTick, camera_view_proj
    a2 = argparse.Namespace(**vars(args))
    a2.workload = workload
    w, SX, SZ = make_world(a2, 0, (1, 1))
    stages = ["xform", "cull", "broadphase"]
    t = WorldTick.from_world(w, device=device, broadphase=True)
    vp = camera_view_proj(w.camera)
    t.set_view_proj(vp)
    kind, param = (2, 1.0 / 60.0) if workload == "config5" else (1, 0.01)
    t.set_frame_producer(kind, param)
    (t.advance_movers if kind == 2 else t.nudge_roots_x)(param)
    flags = capi.FULL | capi.PRODUCE_NEXT
    ticks = 0
    device_wake(torch, device)
    for _ in range(warmup):
        t.run(flags); ticks += 1
    t.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        t.run(flags); ticks += 1
    t.sync(); torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    t.set_profiling(1)
    for _ in range(steps):
        t.run(flags); ticks += 1
    t.sync()
    k1 = t.kernel_times_ms(capi.K_XFORM_CULL)
    kp = t.kernel_times_ms(capi.K_PAIRS)
    t.set_profiling(0)
    counts = t.counts()
    rec_frac, clean_stay = records_written_frac(t) if "broadphase" in stages else (1.0, False)
    leg = OracleLeg(w)
    parity = leg.parity(t, ticks, stages, vp)
    leg.close()
    eot_ms, eot_bytes, eot_achieved = end_of_tick_line(w, workload, SX, SZ, counts, kp, stages, kind)
    dirty_frac = 0.5 if workload == "config5" else 1.0
    bpe = algorithmic_bytes_per_entity(0.0 if workload == "config5" else float((w.parent >= 0).mean()), stages, dirty_frac, rec_frac, clean_stay)
    k1_ms = float(np.mean(k1))
    out = {"workload": workload, "entities": int(w.n), "steps": steps, "warmup": warmup,
           "ms_per_step": elapsed / steps * 1e3, "value": w.n * steps / elapsed, "unit": "entities/s",
           "pairs": int(counts.pairs), "visible": int(counts.visible), "border_lost": int(counts.border_lost),
           "k_xform_cull": {"avg_launch_ms": k1_ms, "bytes_per_entity": bpe, "records_written_frac": rec_frac, "frac": w.n * bpe / (k1_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "traffic": pmc_traffic(stages, w.n, workload)},
           "end_of_tick_kernel": {"avg_launch_ms": eot_ms, "algorithmic_bytes": eot_bytes, "achieved": eot_achieved,
                                  "frac": (eot_achieved / HBM_PEAK_GBS) if eot_achieved else None,
                                  "traffic": pmc_traffic(stages, w.n, workload, "k_compact_pairs")},
           "parity_in_run": parity}
    t.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--sectors", type=int, default=TILE_SECTORS, help="tile side in sectors (default 256 = 1M entities per GPU)")
    ap.add_argument("--stages", default="auto", help="comma list of xform,cull,broadphase (auto = all that are built)")
    ap.add_argument("--graph", type=int, default=0, help="replay the frame from a hipGraph")
    ap.add_argument("--workload", default="config3", choices=["config3", "config3dyn", "config5"],
                    help="config3: 16 static entities per sector, every root nudged each step (the metric's config); "
                         "config3dyn: the same with one prop per sector a dynamic body (every bin holds pair work); "
                         "config5: 16 static + 12 vehicles + 4 peds per sector, agents advanced on device each step")
    ap.add_argument("--sample", type=int, default=0, help="record HIP events on every n-th step (0 = every step up to 64 steps, else every 8th)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the in-run parity gate (profiling sessions only; the line then says so)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the config3dyn / config 5 legs of the default line (N = 1, config 3)")
    ap.add_argument("--profile-run", action="store_true",
                    help="rocprofv3 sessions: warm-up + the timed steps and nothing else -- no event timing, no every-launch pass, no end-to-end "
                         "pass, no parity, no CPU leg -- so the profiler's per-kernel averages ARE the timed pass (profiles/README.md)")
    ap.add_argument("--pipeline", type=int, default=1, help="N>1: pair-search half of tick t on a second stream under the fused kernel of tick t+1")
    ap.add_argument("--control", default="nccl", help="torch.distributed backend of the CONTROL plane at N>1 (rendezvous of the RCCL id, barrier, "
                                                       "max-over-ranks timing); the data path is the library's own RCCL communicator either way")
    ap.add_argument("--same-device", action="store_true", help="rehearsal on one GPU: every rank uses GPU 0 and the messages move by host staging (no RCCL)")
    args = ap.parse_args()
    if args.profile_run:
        args.no_parity = args.no_cpu_baseline = args.no_secondary = True

    import torch
    import torch.distributed as dist
    from sc_gameengine_amd import capi, tiles
    from sc_gameengine_amd.tick import WorldTick, camera_view_proj

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world_size != args.gpus:
        sys.exit(f"--gpus {args.gpus} but WORLD_SIZE={world_size}: launch with python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if args.same_device:
        local_rank = 0
        args.control = "gloo"          # several ranks on one GPU cannot form an RCCL communicator
    torch.cuda.set_device(local_rank)
    if world_size > 1:
        # stdout carries ONE JSON line; some backends announce themselves there ("[Gloo] Rank 0 is connected ..."): keep the
        # descriptor pointed at stderr while the process group comes up
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            if args.control == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group(args.control)
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
    ctl_device = "cuda" if (world_size > 1 and args.control == "nccl") else "cpu"

    grid = tiles.tile_grid(world_size)
    tx, tz = grid
    w, SX, SZ = make_world(args, rank, grid)
    cam = w.camera

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
    view_proj = camera_view_proj(cam)
    t.set_view_proj(view_proj)
    # N > 1: the broadphase's border boxes are the one exchange on the path.  The library owns it: its own RCCL communicator
    # (ncclCommInitRank from the id rank 0 made), the message buffers of both tick parities, and -- pipelined -- a second
    # stream on which exchange, merge and pair search of tick t run under the fused kernel of tick t+1.  A step is ONE call
    # (scTickTileStep).  Any failure here raises: a line is never printed with a stage dropped.
    exchange = "none"
    borders = None
    if world_size > 1 and (flags & capi.BROADPHASE):
        if args.same_device:
            # rehearsal without RCCL (several ranks on one GPU cannot form a communicator): host-staged point-to-point over the
            # control plane; exercises the multi-rank bench flow only
            borders = tiles.BorderBuffers(t, rank, grid, torch.device("cuda", local_rank), pipelined=False)
            exchange = "REHEARSAL: host-staged point-to-point over the control plane (no RCCL)"
        else:
            uid = tiles.rendezvous_unique_id(rank, capi.comm_unique_id)
            if args.graph:
                args.pipeline = 0      # a captured step is one in-order graph per tick parity (successive launches of a graph cannot overlap)
            tiles.setup_tile(t, rank, grid, uid, pipelined=bool(args.pipeline))
            # the tiled world's layer vocabulary: every tile of a SynthWorld configuration is generated by the same rules, so the
            # OR over this tile's colliders IS the world's (config 3: static props only; config 5 / config3dyn: dynamic bodies too,
            # and then every bin is written as before) -- what lets a pipelined tile leave never-needed bins unwritten
            t.set_world_layers(w.group, w.mask)
            exchange = ("border AABBs to <=8 neighbour tiles per step: one group of ncclSend/ncclRecv issued by libsc_tick.so on its own "
                        "RCCL communicator" + (", on the pairs stream under the next tick's fused kernel" if args.pipeline else ""))

    t.set_graph_mode(bool(args.graph))

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
    ticks_done = [0]

    def step():
        ticks_done[0] += 1
        if borders is not None:
            t.run(flags | capi.SPLIT_PAIRS)
            t.sync()
            borders.exchange()
            t.run_pairs()
        else:
            t.tile_step(flags)                    # N == 1: a plain scTickRun; N > 1: tick + pack, RCCL group, merge + pair search

    def fence():
        t.sync()
        torch.cuda.synchronize()
        if world_size > 1:
            dist.barrier()
            torch.cuda.synchronize()

    device_wake(torch, local_rank)
    for _ in range(args.warmup):
        step()
    fence()
    # Kernel durations come from events that take the dispatch's own begin / end timestamps.  Timing a launch that way costs
    # about 12 us of extra gap per step (measured: 60.8 against 51.5 us per step with every step timed), so inside the timed
    # region only every n-th step is timed (default: every 8th, two launches in all for runs of 32 steps or fewer), and right after it -- same process,
    # same resident world, the next frames -- a second pass of the same length times EVERY launch.  Both averages are
    # reported; the roofline uses the every-launch pass.
    sample = args.sample if args.sample > 0 else (8 if args.steps > 32 else max(args.steps // 2, 1))
    # (inside the timed region only the DOMINANT kernel is timed: every timed launch costs the queue ~6 us, and the end-of-tick kernel's
    #  figure comes from the every-launch pass below anyway; its in-region cross-check went with round 4)
    t.set_profiling_kernels([capi.K_XFORM_CULL])
    t.set_profiling(0 if args.profile_run else sample)
    learn_before = t.learn_ticks() if (flags & capi.BROADPHASE) else 0       # (a host-side counter: nothing between the warm-up's fence and the timed region idles the device)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    learn_in_region = (t.learn_ticks() - learn_before) if (flags & capi.BROADPHASE) else 0
    k1_region = t.kernel_times_ms(capi.K_XFORM_CULL)
    kp_region = t.kernel_times_ms(capi.K_PAIRS)
    t.set_profiling_kernels(None)
    t.set_profiling(0 if args.profile_run else 1)
    for _ in range(0 if args.profile_run else min(args.steps, 64)):
        step()
    fence()
    k1 = t.kernel_times_ms(capi.K_XFORM_CULL)
    k2 = t.kernel_times_ms(capi.K_COMPACT)
    kn = t.kernel_times_ms(capi.K_NUDGE)
    kp = t.kernel_times_ms(capi.K_PAIRS)
    t.set_profiling(0)
    counts = t.counts()
    rec_frac, clean_stay = records_written_frac(t) if "broadphase" in stages else (1.0, False)

    # End to end, as the engine would run it in resident mode: every frame also emits the draw items (the sandbox's budget,
    # 6000: src/sandbox/src/main.cpp:96) and reads back counts + visible list + draw items -- staged by a small kernel, ONE
    # device-to-host copy per frame on a copy stream under the next tick -- and the host consumes frame t-1 while tick t runs.
    end_to_end = None
    if world_size == 1 and not args.graph and not args.profile_run:
        t.set_draw_budget(6000)
        t.set_frame_readback(8192, 6000)
        e2e_flags = flags | capi.DRAWS
        check = 0

        def e2e_step(k):
            nonlocal check
            ticks_done[0] += 1
            t.run(e2e_flags)
            if k:
                fr, vis, draws = t.acquire_frame(frames_back=1, copy=False)
                check += int(vis[-1]) + int(fr.draws_in_buffer) if fr.visible_in_buffer else 0     # the host really touches the frame
        # (this leg is at least 200 steps long whatever --steps says: the first frames' acquire -- the host takes frame t-1 while the copy
        #  stream has barely started -- dominates a 20-step run: 53 us per step over 20 steps, 47 over 200 on the same box)
        e2e_steps = max(args.steps, 200)
        for k in range(max(args.warmup, 10)):
            e2e_step(k)
        fence()
        t0 = time.perf_counter()
        for k in range(e2e_steps):
            e2e_step(k + 1)
        fr, vis, draws = t.acquire_frame(frames_back=0, copy=False)
        e2e_elapsed = time.perf_counter() - t0
        fence()
        end_to_end = {"ms_per_step": e2e_elapsed / e2e_steps * 1e3, "steps": e2e_steps,
                      "added_us_per_step": (e2e_elapsed / e2e_steps - elapsed / args.steps) * 1e6,
                      "visible_read_back": int(fr.visible_in_buffer), "draws_read_back": int(fr.draws_in_buffer),
                      "bytes_per_frame": 64 + 8192 * 4 + 6000 * 80,
                      "what": "tick with draw emission (budget 6000) and the frame block written by the end-of-tick kernel's compaction role + one pinned D2H per frame on a copy stream; the host takes frame t-1 while tick t runs"}
        t.set_frame_readback(0, 0)

    own_elapsed = elapsed
    if world_size > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=ctl_device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- parity gate: every rank checks its own tile against the oracle on the same frame ----
    leg = None
    parity = {"skipped": "--no-parity"}
    if not args.no_parity:
        leg = OracleLeg(w)
        parity = leg.parity(t, ticks_done[0], [s for s in stages if not (s == "broadphase" and world_size > 1)], view_proj)
        if world_size > 1:
            parity["note"] = ("per rank: visible list and matrices of the rank's own tile; pair set of the rank against the oracle on its tile plus the "
                              "one-sector ring of its neighbours' sectors (pairs whose low corner lies in a sector the rank owns)")
            if "broadphase" in stages:
                got, total = t.pairs()
                gate = tile_pair_gate(args, rank, grid, ticks_done[0], got, total)
                parity.update(gate)
                cnt_now = t.counts()
                parity["border_lost"] = int(cnt_now.border_lost)
                parity["vocabulary_violations"] = int(cnt_now.vocabulary_violations)      # a neighbour's record outside the declared layer vocabulary
                parity["ok"] = bool(parity["ok"] and gate["pairs_equal"] and parity["border_lost"] == 0 and parity["vocabulary_violations"] == 0)
            ok = torch.tensor([1 if parity["ok"] else 0], dtype=torch.int32, device=ctl_device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            parity["all_ranks_ok"] = bool(int(ok.item()))
        if not parity["ok"] or not parity.get("all_ranks_ok", True):
            print(f"PARITY MISMATCH on rank {rank}: {json.dumps(parity)}", file=sys.stderr)
            t.close()
            if world_size > 1:
                dist.destroy_process_group()
            sys.exit(3)

    # ---- result assembly (N > 1): the global visible list = the ranks' lists in rank order, each at the offset its rank gets from
    # the all-gathered counts (the library's own communicator: scTickGatherVisibleCounts; the host's control plane in the one-GPU
    # rehearsal).  Rank 0 checks the assembled list against the ranks' ORACLE lists assembled the same way.
    visible_assembly = None
    if world_size > 1 and "cull" in stages:
        off, total, ids = tiles.global_visible(t, rank, w.n)
        mine_vis = {"rank": rank, "offset": off, "total": total, "ids": ids.tolist(),
                    "oracle_ids": (leg.ow.visible().astype(np.uint64) + np.uint64(rank * w.n)).tolist() if leg is not None else None}
        boxv = [None] * world_size
        dist.all_gather_object(boxv, mine_vis)
        if rank == 0:
            whole = np.full(total, np.uint64(0xFFFFFFFFFFFFFFFF))
            for m in boxv:
                whole[m["offset"]:m["offset"] + len(m["ids"])] = np.asarray(m["ids"], np.uint64)
            visible_assembly = {"visible_total": int(total), "offsets": [int(m["offset"]) for m in boxv],
                                "totals_agree": all(m["total"] == total for m in boxv),
                                "every_slot_filled": bool((whole != np.uint64(0xFFFFFFFFFFFFFFFF)).all())}
            if leg is not None:
                want = np.concatenate([np.asarray(m["oracle_ids"], np.uint64) for m in boxv]) if total else np.zeros(0, np.uint64)
                visible_assembly["equals_oracle_concatenation"] = bool(len(want) == total and np.array_equal(whole, want))
            bad = not (visible_assembly["totals_agree"] and visible_assembly["every_slot_filled"] and visible_assembly.get("equals_oracle_concatenation", True))
            if bad:
                print(f"GLOBAL VISIBLE LIST MISMATCH: {json.dumps(visible_assembly)}", file=sys.stderr)
        okv = torch.tensor([0 if (rank == 0 and bad) else 1], dtype=torch.int32, device=ctl_device)
        dist.all_reduce(okv, op=dist.ReduceOp.MIN)
        if not int(okv.item()):
            t.close()
            dist.destroy_process_group()
            sys.exit(3)

    # ---- per-rank diagnostics (N > 1): what the first scaling curve will have to be read with ----
    per_rank = None
    if world_size > 1:
        ci = t.comm_info() if borders is None else {}
        mine = {"rank": rank, "border_lost": int(counts.border_lost), "vocabulary_violations": int(counts.vocabulary_violations),
                "pairs": int(counts.pairs), "visible": int(counts.visible),
                "ms_per_step_own_clock": own_elapsed / args.steps * 1e3,
                "tick_chain_us": {"k_xform_cull": float(np.mean(k1)) * 1e3 if len(k1) else None,
                                  "compaction_and_pack": float(np.mean(k2)) * 1e3 if len(k2) else None},
                "pair_chain_us": float(np.mean(kp)) * 1e3 if len(kp) else None,
                "comm": ci}
        box = [None] * world_size
        dist.all_gather_object(box, mine)
        per_rank = box

    if rank == 0:
        n_total = w.n * world_size
        child_frac = float((w.parent >= 0).mean())
        roots = int((w.parent < 0).sum())
        dirty_frac = 0.5 if args.workload == "config5" else 1.0      # config 5: only the movers (half the world, all roots) are rebuilt
        bpe = algorithmic_bytes_per_entity(0.0 if args.workload == "config5" else child_frac, stages, dirty_frac, rec_frac, clean_stay)
        bpe_all = algorithmic_bytes_per_entity(0.0 if args.workload == "config5" else child_frac, stages, dirty_frac)
        k1_ms = float(np.mean(k1)) if len(k1) else None
        achieved = (w.n * bpe) / (k1_ms * 1e-3) / 1e9 if len(k1) else None
        eot_ms, eot_bytes, eot_achieved = end_of_tick_line(w, args.workload, SX, SZ, counts, kp, stages, kind)
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
                "workload": {
                    "config3": f"SynthWorld v1 config 3 per GPU: {SX}x{SZ} sectors x (15 props + ground) = {w.n} entities, depths 0/1/2, every root "
                               f"nudged +0.01 m in x and marked dirty each step; all bodies static, so the pair pass is filter-skipped (pairs: 0)",
                    "config3dyn": f"SynthWorld v1 config 3 per GPU with one prop per sector a dynamic body: {SX}x{SZ} sectors x 16 = {w.n} entities, "
                                  f"every root nudged each step; every bin holds pair work",
                    "config5": f"SynthWorld v1 config 5 per GPU: {SX}x{SZ} sectors x (ground + 15 props + 12 vehicles + 4 peds) = {w.n} "
                               f"entities, vehicles and peds advanced on device each step (dt 1/60), props static",
                }[args.workload],
                "stages": stages,
                "producer": "per step, fused into the end-of-tick kernel as the next frame's producer",
                "tiles": f"{tx}x{tz}",
                "entities_total": n_total,
                "visible": int(counts.visible),
                "pairs": int(counts.pairs),
                "graph": bool(args.graph),
                "pipelined": bool(world_size > 1 and args.pipeline and borders is None),
                "exchange": exchange,
                "resident": "device SoA authoritative; no per-step host transfer",
                "control_plane": args.control if world_size > 1 else None,
                "rehearsal_same_device": bool(args.same_device),
                # neutral device activity right ahead of the W warm-up steps, in ms (device_wake(): the device idled while the world was built)
                "device_wake_ms": DEVICE_WAKE_MS,
                # the bins' remembered slots are relearned every `period` broadphase ticks (SC_TICK_HOME_PERIOD, default 64): that tick
                # reserves every slot with atomics again and runs two or three small kernels behind the fused one (slot counts, the
                # bins' cast-first order where the pair search takes fast sectors, the slots' flags)
                "learn_tick": {"period": int(os.environ.get("SC_TICK_HOME_PERIOD", "64")), "learn_ticks_inside_the_timed_region": int(learn_in_region),
                               "timed_steps": args.steps,
                               "cost": "at 1M entities a learn tick takes ~24 us more than an ordinary one on a world that cannot pair (the fused kernel's learn "
                                       "instance +10, k_snapshot_home 5, k_home_flags 9) and ~45 us more on a searching world (+ k_order_home 17-20): the "
                                       "k_xform_cull<..., 1u> / k_snapshot_home / k_order_home / k_home_flags rows of profiles/r04/<workload>_kernel_stats.csv; "
                                       "0.4-0.7 us per step amortised, inside the timed region only when learn_ticks_inside_the_timed_region > 0"},
                # the host's launch-shape hint (scTickGetBinStats bit 2): no two layer words of this world admit a pair, so the pair role of
                # the end-of-tick kernel is sized as a sweep over the bins' counters (256 workgroups instead of 1041); the search is unchanged
                "pair_role_sweep_only": bool(t.bin_stats()["pair_role_sweep_only"]) if (flags & capi.BROADPHASE) else None,
            },
            "parity_in_run": parity,
            "visible_assembly": visible_assembly,
            "per_rank": per_rank,
            "end_to_end": end_to_end,
            "roofline": {
                "bound": "hbm", "kernel": "k_xform_cull",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                "traffic": pmc_traffic(stages, w.n, args.workload),
                "traffic_source": (f"replayed from {PMC_SOURCES['k_xform_cull']} (rocprofv3 --pmc passes of an earlier session; not a measurement of this run)"
                                   if "k_xform_cull" in PMC_SOURCES else None),
                "bytes_per_entity": bpe, "records_written_frac": rec_frac, "avg_launch_ms": k1_ms, "launches_timed": int(len(k1)),
                # the same launch priced at the bytes of a kernel that writes every record every tick (round 2's kernel; SURVEY 8d's
                # per-entity figure for this design): what the time would be worth had the work not been removed
                "bytes_per_entity_every_record": bpe_all,
                "frac_at_every_record_bytes": ((w.n * bpe_all) / (k1_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if len(k1) else None,
                "timed_where": "every launch of the steps that follow the timed region directly (timing a launch costs ~6 us of gap on its queue, ~12 us per step with both kernels timed; inside the timed region only this kernel is sampled)",
                "avg_launch_ms_in_timed_region": float(np.mean(k1_region)) if len(k1_region) else None,
                "launches_timed_in_timed_region": int(len(k1_region)),
                "end_of_tick_kernel": {
                    "kernel": "k_compact_pairs (compaction + dirty clear + next frame's producer + pair search, one launch)" if not len(k2) else "k_pairs",
                    "avg_launch_ms": eot_ms, "launches_timed": int(len(kp)), "algorithmic_bytes": eot_bytes,
                    "avg_launch_ms_in_timed_region": float(np.mean(kp_region)) if len(kp_region) else None,
                    "achieved": eot_achieved, "frac": (eot_achieved / HBM_PEAK_GBS) if eot_achieved else None,
                    "traffic": pmc_traffic(stages, w.n, args.workload, "k_compact_pairs"),
                    "traffic_source": (f"replayed from {PMC_SOURCES['k_compact_pairs']}" if "k_compact_pairs" in PMC_SOURCES else None),
                    "timing": "begin / end timestamps of the dispatch itself (hipExtLaunchKernelGGL events), as for the fused kernel",
                },
                "other_kernels_ms": {"k_compact": float(np.mean(k2)) if len(k2) else None,
                                     "frame producer": float(np.mean(kn)) if len(kn) else "fused into the end-of-tick kernel (SC_TICK_PRODUCE_NEXT)"},
            },
        }
        if world_size == 1:
            ceiling = copy_ceiling_gbs(torch, torch.device("cuda", local_rank))
            out["roofline"]["torch_copy_rate"] = ceiling
            out["roofline"]["torch_copy_rate_note"] = "torch D2D copy of 1 GiB on this box: a reference point, not the ceiling (DESIGN 5: plain streams of this kernel's traffic run at 5.7-6.9 TB/s)"
            # resident mode's per-frame read-back: the visible list and the matrices of the visible entities
            for _ in range(2):                      # the first call allocates scratch; report the warm one
                t0 = time.perf_counter()
                vis = t.visible()
                mats = t.world_matrices_indexed(vis[:65536])
                out["config"]["readback_ms_visible_list_and_matrices"] = (time.perf_counter() - t0) * 1e3
            del mats
        if world_size == 1 and not args.no_cpu_baseline:
            if leg is None:
                leg = OracleLeg(w)
            out["cpu_baseline"] = leg.baseline()
            one = leg.baseline(ticks=5, warm=1, workers=0)
            out["cpu_baseline"]["single_thread_value"] = one["value"]
            out["cpu_baseline"]["single_thread_sample"] = one["sample"]
        if world_size == 1 and args.workload == "config3" and args.sectors == TILE_SECTORS and not args.no_secondary and not args.graph:
            # the pair search under the same clock: the headline world's pair pass is filter-skipped (all bodies static)
            sec = []
            for wl in ("config3dyn", "config5"):
                leg2 = secondary_leg(args, wl, local_rank)
                if not leg2["parity_in_run"]["ok"]:
                    print(f"PARITY MISMATCH in the secondary leg {wl}: {json.dumps(leg2['parity_in_run'])}", file=sys.stderr)
                    t.close()
                    sys.exit(3)
                sec.append(leg2)
            out["secondary"] = sec
        print(json.dumps(out))
    if leg is not None:
        leg.close()
    t.close()
    if world_size > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
