#!/usr/bin/env python3
"""How even is the pair role's work over its waves?  Needs a library built with -DSC_DIAG_WAVETIME (make -C sc_gameengine_amd/csrc
OUT=.../wavetime.so EXTRA=-DSC_DIAG_WAVETIME): every pair-role wave notes the s_memtime span of its sweep; per launch this prints the
average and the longest wave against the span from the first wave's start to the last wave's end, and how late the last wave started.
Usage: SC_TICK_LIB=build_ab/wavetime.so python tools/wave_time.py --workload config5"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sc_gameengine_amd import capi, synth_world as sw           # noqa: E402
from sc_gameengine_amd.tick import WorldTick, camera_view_proj   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="config5")
ap.add_argument("--ticks", type=int, default=12)
args = ap.parse_args()
if args.workload == "config5":
    w = sw.generate_config5(128, 256)
    kind, param = 2, 1.0 / 60.0
else:
    w = sw.config("config3")
    if args.workload == "config3dyn":
        dyn = (np.arange(w.n) % 16) == 4
        w.group[dyn], w.mask[dyn] = sw.GROUP_DYNAMIC, sw.MASK_ALL
    kind, param = 1, 0.01
t = WorldTick.from_world(w, broadphase=True)
t.set_view_proj(camera_view_proj(w.camera))
t.set_frame_producer(kind, param)
(t.advance_movers if kind == 2 else t.nudge_roots_x)(param)
flags = capi.FULL | capi.PRODUCE_NEXT
for _ in range(70):
    t.run(flags)
t.sync()
lib = C.CDLL(capi.LIB_PATH)
fn = lib.scTickDiagWaveTime
fn.argtypes = [C.POINTER(C.c_uint64), C.c_int]
out = (C.c_uint64 * 40)()
rows = []
for k in range(args.ticks):
    assert fn(None, 1)
    t.run(flags); t.sync()
    fn(out, 0)
    tot, mx, n, t0min, t1max, t0max = (int(out[i]) for i in range(6))
    span = t1max - t0min
    rows.append({"waves": n, "avg": round(tot / max(n, 1), 1), "longest": mx, "first_start_to_last_end": span, "last_wave_started_after": t0max - t0min,
                 "avg_over_span": round(tot / max(n, 1) / max(span, 1), 3), "longest_over_span": round(mx / max(span, 1), 3),
                 "waves_by_1.28us_bucket": [int(out[8 + b]) for b in range(32)]})
rows_fn = lib.scTickDiagWaveRows
rows_fn.argtypes = [C.POINTER(C.c_uint32), C.c_uint32]
nw = rows[-1]["waves"]
buf = np.zeros((nw, 8), np.uint32)
assert rows_fn(buf.ctypes.data_as(C.POINTER(C.c_uint32)), nw)
dur = buf[:, 0].astype(np.float64)
order = np.argsort(dur)
def desc(ix):
    return [{"wave": int(buf[i, 7]), "ticks": int(buf[i, 0]), "fast_sectors": int(buf[i, 1]), "sweeper_x_records": int(buf[i, 2]), "general_sectors": int(buf[i, 3]),
             "general_records": int(buf[i, 4]), "rounds": int(buf[i, 5])} for i in ix]
corr = {"ticks_vs_sweeper_x_records": round(float(np.corrcoef(dur, buf[:, 2])[0, 1]), 3), "ticks_vs_general_sectors": round(float(np.corrcoef(dur, buf[:, 3])[0, 1]), 3),
        "ticks_vs_wave_index": round(float(np.corrcoef(dur, buf[:, 7])[0, 1]), 3)}
pct = {str(q): float(np.percentile(dur, q)) for q in (5, 25, 50, 75, 95, 99, 100)}
dec = [round(float(dur[(buf[:, 7] * 10 // nw) == k].mean()), 1) for k in range(10)]
t.set_profiling(1)
for _ in range(20):
    t.run(flags)
t.sync()
eot_ms = t.kernel_times_ms(capi.K_PAIRS)
t.set_profiling(0)
by_block_pos = [round(float(dur[(buf[:, 7] % 4) == k].mean()), 1) for k in range(4)]
last = {"end_of_tick_kernel_us_by_events": round(float(np.mean(eot_ms)) * 1e3, 2), "mean_ticks_by_decile_of_wave_index": dec, "percentiles_ticks": pct, "correlations": corr, "mean_ticks_by_wave_of_workgroup": by_block_pos, "fastest": desc(order[:6]), "slowest": desc(order[-10:]),
        "general_sectors_total": int(buf[:, 3].sum()), "fast_sectors_total": int(buf[:, 1].sum())}
print(json.dumps({"workload": args.workload, "last_launch_per_wave": last, "unit": "wall_clock64 ticks (10 ns)", "launches": rows[2:]}, indent=1))
