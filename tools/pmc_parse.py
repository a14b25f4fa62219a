#!/usr/bin/env python3
"""Turn the rocprofv3 --pmc CSVs of tools/pmc_session.sh into per-launch HBM traffic per kernel.

Calibration: the copy kernels move a known byte count; the ratio counter/bytes gives the unit (and
the gfx950 FETCH_SIZE under-count for each access width).  Tick kernels are then priced with the
calibration of the kernel whose access shape they share (soa13 for k_xform_cull)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def load(dirname, counter):
    rows = defaultdict(list)
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter:
                rows[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return rows


def short(name):
    for k in ("k_xform_cull", "k_compact_pairs", "k_compact_pack", "k_pairs", "k_compact", "k_emit_draws_staged", "k_nudge_roots_x", "calib_copy_dword", "calib_copy_float4", "calib_soa13_to_rows"):
        if k in name:
            return k
    return None


def main(out):
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (tools/pmc_session.sh)", "calibration": {}, "kernels": {}}
    known = {"calib_copy_dword": (1 << 30, 1 << 30), "calib_copy_float4": (1 << 30, 1 << 30), "calib_soa13_to_rows": ((16 << 20) * 52, (16 << 20) * 48)}
    for ci, counter in enumerate(("FETCH_SIZE", "WRITE_SIZE")):
        for kname, vals in load(os.path.join(out, f"calib_{counter}"), counter).items():
            s = short(kname)
            if s in known:
                mean = sum(vals) / len(vals)
                res["calibration"].setdefault(s, {})[counter] = {"counter_mean": mean, "known_bytes": known[s][ci], "bytes_per_count": known[s][ci] / mean}
    for ci, counter in enumerate(("FETCH_SIZE", "WRITE_SIZE")):
        for kname, vals in load(os.path.join(out, f"tick_{counter}"), counter).items():
            s = short(kname)
            if not s or s.startswith("calib"):
                continue
            vals = vals[len(vals) // 4:]                      # drop warm-up launches
            mean = sum(vals) / len(vals)
            shape = "calib_soa13_to_rows" if s == "k_xform_cull" else "calib_copy_float4" if s in ("k_pairs", "k_emit_draws_staged") else "calib_copy_dword"
            unit = res["calibration"].get(shape, {}).get(counter, {}).get("bytes_per_count")
            res["kernels"].setdefault(s, {})[counter] = {"counter_mean": mean, "launches": len(vals), "calibrated_with": shape,
                                                          "bytes_per_launch": mean * unit if unit else None}
    for s, k in res["kernels"].items():
        if all(k.get(c, {}).get("bytes_per_launch") is not None for c in ("FETCH_SIZE", "WRITE_SIZE")):
            k["hbm_bytes_per_launch"] = k["FETCH_SIZE"]["bytes_per_launch"] + k["WRITE_SIZE"]["bytes_per_launch"]
    res["bench_config"] = {"stages": ["xform", "cull", "broadphase"], "entities_per_gpu": 1048576, "workload": os.environ.get("PMC_WORKLOAD", "config3")}
    json.dump(res, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
