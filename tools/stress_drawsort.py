#!/usr/bin/env python3
"""The renderer's draw order on random worlds, handle ranges, pipeline tables and budgets against the oracle's stable sort
(tests/test_gpu_drawsort.py::check_sorted does the comparing).  python tools/stress_drawsort.py [--seeds 20]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle_py                                   # noqa: E402
from tests import worlds                                       # noqa: E402
from tests.test_gpu_drawsort import check_sorted               # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seeds", type=int, default=20)
args = ap.parse_args()
oracle_py.build()
bad = 0
for seed in range(args.seeds):
    rng = np.random.default_rng(11000 + seed)
    n = int(rng.choice([1, 65, 300, 5000, 30000, 90000]))
    w = worlds.random_world(n, seed=seed, spread=float(rng.choice([40.0, 150.0])), p_child=float(rng.choice([0.0, 0.3])), p_no_mesh=float(rng.choice([0.0, 0.2])))
    nmesh = int(rng.choice([1, 3, 40, 70000, 2 ** 24]))
    nmat = int(rng.choice([1, 6, 300, 65538]))
    w.mesh = rng.integers(0, min(nmesh + 2, 2 ** 24), w.n).astype(np.uint32)
    w.material = rng.integers(0, nmat, w.n).astype(np.uint32)
    pipeline = rng.integers(0, 128, nmat).astype(np.uint8)
    pipeline[rng.random(nmat) < 0.1] = 0xFF
    budget = int(rng.choice([0, 1, 64, 4096, 1 << 20]))
    try:
        check_sorted(oracle_py, w, pipeline, mesh_count=nmesh, max_draws=budget, graph=bool(rng.integers(0, 2)), expect_min=0)
    except AssertionError as e:
        bad += 1
        print(f"seed {seed}: n {n} meshes {nmesh} materials {nmat} budget {budget}: {str(e)[:300]}", flush=True)
print(f"{args.seeds - bad} of {args.seeds} sorted draw lists equal")
sys.exit(1 if bad else 0)
