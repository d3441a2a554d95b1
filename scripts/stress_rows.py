#!/usr/bin/env python3
"""Stress of the device row builder against a k-d tree (scipy): random clouds, free and periodic, fp32 and fp64, with
the cell buckets at their managed size or squeezed (second argument: places per bucket, through mythos_debug_set).
Compares the number of listed pairs exactly (every pair is in two rows) and the longest row.

    python scripts/stress_rows.py [seed] [bucket_cap]"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from mythos_amd import _lib  # noqa: E402
from mythos_amd.hip_system import OxdnaSystem  # noqa: E402
from mythos_amd.simulators.neighbors import verlet_pairs_numpy  # noqa: E402

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
if len(sys.argv) > 2:
    _lib.debug_set("cell_bucket_cap", int(sys.argv[2]))
bad = 0
for case in range(24):
    n = int(rng.integers(600, 6000))
    periodic = bool(case % 2)
    dtype = torch.float64 if case % 3 == 0 else torch.float32
    r_list = float(rng.uniform(2.0, 4.5))
    per_ball = float(rng.uniform(8, 45))  # mean neighbours inside r_list
    vol = n * (4.0 / 3.0) * np.pi * r_list**3 / per_ball
    if periodic:
        edge = max(vol ** (1.0 / 3.0), 3.05 * r_list)
        box = np.array([edge, edge * 1.1, edge * 0.95])
        c = rng.uniform(0, 1, size=(n, 3)) * box
        if case % 4 == 1:
            c += rng.integers(-2, 3, size=(n, 3)) * box  # unwrapped coordinates
    else:
        box = None
        c = rng.normal(size=(n, 3)) * (vol ** (1.0 / 3.0)) / 2.2  # a blob: dense centre, sparse halo
    strands = [n // 2, n - n // 2]
    bonded = np.array([[i, i + 1] for i in range(strands[0] - 1)] + [[i, i + 1] for i in range(strands[0], n - 1)], dtype=np.int32)
    seq = rng.integers(0, 4, size=n).astype(np.int32)
    is_end = np.zeros(n, dtype=np.int32)
    s = OxdnaSystem(2, seq, is_end, bonded, box=box, dtype=dtype)
    cd = torch.as_tensor(c, dtype=dtype, device=s.device).contiguous()
    s.build_neighbors(cd, r_list, 0.0)
    mx, mean = s.neighbor_stats()
    ref = verlet_pairs_numpy(cd.double().cpu().numpy(), bonded, r_list, box=box)
    counts = np.bincount(ref.reshape(-1), minlength=n)
    ok = round(mean * n) == 2 * len(ref) and mx == counts.max()
    bad += not ok
    print(f"case {case:2d} n {n:5d} {'box ' if periodic else 'free'} {str(dtype)[-7:]} r {r_list:.2f}: pairs {round(mean * n / 2)} "
          f"ref {len(ref)} max row {mx} ref {counts.max()} {'ok' if ok else 'MISMATCH'}", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
