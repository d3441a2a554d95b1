#!/usr/bin/env python3
"""dev: the stand-alone observables launch (mythos_observables_eval) on the DiffTRe shape - 6 400 frames of a 32 bp
duplex - for rocprofv3; prints ms per call."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from mythos_amd.energy.base import Quaternion, space  # noqa: E402
from mythos_amd.input import defaults  # noqa: E402
from mythos_amd.observables import PersistenceLength, PitchAngle, PropellerTwist, Rise, get_duplex_quartets  # noqa: E402
from mythos_amd.simulators.io import SimulatorTrajectory  # noqa: E402
from mythos_amd.utils import generators  # noqa: E402

bp, frames = 32, 6400
top, c, q = generators.ideal_duplex(bp, model=2, seed=7)
rng = np.random.default_rng(0)
C = c[None] + 0.05 * rng.standard_normal((frames, *c.shape))
Q = q[None] + 0.03 * rng.standard_normal((frames, *q.shape))
Q /= np.linalg.norm(Q, axis=-1, keepdims=True)
dev = torch.device("cuda", 0)
_, cfg = defaults.default_configs_for("dna2")
disp = space.free()[0]
quartets = get_duplex_quartets(bp)
pairs = np.stack([np.arange(bp), 2 * bp - 1 - np.arange(bp)], axis=1)[1:-1]
for dtype in (torch.float32, torch.float64):
    traj = SimulatorTrajectory(center=torch.as_tensor(C, dtype=dtype, device=dev), orientation=Quaternion(vec=torch.as_tensor(Q, dtype=dtype, device=dev)))
    obs = [PropellerTwist(pairs), Rise(quartets, disp, cfg["geometry"]), PitchAngle(quartets, disp, cfg["geometry"]),
           PersistenceLength(quartets, disp, cfg["geometry"], truncate=10)]
    for o in obs:
        o(traj)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        for o in obs:
            o(traj)
    torch.cuda.synchronize()
    print(f"{str(dtype).split('.')[-1]}: {1e3 * (time.perf_counter() - t0) / 40:.3f} ms per observable call, {frames} frames x {2 * bp} nt")
