#!/usr/bin/env python3
"""Replica batching (HipMDSimulator.n_replicas): R copies of a 32-bp oxDNA2 duplex (BASELINE configs[4]: 64 replicas)
advanced by one launch per step, against one copy at a time.  Prints replica-steps per second."""
import dataclasses as dc
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from mythos_amd.energy import dna2  # noqa: E402
from mythos_amd.energy.base import Quaternion, RigidBody, space  # noqa: E402
from mythos_amd.simulators.hip_md import HipMDSimulator, StaticSimulatorParams, nvt_langevin  # noqa: E402
from mythos_amd.simulators.neighbors import VerletNeighborList  # noqa: E402
from mythos_amd.utils import generators  # noqa: E402

KT = 296.15 * 0.1 / 300.0
top, c0, q0 = generators.ideal_duplex(32, model=2, seed=7)
disp, shift = space.free()
ef = dna2.create_default_energy_fn(top, disp)
params = StaticSimulatorParams(seq=top.seq, mass=(1.0, (1.0, 1.0, 1.0)), gamma=(KT / 2.5, KT / 7.5),
                               bonded_neighbors=top.bonded_neighbors, checkpoint_every=0, dt=5e-3, kT=KT)
init = RigidBody(center=torch.as_tensor(c0, dtype=torch.float32), orientation=Quaternion(vec=torch.as_tensor(q0, dtype=torch.float32)))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
skin = float(sys.argv[2]) if len(sys.argv) > 2 else 0.9    # (bench.py's policy; 0.6 / 25 was round 2's)
every = int(sys.argv[3]) if len(sys.argv) > 3 else 50
base = HipMDSimulator(energy_fn=ef, simulator_params=params, space=(disp, shift), simulator_init=nvt_langevin,
                      neighbors=VerletNeighborList(3.25, skin, every), save_every=100)
for reps in (1, 8, 64, 256):
    sim = dc.replace(base, n_replicas=reps)
    sim.run({}, init, 200, key=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = sim.run({}, init, steps, key=2)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{reps:4d} replicas x {top.n_nucleotides} nt: {steps / dt:9.0f} steps/s, {reps * steps / dt:11.0f} replica-steps/s, "
          f"{out.observables[0].center.shape[0]} saved states")
