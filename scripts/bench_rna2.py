#!/usr/bin/env python3
"""oxRNA2 MD throughput: R replicas of the 12 bp RNA helix of the reference's rna2 golden (24 nt each), one launch per
step, with the oxDNA2 golden helix (16 nt) at the same total size beside it.  Prints steps per second of the whole batch."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from mythos_amd.energy import dna2, rna2  # noqa: E402
from mythos_amd.energy.base import Quaternion, RigidBody, space  # noqa: E402
from mythos_amd.simulators.hip_md import HipMDSimulator, StaticSimulatorParams, nvt_langevin  # noqa: E402
from mythos_amd.simulators.neighbors import VerletNeighborList  # noqa: E402
from scripts import _golden  # noqa: E402

KT = 296.15 * 0.1 / 300.0
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
total = int(sys.argv[2]) if len(sys.argv) > 2 else 24000
disp, shift = space.free()
for model, mod, name in ((3, rna2, "simple-helix-12bp"), (2, dna2, "simple-helix")):
    top, traj, _ = _golden.load({2: "dna2", 3: "rna2"}[model], name)
    reps = total // top.n_nucleotides
    ef = mod.create_default_energy_fn(top, disp)
    params = StaticSimulatorParams(seq=top.seq, mass=(1.0, (1.0, 1.0, 1.0)), gamma=(KT / 2.5, KT / 7.5),
                                   bonded_neighbors=top.bonded_neighbors, checkpoint_every=0, dt=3e-3, kT=KT)
    init = RigidBody(center=torch.as_tensor(traj.center[0], dtype=torch.float32), orientation=Quaternion(vec=torch.as_tensor(traj.quaternions[0], dtype=torch.float32)))
    for dtype in (torch.float32, torch.float64):
        sim = HipMDSimulator(energy_fn=ef, simulator_params=params, space=(disp, shift), simulator_init=nvt_langevin,
                             neighbors=VerletNeighborList(3.25, 0.9, 50), save_every=0, dtype=dtype, n_replicas=reps)
        sim.run({}, init, 300, key=1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sim.run({}, init, steps, key=2)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"model {model} {name}: {reps} replicas x {top.n_nucleotides} nt = {reps * top.n_nucleotides} nt, {dtype}: {steps / dt:9.0f} steps/s", flush=True)
