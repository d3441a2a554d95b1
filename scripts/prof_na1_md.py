#!/usr/bin/env python3
"""dev: the oxNA MD loop alone (for rocprofv3 --kernel-trace --stats; MYTHOS_NA1_UNFUSED=1 for the two-launch path): 1 500
replicas of the DNA-RNA golden helix."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from mythos_amd.energy import na1  # noqa: E402
from mythos_amd.energy.base import Quaternion, RigidBody, space  # noqa: E402
from mythos_amd.simulators.hip_md import HipMDSimulator, StaticSimulatorParams, nvt_langevin  # noqa: E402
from mythos_amd.simulators.neighbors import VerletNeighborList  # noqa: E402
from scripts import _golden  # noqa: E402

KT = 296.15 * 0.1 / 300.0
every = int(sys.argv[1]) if len(sys.argv) > 1 else 10
top, traj, _ = _golden.load("na1", "simple-helix-dna-rna", new_format=True)
disp, shift = space.free()
ef = na1.create_default_energy_fn(top, disp)
params = StaticSimulatorParams(seq=top.seq, mass=(1.0, (1.0, 1.0, 1.0)), gamma=(KT / 2.5, KT / 7.5), bonded_neighbors=top.bonded_neighbors,
                               checkpoint_every=0, dt=3e-3, kT=KT)
init = RigidBody(center=torch.as_tensor(traj.center[0], dtype=torch.float32), orientation=Quaternion(vec=torch.as_tensor(traj.quaternions[0], dtype=torch.float32)))
sim = HipMDSimulator(energy_fn=ef, simulator_params=params, space=(disp, shift), simulator_init=nvt_langevin,
                     neighbors=VerletNeighborList(3.25, 0.6, every), save_every=0, dtype=torch.float32, n_replicas=1500)
sim.run({}, init, 100, key=1)
torch.cuda.synchronize()
t0 = time.perf_counter()
sim.run({}, init, 1000, key=2)
torch.cuda.synchronize()
print(f"rebuild every {every}: {1000 / (time.perf_counter() - t0):8.0f} steps/s", flush=True)
