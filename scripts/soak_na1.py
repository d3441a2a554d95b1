#!/usr/bin/env python3
"""Stability soak of the oxNA step kernel: 256 replicas of the DNA-RNA golden helix (4 096 nt), energy trace every
2 000 steps: potential per nucleotide, kinetic temperature, H-bond energy."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from mythos_amd.energy import na1  # noqa: E402
from mythos_amd.energy.base import Quaternion, RigidBody, space  # noqa: E402
from mythos_amd.simulators.hip_md import HipMDSimulator, StaticSimulatorParams, nvt_langevin  # noqa: E402
from mythos_amd.simulators.neighbors import VerletNeighborList  # noqa: E402
from scripts import _golden  # noqa: E402

KT = 296.15 * 0.1 / 300.0
n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
reps = 256
top, traj, _ = _golden.load("na1", "simple-helix-dna-rna", new_format=True)
disp, shift = space.free()
ef = na1.create_default_energy_fn(top, disp)
params = StaticSimulatorParams(seq=top.seq, mass=(1.0, (1.0, 1.0, 1.0)), gamma=(KT / 2.5, KT / 7.5), bonded_neighbors=top.bonded_neighbors,
                               checkpoint_every=0, dt=3e-3, kT=KT)
state = {"init_state": RigidBody(center=torch.as_tensor(traj.center[0], dtype=torch.float32),
                                 orientation=Quaternion(vec=torch.as_tensor(traj.quaternions[0], dtype=torch.float32))), "key": 1}
sim = HipMDSimulator(energy_fn=ef, simulator_params=params, space=(disp, shift), simulator_init=nvt_langevin,
                     neighbors=VerletNeighborList(3.25, 0.6, 20), save_every=2000, dtype=torch.float32, n_replicas=reps)
n = top.n_nucleotides
for blk in range(n_steps // 2000):
    out = sim.run({}, state["init_state"], 2000, key=state["key"])
    state = {"init_state": out.state["init_state"], "key": out.state["key"]}
    tr = out.observables[0]  # one saved state per replica
    terms = ef.compute_terms(RigidBody(center=tr.center.double(), orientation=Quaternion(vec=tr.orientation.vec.double()))).cpu().numpy() / n
    p, L = out.state["momentum"]
    t_kin = float((p.double() ** 2).sum() + (L.double() ** 2).sum()) / (6 * n * reps * KT)
    assert np.isfinite(terms).all()
    print(f"step {2000 * (blk + 1):6d}  U/N {terms.sum(1).mean():8.4f}  T_kin/T {t_kin:6.4f}  HB/N {terms[:, 4].mean():7.4f}  stacking/N {terms[:, 2].mean():7.4f}", flush=True)
print("soak ok")
