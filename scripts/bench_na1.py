#!/usr/bin/env python3
"""oxNA energy path: the reference's hybrid goldens (16 nt, 100 frames tiled to 6 400) through the model-4 instantiation -
all-DNA, all-RNA and DNA-RNA duplexes (the last one diverges over the three code paths) - with the all-DNA duplex through
the oxDNA2 instantiation beside them.  ms per call of energies, + forces, + dU/dtheta."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from mythos_amd import _lib  # noqa: E402
from mythos_amd.energy import flat_params as fp  # noqa: E402
from mythos_amd.hip_system import OxdnaSystem  # noqa: E402
from mythos_amd.input import defaults  # noqa: E402
from scripts import _golden  # noqa: E402

sim, cfg = defaults.default_configs_for("na1")
flat4 = fp.pack_flat_na1(fp.derive_flat_na1(cfg["dna"], cfg["rna"], cfg["drh"], kt=sim["kT"], salt_conc=0.5, half_charged_ends=False), _lib.param_names())
flat2 = fp.pack_flat(fp.derive_flat(2, cfg["dna"], kt=sim["kT"], salt_conc=0.5, half_charged_ends=False), _lib.param_names())
for dtype in (torch.float32, torch.float64):
    for name, model in (("simple-helix-dna-dna", 2), ("simple-helix-dna-dna", 4), ("simple-helix-rna-rna", 4), ("simple-helix-dna-rna", 4)):
        top, traj, is_rna = _golden.load("na1", name, new_format=True)
        s = OxdnaSystem(model, top.seq, top.is_end, top.bonded_neighbors, box=traj.box_size, dtype=dtype, is_rna=is_rna if model == 4 else None)
        s.set_params(flat4 if model == 4 else flat2)
        s.set_neighbors(top.unbonded_neighbors)
        c = torch.as_tensor(np.tile(traj.center, (64, 1, 1)), dtype=dtype, device=s.device)
        q = torch.as_tensor(np.tile(traj.quaternions, (64, 1, 1)), dtype=dtype, device=s.device)
        row = []
        for kw in ({}, {"grads": True}, {"grads": True, "param_grads": True}):
            s.energy(c, q, **kw)
            torch.cuda.synchronize()
            ts = []
            for _ in range(15):
                t0 = time.perf_counter()
                s.energy(c, q, **kw)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            row.append(round(1e3 * float(np.median(ts)), 3))
        print(f"{str(dtype).split('.')[-1]:8s} model {model} {name:22s} {c.shape[0]} frames x {top.n_nucleotides} nt: {row} ms", flush=True)

# ---- MD of hybrid systems: the fused step kernel's oxNA instantiation (MYTHOS_NA1_UNFUSED=1: the two-launch path), 1 500
#      replicas of the DNA-RNA golden helix = 24 000 nt in one system, Verlet list rebuilt every 10 steps
from mythos_amd.energy import na1  # noqa: E402
from mythos_amd.energy.base import Quaternion, RigidBody, space  # noqa: E402
from mythos_amd.simulators.hip_md import HipMDSimulator, StaticSimulatorParams, nvt_langevin  # noqa: E402
from mythos_amd.simulators.neighbors import VerletNeighborList  # noqa: E402

KT = 296.15 * 0.1 / 300.0
top, traj, _ = _golden.load("na1", "simple-helix-dna-rna", new_format=True)
disp, shift = space.free()
ef = na1.create_default_energy_fn(top, disp)
params = StaticSimulatorParams(seq=top.seq, mass=(1.0, (1.0, 1.0, 1.0)), gamma=(KT / 2.5, KT / 7.5), bonded_neighbors=top.bonded_neighbors,
                               checkpoint_every=0, dt=3e-3, kT=KT)
init = RigidBody(center=torch.as_tensor(traj.center[0], dtype=torch.float32), orientation=Quaternion(vec=torch.as_tensor(traj.quaternions[0], dtype=torch.float32)))
for dtype in (torch.float32, torch.float64):
    sim = HipMDSimulator(energy_fn=ef, simulator_params=params, space=(disp, shift), simulator_init=nvt_langevin,
                         neighbors=VerletNeighborList(3.25, 0.6, 10), save_every=0, dtype=dtype, n_replicas=1500)
    sim.run({}, init, 100, key=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sim.run({}, init, 1000, key=2)
    torch.cuda.synchronize()
    print(f"MD, 1500 replicas x 16 nt, {str(dtype).split('.')[-1]}: {1000 / (time.perf_counter() - t0):8.0f} steps/s", flush=True)
