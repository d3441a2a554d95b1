#!/usr/bin/env python3
"""round 4: the DiffTRe iteration's MD (64 replicas x 64 nt in one launch per step) on the reference's all-pairs list against
a device-built Verlet list - step kernel time by events and wall time of 2 000 steps."""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from mythos_amd.energy import dna2
from mythos_amd.energy.base import Quaternion, RigidBody, space
from mythos_amd.simulators.hip_md import HipMDSimulator, StaticSimulatorParams, nvt_langevin
from mythos_amd.simulators.neighbors import NoNeighborList, VerletNeighborList
from mythos_amd.utils import generators

dev = torch.device("cuda", 0)
kT = 296.15 * 0.1 / 300.0
top, c0, q0 = generators.ideal_duplex(32, model=2, seed=21)
disp, shift = space.free()
ef = dna2.create_default_energy_fn(topology=top, displacement_fn=disp)
init = RigidBody(center=torch.as_tensor(c0, device=dev), orientation=Quaternion(vec=torch.as_tensor(q0, device=dev)))
sp = StaticSimulatorParams(seq=top.seq, mass=(1.0, (1.0, 1.0, 1.0)), gamma=(kT / 2.5, kT / 7.5), bonded_neighbors=top.bonded_neighbors,
                           checkpoint_every=0, dt=0.005, kT=kT)
for dtype in (torch.float64, torch.float32):
    for name, nb in (("all pairs", NoNeighborList(unbonded_nbrs=top.unbonded_neighbors)), ("verlet 0.6/25", VerletNeighborList(3.25, 0.6, 25)),
                     ("verlet 0.9/50", VerletNeighborList(3.25, 0.9, 50))):
        sim = HipMDSimulator(energy_fn=ef, simulator_params=sp, space=(disp, shift), simulator_init=nvt_langevin, neighbors=nb, save_every=20,
                             dtype=dtype, n_replicas=64)
        o = sim.run({}, init, 2000, key=1)
        state = o.state["init_state"]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        o = sim.run({}, state, 2000, key=2)
        torch.cuda.synchronize()
        wall = 1e3 * (time.perf_counter() - t0)
        system, integ, _ = next(iter(sim._resident.values()))
        integ.set_timing(16)
        sim.run({}, state, 512, key=3)
        kms = integ.last_kernel_ms()["kernel_ms"]
        print(f"{str(dtype):14s} {name:14s} 2000 steps {wall:7.2f} ms   step kernel {1e3 * kms:6.2f} us   rows mean {system.neighbor_stats()[1]:.1f}")
        sim.release()
