#!/usr/bin/env python3
"""dev (round 4): where the host time of a DiffTRe iteration goes - cProfile over HipMDSimulator.run(0 steps), energy_fn.map
and compute_loss_and_grad on the bench's configs[4] shape; and the cfg1 fp64 energy check by hand."""
import cProfile, pstats, sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from mythos_amd import _lib
from mythos_amd.energy import dna2, flat_params as fp
from mythos_amd.energy.base import Quaternion, RigidBody, space
from mythos_amd.hip_system import OxdnaSystem
from mythos_amd.input import defaults
from mythos_amd.observables import PropellerTwist
from mythos_amd.optimization import objective as O
from mythos_amd.simulators.hip_md import HipMDSimulator, StaticSimulatorParams, nvt_langevin
from mythos_amd.simulators.neighbors import NoNeighborList
from mythos_amd.utils import generators

dev = torch.device("cuda", 0)
kT = 296.15 * 0.1 / 300.0
top, c0, q0 = generators.ideal_duplex(32, model=2, seed=21)
n = top.n_nucleotides
disp, shift = space.free()
ef = dna2.create_default_energy_fn(topology=top, displacement_fn=disp)
init = RigidBody(center=torch.as_tensor(c0, device=dev), orientation=Quaternion(vec=torch.as_tensor(q0, device=dev)))
sp = StaticSimulatorParams(seq=top.seq, mass=(1.0, (1.0, 1.0, 1.0)), gamma=(kT / 2.5, kT / 7.5), bonded_neighbors=top.bonded_neighbors,
                           checkpoint_every=0, dt=0.005, kT=kT)
simr = HipMDSimulator(energy_fn=ef, simulator_params=sp, space=(disp, shift), simulator_init=nvt_langevin,
                      neighbors=NoNeighborList(unbonded_nbrs=top.unbonded_neighbors), save_every=20, dtype=torch.float64, n_replicas=64)
opt = {"eps_stack_base": 1.3523, "eps_hb": 1.0678, "theta0_hb_4": float(np.pi)}
o = simr.run(opt, init, 2000, key=1)
traj = o.observables[0]
state = o.state["init_state"]
torch.cuda.synchronize()
half = n // 2
ptwist = PropellerTwist(np.stack([np.arange(half), n - 1 - np.arange(half)], axis=1)[1:-1])


def loss_fn(ref_states, weights, energy_fn, opt_params, observables):
    m = (weights * ptwist(ref_states).to(weights.dtype)).sum()
    return (m - 21.7) ** 2, (("propeller_twist", m.detach()), {})


def block(name, fn, reps=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    print(f"== {name}: {1e3 * (time.perf_counter() - t0) / reps:.3f} ms per call")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(22)


k = [10]
def run0():
    k[0] += 1
    simr.run(opt, state, 0, key=k[0])
block("HipMDSimulator.run(0 steps)", run0)
with torch.no_grad():
    ref_e = ef.with_params(opt).map(traj).detach()
def do_map():
    with torch.no_grad():
        ef.with_params(opt).map(traj)
block("energy_fn.with_params(opt).map(6400 frames)", do_map)
block("compute_loss_and_grad", lambda: O.compute_loss_and_grad(opt, ef, 1.0 / kT, loss_fn, traj, ref_e, [traj]), reps=5)

# ---- cfg1 energy check by hand
sim, cfg = defaults.default_configs_for("dna2")
flat = fp.pack_flat(fp.derive_flat(2, cfg, kt=sim["kT"], salt_conc=0.5, half_charged_ends=True), _lib.param_names())
top1, c1, q1 = generators.ideal_duplex(1000, model=2, seed=1234)
for dtype in (torch.float32, torch.float64):
    s = OxdnaSystem(2, top1.seq, top1.is_end, top1.bonded_neighbors, box=None, dtype=dtype, device=dev)
    s.set_params(flat)
    c = torch.as_tensor(c1, dtype=dtype, device=dev); q = torch.as_tensor(q1, dtype=dtype, device=dev)
    s.build_neighbors(c, 3.25, 0.0)
    e, g, _, _ = s.energy(c, q, grads=True)
    print(dtype, "skin 0.0:", e.cpu().numpy().round(4), s.neighbor_stats())
    s.build_neighbors(c, 3.25, 0.1)
    e, g, _, _ = s.energy(c, q, grads=True)
    print(dtype, "skin 0.1:", e.cpu().numpy().round(4), s.neighbor_stats())
