#!/usr/bin/env python3
"""DiffTRe across the GPUs of one node (BASELINE configs[4]): oxDNA2 32 bp duplex, 64 replicas sharded over the ranks.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        scripts/difftre_ranks.py [--replicas 64] [--iterations 3] [--check]
    python scripts/difftre_ranks.py                       (one rank)
    ... --rehearse-on-one-gpu                             (all ranks on cuda:0, gloo: a 1-GPU box cannot run RCCL)

Per optimisation iteration, on every rank:
  1. the rank's replicas (r with r mod world == rank) advance in ONE launch per MD step (HipMDSimulator.n_replicas)
     and their stored frames stay on the rank's GPU - no trajectory crosses xGMI;
  2. ``map`` + dU/dtheta of the local frames (one launch of the HIP energy kernel);
  3. ``distributed_compute_loss_and_grad``: all-reduce(MAX) of the softmax exponent and ONE all-reduce(SUM) of
     4 + 2K doubles give the global weights' moments, n_eff, <O> and d<O>/dtheta - identical on every rank;
  4. the same Adam step everywhere (no broadcast of parameters needed).
This is what replaces the reference's Ray fan-out of simulators and gather of whole trajectories into one objective
task (mythos/optimization/optimization.py:151-169, 225-247; objective.py:277-389).

--check: also all-gathers every frame to every rank and recomputes loss and gradient with the single-process
``compute_loss_and_grad``; the two must agree to 1e-10 (needs S_total frames to fit one GPU: they do).
Rank 0 prints one JSON line per iteration.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from mythos_amd import distributed as md  # noqa: E402
from mythos_amd.energy import dna2  # noqa: E402
from mythos_amd.energy.base import Quaternion, RigidBody, space  # noqa: E402
from mythos_amd.observables import PropellerTwist  # noqa: E402
from mythos_amd.optimization import objective as O  # noqa: E402
from mythos_amd.optimization.optimization import Adam, apply_updates  # noqa: E402
from mythos_amd.simulators.hip_md import HipMDSimulator, StaticSimulatorParams, nvt_langevin  # noqa: E402
from mythos_amd.simulators.io import SimulatorTrajectory  # noqa: E402
from mythos_amd.simulators.neighbors import NoNeighborList  # noqa: E402
from mythos_amd.utils import generators  # noqa: E402

KT = 296.15 * 0.1 / 300.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--replicas", type=int, default=64)
    ap.add_argument("--bp", type=int, default=32)
    ap.add_argument("--steps", type=int, default=2000, help="MD steps per replica per iteration")
    ap.add_argument("--save-every", type=int, default=100)
    ap.add_argument("--equilibration-frames", type=int, default=5)
    ap.add_argument("--iterations", type=int, default=3)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--dtype", choices=["f32", "f64"], default="f64")
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true")
    args = ap.parse_args()

    backend = "gloo" if args.rehearse_on_one_gpu else None
    rank, world, local = md.init(backend)
    dev = torch.device("cuda", 0 if args.rehearse_on_one_gpu else local)
    torch.cuda.set_device(dev)
    dtype = torch.float32 if args.dtype == "f32" else torch.float64
    # collectives run on the GPU tensors over RCCL; the one-GPU rehearsal meets over gloo on host copies
    to_comm = (lambda t: t.cpu()) if args.rehearse_on_one_gpu else (lambda t: t)

    top, c0, q0 = generators.ideal_duplex(args.bp, model=2, seed=21)
    n = top.n_nucleotides
    disp, shift = space.free()
    ef = dna2.create_default_energy_fn(topology=top, displacement_fn=disp)
    init = RigidBody(center=torch.as_tensor(c0, device=dev), orientation=Quaternion(vec=torch.as_tensor(q0, device=dev)))
    sp = StaticSimulatorParams(seq=top.seq, mass=(1.0, (1.0, 1.0, 1.0)), gamma=(KT / 2.5, KT / 7.5),
                               bonded_neighbors=top.bonded_neighbors, checkpoint_every=0, dt=0.005, kT=KT)
    mine = md.shard_replicas(args.replicas, rank, world)
    sim = HipMDSimulator(energy_fn=ef, simulator_params=sp, space=(disp, shift), simulator_init=nvt_langevin,
                         neighbors=NoNeighborList(unbonded_nbrs=top.unbonded_neighbors), save_every=args.save_every,
                         dtype=dtype, n_replicas=len(mine))
    half = n // 2
    ptwist = PropellerTwist(np.stack([np.arange(half), n - 1 - np.arange(half)], axis=1)[1:-1])
    target = 21.7
    opt = {"eps_stack_base": 1.3523, "eps_hb": 1.0678, "theta0_hb_4": float(np.pi)}
    adam = Adam(learning_rate=args.lr)
    adam_state = adam.init(opt)
    frames_per_replica = args.steps // args.save_every
    keep = slice(args.equilibration_frames, frames_per_replica)
    state = {"init_state": init, "key": 1000 + rank}

    for it in range(args.iterations):
        t0 = time.perf_counter()
        out = sim.run(opt, state["init_state"], args.steps, key=state["key"])
        state = {"init_state": out.state["init_state"], "key": 1000 + rank + (it + 1) * world}  # a fresh stream per rank and iteration
        traj = out.observables[0]  # replica-major (R_local * S, n, .)
        sel = torch.cat([torch.arange(r * frames_per_replica, (r + 1) * frames_per_replica)[keep] for r in range(len(mine))])
        local_traj = traj.slice(sel)
        torch.cuda.synchronize(dev)
        t_sim = time.perf_counter() - t0

        t0 = time.perf_counter()
        with torch.no_grad():
            ref_e = ef.with_params(opt).map(local_traj).detach()
        beta = 1.0 / KT

        class _Comm:  # energy function whose map() lands where the collectives run
            def __init__(self, f):
                self.f = f

            def with_params(self, *a, **k):
                return _Comm(self.f.with_params(*a, **k))

            def map(self, st):
                return to_comm(self.f.map(st))

        (loss, (neff, mean_o, _)), grads = O.distributed_compute_loss_and_grad(
            opt, _Comm(ef), beta, lambda st: to_comm(ptwist(st)), lambda m: (m - target) ** 2, local_traj, to_comm(ref_e))
        torch.cuda.synchronize(dev)
        t_rw = time.perf_counter() - t0

        rec = {"iteration": it, "world": world, "replicas": args.replicas, "replicas_this_rank": len(mine),
               "frames_total": args.replicas * (frames_per_replica - args.equilibration_frames), "loss": float(loss),
               "neff": float(neff), "propeller_twist": float(mean_o), "grads": {k: float(v) for k, v in grads.items()},
               "md_s": t_sim, "reweight_s": t_rw, "collective_doubles": 1 + 4 + 2 * len(opt)}
        if args.check:
            # every frame to every rank, replica-id order; then the single-process autograd value on all of them
            per = frames_per_replica - args.equilibration_frames
            cs = md.all_gather_observables(to_comm(local_traj.center.reshape(len(mine), per, n, 3).contiguous()))
            qs = md.all_gather_observables(to_comm(local_traj.orientation.vec.reshape(len(mine), per, n, 4).contiguous()))
            full = SimulatorTrajectory(center=cs.reshape(-1, n, 3).to(dev), orientation=Quaternion(vec=qs.reshape(-1, n, 4).to(dev)))
            with torch.no_grad():
                ref_all = ef.with_params(opt).map(full).detach()

            def loss_fn(ref_states, weights, energy_fn, opt_params, observables):  # noqa: ARG001
                m = (weights * ptwist(ref_states).to(weights.dtype)).sum()
                return (m - target) ** 2, (("propeller_twist", m.detach()), {})

            (l1, (neff1, meas, _)), g1 = O.compute_loss_and_grad(opt, ef, beta, loss_fn, full, ref_all, [full])
            err = max(abs(float(g1[k]) - float(grads[k])) / max(1.0, abs(float(g1[k]))) for k in opt)
            assert abs(float(l1) - float(loss)) <= 1e-10 * max(1.0, abs(float(l1))), (float(l1), float(loss))
            assert abs(float(neff1) - float(neff)) <= 1e-10 and err <= 1e-10, (err, g1, grads)
            rec["check"] = {"frames": int(full.center.shape[0]), "max_rel_grad_err": err, "loss_single_process": float(l1)}
        if rank == 0:
            print(json.dumps(rec), flush=True)
        upd, adam_state = adam.update({k: torch.as_tensor(v) for k, v in grads.items()}, adam_state, opt)
        opt = {k: float(v) for k, v in apply_updates(opt, upd).items()}

    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
