"""MD sampler on the fused HIP Langevin kernel, shaped like the reference's ``JaxMDSimulator``
(mythos/simulators/jax_md/jaxmd.py:20-103) and ``StaticSimulatorParams``
(mythos/simulators/jax_md/utils.py:129-159).

    sim = HipMDSimulator(energy_fn=..., simulator_params=..., space=space.free(),
                         simulator_init=nvt_langevin, neighbors=NoNeighborList(top.unbonded_neighbors))
    out = sim.run(opt_params, init_state, n_steps, key)      # SimulatorOutput([SimulatorTrajectory])

The reference scans ``step_fn`` and stores the state after EVERY step; ``run`` does the same by
default (``save_every=1``) but takes a cadence, and the whole loop is one C-ABI call.
"""

from __future__ import annotations

import dataclasses as dc
from typing import Any, Callable

import numpy as np
import torch

from mythos_amd import _lib
from mythos_amd.energy import flat_params as fp
from mythos_amd.energy.base import ComposedEnergyFunction, EnergyFunction, Quaternion, RigidBody, _np
from mythos_amd.energy import terms as _terms
from mythos_amd.hip_system import LangevinIntegrator, OxdnaSystem
from mythos_amd.simulators.base import Simulator, SimulatorOutput
from mythos_amd.simulators.io import SimulatorTrajectory
from mythos_amd.simulators.neighbors import NoNeighborList, VerletNeighborList


@dc.dataclass
class StaticSimulatorParams:
    """mythos/simulators/jax_md/utils.py:129-159.  ``mass`` / ``gamma`` are RigidBody-shaped in the
    reference (center = translational, orientation = rotational); here plain pairs."""

    seq: Any
    mass: Any  # RigidBody(center=m, orientation=(I1, I2, I3)) or (m, (I1, I2, I3))
    gamma: Any  # RigidBody(center=gamma_t, orientation=gamma_r) or (gamma_t, gamma_r)
    bonded_neighbors: Any
    checkpoint_every: int
    dt: float
    kT: float  # noqa: N815

    @property
    def sim_init_fn(self) -> dict:
        return {"dt": self.dt, "kT": self.kT, "gamma": self.gamma}

    @property
    def init_fn(self) -> dict:
        return {"mass": self.mass}

    @property
    def step_fn(self) -> dict:
        return {}


def _pair(x):
    if isinstance(x, RigidBody):
        return x.center, x.orientation
    return x


def nvt_langevin(energy_or_force_fn, shift_fn, dt, kT, gamma, **_):  # noqa: N803, ARG001
    """Marker with the signature of ``jax_md.simulate.nvt_langevin``: HipMDSimulator recognises it and
    runs the fused kernel instead of returning (init_fn, step_fn) closures."""
    return {"integrator": "nvt_langevin", "dt": dt, "kT": kT, "gamma": gamma}


@dc.dataclass(frozen=True, kw_only=True)
class HipMDSimulator(Simulator):
    energy_fn: EnergyFunction
    simulator_params: StaticSimulatorParams
    space: Any = None
    simulator_init: Callable = nvt_langevin
    neighbors: Any = None
    save_every: int = 1
    dtype: torch.dtype = torch.float32
    device: Any = None
    # defaults for callers that only pass parameters (SimpleOptimizer.step: simulator.run(params, **state))
    init_state: Any = None
    n_steps: int | None = None
    key: int = 0

    def run(self, opt_params: dict, init_state: RigidBody | None = None, n_steps: int | None = None, key: int | None = None,
            **_) -> SimulatorOutput:
        init_state = self.init_state if init_state is None else init_state
        n_steps = self.n_steps if n_steps is None else n_steps
        key = self.key if key is None else key
        if init_state is None or n_steps is None:
            raise ValueError("HipMDSimulator.run needs init_state and n_steps (as arguments or as fields)")
        if self.simulator_init is not nvt_langevin:
            raise NotImplementedError("HipMDSimulator implements nvt_langevin (the integrator every reference example uses)")
        ef = self.energy_fn.with_params(opt_params) if opt_params else self.energy_fn
        if not isinstance(ef, ComposedEnergyFunction):
            ef = ComposedEnergyFunction(energy_fns=[ef])
        first = ef.energy_fns[0]
        geom = next(fn.transform_fn for fn in ef.energy_fns if fn.transform_fn is not None)
        model = geom.model
        sections = {"geometry": geom.params}
        tw = [0.0] * 8
        w_user = ef.weights if ef.weights is not None else torch.ones(len(ef.energy_fns), dtype=torch.float64)
        kt_e = salt = hce = None
        from mythos_amd.energy.base import TERM_ORDER

        for fn, w in zip(ef.energy_fns, w_user):
            sections[fn.term] = {n: fn.params[n] for n in (*type(fn.params).required_params, *type(fn.params).optional_params)}
            tw[TERM_ORDER.index(fn.term)] = float(w)
            if "kt" in fn.params and kt_e is None:
                kt_e = fn.params["kt"]
            if fn.term == "debye":
                salt, hce = fn.params["salt_conc"], bool(fn.params["half_charged_ends"])
        _terms.fill_missing_sections(model, sections)
        sp = self.simulator_params
        flat = fp.pack_flat(
            fp.derive_flat(model, sections, kt=sp.kT if kt_e is None else kt_e, salt_conc=0.5 if salt is None else salt,
                           half_charged_ends=True if hce is None else hce, term_weights=tw),
            _lib.param_names(),
        )
        dev = torch.device(self.device) if self.device is not None else init_state.center.device
        if dev.type != "cuda":
            dev = torch.device("cuda", torch.cuda.current_device())
        box = getattr(first.displacement_fn, "box", None)
        system = OxdnaSystem(model, _np(first.seq), None if first.is_end is None else _np(first.is_end),
                             _np(first.bonded_neighbors), box=box, dtype=self.dtype, device=dev)
        system.set_params(flat.detach())
        mass, inertia = _pair(sp.mass)
        gamma_t, gamma_r = _pair(sp.gamma)
        integ = LangevinIntegrator(system, dt=sp.dt, kT=sp.kT, gamma_t=float(gamma_t), gamma_r=float(gamma_r),
                                   mass=float(np.asarray(mass).reshape(-1)[0]),
                                   inertia=np.asarray(inertia, dtype=np.float64).reshape(-1)[:3], seed=int(key))
        nb = self.neighbors
        if isinstance(nb, VerletNeighborList):
            integ.set_neighbor_policy(nb.r_cutoff, nb.dr_threshold, nb.rebuild_every)
        else:
            pairs = nb.idx if nb is not None else first.unbonded_neighbors
            from mythos_amd.energy.base import _pairs_2xP

            system.set_neighbors(_pairs_2xP(pairs, system.n))
        c = init_state.center.to(device=dev, dtype=self.dtype).contiguous().clone()
        q = init_state.orientation.vec.to(device=dev, dtype=self.dtype).contiguous().clone()
        p, ang = integ.init_momenta()
        tc, tq, et = integ.run(c, q, p, ang, int(n_steps), save_every=self.save_every)
        n_saved = 0 if tc is None else tc.shape[0]
        traj = SimulatorTrajectory(
            center=tc if tc is not None else c[None][:0],
            orientation=Quaternion(vec=tq if tq is not None else q[None][:0]),
            temperature=torch.full((n_saved,), sp.kT, dtype=torch.float64, device=dev),
            metadata=None if et is None else {"energy_terms": et[:, :8], "kinetic": et[:, 8:]},
        )
        final = RigidBody(center=c, orientation=Quaternion(vec=q))
        # the state handed to the next run(params, **state): continue from the last configuration with a new key
        return SimulatorOutput(observables=[traj], state={"init_state": final, "key": int(key) + 1, "final_state": final,
                                                          "momentum": (p, ang), "steps": integ.step})
