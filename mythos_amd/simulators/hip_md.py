"""MD sampler on the fused HIP Langevin kernel, shaped like the reference's ``JaxMDSimulator``
(mythos/simulators/jax_md/jaxmd.py:20-103) and ``StaticSimulatorParams``
(mythos/simulators/jax_md/utils.py:129-159).

    sim = HipMDSimulator(energy_fn=..., simulator_params=..., space=space.free(),
                         simulator_init=nvt_langevin, neighbors=NoNeighborList(top.unbonded_neighbors))
    out = sim.run(opt_params, init_state, n_steps, key)      # SimulatorOutput([SimulatorTrajectory])

The reference scans ``step_fn`` and stores the state after EVERY step - ``state.position`` and nothing else
(jaxmd.py:84-99); ``run`` does the same by default (``save_every=1``, positions only: the step launch that produces a
saved state writes it to its row, no other cost) but takes a cadence, and the whole loop is one C-ABI call.
``trace_energy=True`` adds the term and kinetic energies of the saved steps (``metadata``), at the price of the
energy-trace instantiation of the step kernel; the reference's way to the energies of a trajectory is ``energy_fn.map``.

The device-side objects of a run - the system handle with its neighbour rows, the integrator with its frames - are kept
on the simulator between calls, keyed by everything they were built from (topology, model, precision, device, replica
count, list, integrator constants).  A later ``run`` with other parameters uploads the 2 KB parameter vector, sets the
key and the step counter, and steps (the reference: "parameter update: negligible", docs/source/energy_functions.rst:51-54;
its own run re-traces nothing when only ``opt_params`` change, jaxmd.py:60-68).
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
from mythos_amd.simulators.neighbors import NoNeighborList, VerletNeighborList  # noqa: F401  (NoNeighborList: re-exported, the docstring's example)


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


@dc.dataclass
class NVTLangevinState:
    """What ``init_fn`` / ``step_fn`` hand back: the fields the reference reads off jax_md's state (``.position``, ``.mass``:
    mythos/simulators/jax_md/utils.py:19-28, jaxmd.py:84-92) plus the momenta and the step counter.  Self-contained: the
    tensors are copies of the integrator's resident frames, so an older state stays valid and can be stepped again."""

    position: RigidBody
    momentum: RigidBody  # center: linear momenta (N, 3); orientation: body-frame angular momenta (N, 3)
    mass: RigidBody
    step: int
    _version: int = dc.field(default=-1, repr=False)


class NVTLangevin:
    """``init_fn, step_fn = nvt_langevin(energy_fn, shift_fn, dt=, kT=, gamma=)`` - the integrator plug of the reference
    (``simulator_init``, mythos/simulators/jax_md/utils.py:19-28; call sites jaxmd.py:73-92) on the fused HIP kernel.

    ``init_fn(key, R, mass=..., **kw) -> state`` draws the momenta at kT from ``key`` and loads the state into the
    integrator; ``step_fn(state, **kw) -> state`` advances ONE step (one launch) and reads the frames back, so that every
    state is a value as it is in JAX: stepping the newest state continues the resident frames, stepping an older one
    reloads it first and reproduces what it produced before to rounding (the noise of a step is a function of key, step and
    nucleotide; a reloaded state's closing half kick and the next opening one are two additions where the resident frames
    have one).  This is the reference's calling shape, at one C-ABI call and four small copies per step; the loop the
    reference builds from it (``run_fn``) is ``HipMDSimulator.run`` - one call for all steps - and that is the fast path.

    ``energy_or_force_fn`` must be one of this package's energy functions (it names the topology, the model and the
    parameters the kernel steps with); an arbitrary Python callable cannot be run by the HIP kernel and is refused.
    ``dtype`` defaults to the reference's fp64; ``neighbors`` as for HipMDSimulator (default: the energy function's pairs).
    """

    def __init__(self, energy_or_force_fn, shift_fn, dt, kT, gamma, dtype=torch.float64, device=None, neighbors=None):  # noqa: N803
        if not isinstance(energy_or_force_fn, EnergyFunction):
            raise NotImplementedError("nvt_langevin steps an energy function of this package (dna1 / dna2 / rna2 / na1); "
                                      f"got {type(energy_or_force_fn).__name__}")
        self.energy_fn, self.shift_fn = energy_or_force_fn, shift_fn
        self.dt, self.kT, self.gamma = float(dt), float(kT), gamma
        self.dtype, self.device, self.neighbors = dtype, device, neighbors
        self._sim = self._integ = None
        self._version = 0

    def __iter__(self):  # init_fn, step_fn = nvt_langevin(...)
        yield self.init_fn
        yield self.step_fn

    def __getitem__(self, k):  # (the marker this function used to return)
        return {"integrator": "nvt_langevin", "dt": self.dt, "kT": self.kT, "gamma": self.gamma}[k]

    def _read(self, mass, step):
        s = self._integ.system
        c, p, ang = (torch.empty((s.n, 3), dtype=s.dtype, device=s.device) for _ in range(3))
        q = torch.empty((s.n, 4), dtype=s.dtype, device=s.device)
        self._integ.store(c, q, p, ang)
        self._version += 1
        return NVTLangevinState(position=RigidBody(center=c, orientation=Quaternion(vec=q)), momentum=RigidBody(center=p, orientation=ang),
                                mass=mass, step=step, _version=self._version)

    def init_fn(self, key, R, mass=None, **_):  # noqa: N803
        if mass is None:
            mass = RigidBody(center=1.0, orientation=(1.0, 1.0, 1.0))
        elif not isinstance(mass, RigidBody):
            m, inertia = mass if isinstance(mass, (tuple, list)) else (mass, (1.0, 1.0, 1.0))
            mass = RigidBody(center=m, orientation=inertia)
        first = self.energy_fn.energy_fns[0] if isinstance(self.energy_fn, ComposedEnergyFunction) else self.energy_fn
        sp = StaticSimulatorParams(seq=first.seq, mass=mass, gamma=self.gamma, bonded_neighbors=first.bonded_neighbors, checkpoint_every=0,
                                   dt=self.dt, kT=self.kT)
        if self._sim is not None:
            self._sim.release()
        self._sim = HipMDSimulator(energy_fn=self.energy_fn, simulator_params=sp, space=(None, self.shift_fn), neighbors=self.neighbors,
                                   save_every=0, dtype=self.dtype, device=self.device)
        system, integ, dev, _, _ = self._sim._prepare(None, int(key), R.center.device)
        self._integ = integ
        c = R.center.to(device=dev, dtype=self.dtype).contiguous().clone()
        q = R.orientation.vec.to(device=dev, dtype=self.dtype).contiguous().clone()
        p, ang = integ.init_momenta()
        integ.load(c, q, p, ang)
        return self._read(mass, 0)

    def step_fn(self, state: NVTLangevinState, **_):
        if self._integ is None:
            raise ValueError("step_fn: call init_fn first")
        integ = self._integ
        if state._version != self._version:  # not the resident frames: make it so
            integ.load(state.position.center.contiguous(), state.position.orientation.vec.contiguous(), state.momentum.center.contiguous(),
                       state.momentum.orientation.contiguous())
            integ.step = int(state.step)
        integ.advance(1)
        return self._read(state.mass, int(state.step) + 1)

    def close(self) -> None:
        if self._sim is not None:
            self._sim.release()
        self._sim = self._integ = None


def nvt_langevin(energy_or_force_fn, shift_fn, dt, kT, gamma, **kw):  # noqa: N803
    """Signature of ``jax_md.simulate.nvt_langevin``.  Called, it returns the (init_fn, step_fn) pair (NVTLangevin);
    passed uncalled as ``simulator_init`` - what the reference's simulators are given - HipMDSimulator recognises it and
    runs the whole loop in one call."""
    return NVTLangevin(energy_or_force_fn, shift_fn, dt, kT, gamma, **kw)


@dc.dataclass(frozen=True, kw_only=True)
class HipMDSimulator(Simulator):
    energy_fn: EnergyFunction
    simulator_params: StaticSimulatorParams
    space: Any = None
    simulator_init: Callable = nvt_langevin
    neighbors: Any = None
    save_every: int = 1
    # The reference computes in fp64 (jax_enable_x64).  The default here is fp32 - north_star's "fp32 forces at 1e-3", with
    # centres carried as hi + lo pairs - because sampling does not need more and runs 1.7x faster;
    # pass dtype=torch.float64 for the reference's precision (the step-by-step oracle tests run there).
    dtype: torch.dtype = torch.float32
    device: Any = None
    trace_energy: bool = False  # energies of the saved steps in trajectory.metadata (the energy-trace instantiation)
    # defaults for callers that only pass parameters (SimpleOptimizer.step: simulator.run(params, **state))
    init_state: Any = None
    n_steps: int | None = None
    key: int = 0
    # Independent replicas of the system advanced by ONE launch per step (free space only): the copies are laid out
    # on a grid far apart and integrated as one system, so 64 replicas of a 64-nt duplex cost a step of a 4 096-nt
    # system, not 64 steps.  Every nucleotide has its own Philox stream, so replicas are statistically independent.
    # The reference runs replicas as separate simulator instances (mythos/simulators/base.py MultiSimulator, Ray).
    n_replicas: int = 1
    # device-side objects kept between calls: {key: (OxdnaSystem, LangevinIntegrator, pinned python objects)}
    _resident: dict = dc.field(default_factory=dict, init=False, repr=False, compare=False)

    def __getstate__(self):  # (the reference's simulators travel through Ray: handles stay behind)
        d = {f.name: getattr(self, f.name) for f in dc.fields(self) if f.name != "_resident"}
        return d

    def __setstate__(self, d):
        for k, v in d.items():
            object.__setattr__(self, k, v)
        object.__setattr__(self, "_resident", {})

    def release(self) -> None:
        """Free the device-side objects kept from earlier runs."""
        for system, integ, _ in self._resident.values():
            integ.close()
            system.close()
        self._resident.clear()

    def _prepare(self, opt_params, key, state_device):
        """The device-side objects of this simulator at ``opt_params``: (system, integrator, device, replicas, nucleotides
        per replica) - built on first use, afterwards the kept ones with the new parameter vector, key and step 0."""
        ef = self.energy_fn.with_params(opt_params) if opt_params else self.energy_fn
        if not isinstance(ef, ComposedEnergyFunction):
            ef = ComposedEnergyFunction(energy_fns=[ef])
        first = ef.energy_fns[0]
        from mythos_amd.energy.base import pseq_request

        # a probabilistic sequence rides along into the dynamics, as it does in the reference - there the stacking /
        # hydrogen-bonding configurations carry pseq into whatever energy function a simulator steps with
        # (dna1/stacking.py:284-285, hydrogen_bonding.py:330-331)
        pseq = pseq_request(ef.energy_fns)
        geom = next(fn.transform_fn for fn in ef.energy_fns if fn.transform_fn is not None)
        model = geom.model
        _terms.check_term_models(model, ef.energy_fns)
        is_rna = None
        sections = {"geometry": geom.params} if model != 4 else None
        tw = [0.0] * 8
        w_user = ef.weights if ef.weights is not None else torch.ones(len(ef.energy_fns), dtype=torch.float64)
        kt_e = salt = hce = None
        from mythos_amd.energy.base import TERM_ORDER

        for fn, w in zip(ef.energy_fns if model != 4 else [], w_user):
            sections[fn.term] = {n: fn.params[n] for n in (*type(fn.params).required_params, *type(fn.params).optional_params)
                                 if n not in ("pseq", "pseq_constraints")}
            tw[TERM_ORDER.index(fn.term)] = float(w)
            if "kt" in fn.params and kt_e is None:
                kt_e = fn.params["kt"]
            if fn.term == "debye":
                salt, hce = fn.params["salt_conc"], bool(fn.params["half_charged_ends"])
        sp = self.simulator_params
        if model == 4:
            # oxNA: three flat vectors and the nucleotide types (the step kernel's MODEL 4 instantiation)
            from mythos_amd.energy.base import na1_flat_and_types

            flat, is_rna, _, _ = na1_flat_and_types(ef.energy_fns, ef.weights, geom, kt_default=sp.kT)
        else:
            _terms.fill_missing_sections(model, sections)
        flat = flat if model == 4 else fp.pack_flat(
            fp.derive_flat(model, sections, kt=sp.kT if kt_e is None else kt_e, salt_conc=0.5 if salt is None else salt,
                           half_charged_ends=True if hce is None else hce, term_weights=tw, numbers_ok=True),
            _lib.param_names(),
        )
        dev = torch.device(self.device) if self.device is not None else state_device
        if dev.type != "cuda":
            dev = torch.device("cuda", torch.cuda.current_device())
        box = getattr(first.displacement_fn, "box", None)
        n_rep = int(self.n_replicas)
        n_one = int(_np(first.seq).shape[0])
        seq_a, end_a, bonded_a = _np(first.seq), None if first.is_end is None else _np(first.is_end), _np(first.bonded_neighbors)
        if n_rep > 1:
            if box is not None:
                raise ValueError("HipMDSimulator: replicas are batched in free space; the energy function has a periodic box")
            bonded_2 = np.asarray(bonded_a).reshape(-1, 2)
            seq_a = np.tile(np.asarray(seq_a), n_rep)
            end_a = None if end_a is None else np.tile(np.asarray(end_a), n_rep)
            bonded_a = np.concatenate([bonded_2 + r * n_one for r in range(n_rep)], axis=0)
        if n_rep > 1 and is_rna is not None:
            is_rna = np.tile(np.asarray(is_rna), n_rep)
        mass, inertia = _pair(sp.mass)
        gamma_t, gamma_r = _pair(sp.gamma)
        mass_f = float(np.asarray(mass).reshape(-1)[0])
        inertia_a = np.asarray(inertia, dtype=np.float64).reshape(-1)[:3]
        nb = self.neighbors
        pairs_obj = None if isinstance(nb, VerletNeighborList) else (nb.idx if nb is not None else first.unbonded_neighbors)
        bts = lambda a: None if a is None else np.ascontiguousarray(a).tobytes()  # noqa: E731
        key_res = (
            model, n_rep, self.dtype, str(dev), bts(np.asarray(seq_a, dtype=np.int32)), bts(None if end_a is None else np.asarray(end_a, dtype=np.uint8)),
            bts(np.asarray(bonded_a, dtype=np.int32)), None if box is None else tuple(np.asarray(box, dtype=np.float64).reshape(-1).tolist()),
            bts(None if is_rna is None else np.asarray(is_rna, dtype=np.uint8)),
            ("verlet", float(nb.r_cutoff), float(nb.dr_threshold), int(nb.rebuild_every)) if pairs_obj is None else ("pairs", id(pairs_obj)),
            float(sp.dt), float(sp.kT), float(gamma_t), float(gamma_r), mass_f, tuple(inertia_a.tolist()),
        )
        entry = self._resident.get(key_res)
        if entry is None:
            system = OxdnaSystem(model, seq_a, end_a, bonded_a, box=box, dtype=self.dtype, device=dev, is_rna=is_rna)
            integ = LangevinIntegrator(system, dt=sp.dt, kT=sp.kT, gamma_t=float(gamma_t), gamma_r=float(gamma_r), mass=mass_f,
                                       inertia=inertia_a, seed=int(key))
            system.set_params(flat.detach())
            if pairs_obj is None:
                integ.set_neighbor_policy(nb.r_cutoff, nb.dr_threshold, nb.rebuild_every)
            else:
                from mythos_amd.energy.base import _pairs_2xP

                p2 = _pairs_2xP(pairs_obj, n_one)
                if n_rep > 1:
                    p2 = np.concatenate([np.asarray(p2).reshape(-1, 2) + r * n_one for r in range(n_rep)], axis=0)
                system.set_neighbors(p2)
            while len(self._resident) >= 4:  # (a simulator serves one system; a few variants at most)
                old_system, old_integ, _ = self._resident.pop(next(iter(self._resident)))
                old_integ.close()
                old_system.close()
            # (pairs_obj is pinned: its id is part of the key)
            self._resident[key_res] = (system, integ, pairs_obj)
            had_pseq = False
        else:
            system, integ, _ = entry
            system.set_params(flat.detach())  # 2 KB; everything derived from it on the device follows (param_epoch)
            integ.set_seed(int(key))
            integ.step = 0  # a run starts its noise stream at (key, step 0), as a new integrator would
            had_pseq = system._pseq_terms != 0
        if pseq is not None:
            marg, unit, bp, terms = pseq
            if n_rep > 1:  # every replica its own copy of the base pairs
                n_bp = int(bp.shape[0]) if (unit >= 0).any() else 0
                unit = np.concatenate([np.where(unit >= 0, unit + 2 * n_bp * r, -1) for r in range(n_rep)])
                marg, bp = np.tile(marg, (n_rep, 1)), (np.tile(bp, (n_rep, 1)) if n_bp > 0 else bp)
            system.set_pseq(marg, unit, bp, terms)
        elif had_pseq:
            system.set_pseq()
        return system, integ, dev, n_rep, n_one

    def run(self, opt_params: dict, init_state: RigidBody | None = None, n_steps: int | None = None, key: int | None = None,
            **_) -> SimulatorOutput:
        init_state = self.init_state if init_state is None else init_state
        n_steps = self.n_steps if n_steps is None else n_steps
        key = self.key if key is None else key
        if init_state is None or n_steps is None:
            raise ValueError("HipMDSimulator.run needs init_state and n_steps (as arguments or as fields)")
        if self.simulator_init is not nvt_langevin:
            raise NotImplementedError("HipMDSimulator implements nvt_langevin (the integrator every reference example uses)")
        system, integ, dev, n_rep, n_one = self._prepare(opt_params, key, init_state.center.device)
        sp, nb = self.simulator_params, self.neighbors
        c = init_state.center.to(device=dev, dtype=self.dtype).contiguous().clone()
        q = init_state.orientation.vec.to(device=dev, dtype=self.dtype).contiguous().clone()
        offsets = None
        if n_rep > 1:
            # (n, 3) initial state: every replica starts from it; (R, n, 3): one start per replica
            c = (c if c.dim() == 3 else c[None].expand(n_rep, -1, -1)).reshape(n_rep, n_one, 3).clone()
            q = (q if q.dim() == 3 else q[None].expand(n_rep, -1, -1)).reshape(n_rep, n_one, 4).clone()
            # every replica is centred on its grid node for the run (free space: a translation changes nothing) and gets
            # its own centre of mass back afterwards, so drift accumulated over earlier runs never eats into the spacing
            com = c.mean(dim=1, keepdim=True)
            extent = float((c - com).norm(dim=-1).max())
            r_list = (nb.r_cutoff + nb.dr_threshold) if isinstance(nb, VerletNeighborList) else 4.0
            spacing = 2.0 * extent + 8.0 * r_list + 64.0  # out of each other's list range for as long as a run diffuses
            side = int(np.ceil(n_rep ** (1.0 / 3.0)))
            grid = torch.as_tensor([[r % side, (r // side) % side, r // (side * side)] for r in range(n_rep)], dtype=self.dtype, device=dev)
            offsets = (grid * spacing)[:, None, :] - com
            c = (c + offsets).reshape(n_rep * n_one, 3).contiguous()
            q = q.reshape(n_rep * n_one, 4).contiguous()
        p, ang = integ.init_momenta()
        tc, tq, et = integ.run(c, q, p, ang, int(n_steps), save_every=self.save_every, want_energy=self.trace_energy)
        if n_rep > 1:
            # states of all replicas, replica-major: (R * S, n, .); the offsets of the grid come off again
            def unbatch(t, width, off):
                t = t.reshape(t.shape[0], n_rep, n_one, width)
                if off is not None:
                    t = t - off[None]
                return t.transpose(0, 1).reshape(-1, n_one, width)

            tc = None if tc is None else unbatch(tc, 3, offsets)
            tq = None if tq is None else unbatch(tq, 4, None)
            et = None  # the fused trace sums over the whole launch; per-replica energies come from energy_fn
            c = c.reshape(n_rep, n_one, 3) - offsets
            q = q.reshape(n_rep, n_one, 4)
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
