"""Energy-function plugin surface with the reference's protocol, evaluated by the HIP kernels.

Mirrors mythos/energy/base.py:
  ``EnergyFunction``            :24-93   __call__(body) / map(traj) / with_params / with_props /
                                         with_noopt / params_dict / opt_params
  ``BaseEnergyFunction``        :115-212 one term bound to a topology
  ``ComposedEnergyFunction``    :215-434 linear combination sharing one parameter namespace
  ``QualifiedComposedEnergyFunction`` :437-462

A body is a ``RigidBody(center, orientation=Quaternion(vec))`` of torch tensors on the GPU; a
trajectory is the same with a leading frame axis.  Every evaluation is ONE kernel launch over all
frames that returns the eight term energies; derivatives come from the same launch through
``torch.autograd`` (no tape: the kernels emit dU/dcenter, dU/dquat and dU/dparam analytically and
the host applies d(flat)/d(theta), see flat_params.py).  There is no CPU fallback.
"""

from __future__ import annotations

import hashlib

import dataclasses as dc
from abc import ABC, abstractmethod
from typing import Any, Callable

import numpy as np
import torch

from mythos_amd import _lib
from mythos_amd.energy import flat_params as fp
from mythos_amd.energy.configuration import BaseConfiguration

ERR_PARAM_NOT_FOUND = "Parameter '{key}' not found in {class_name}"
ERR_COMPOSED_ENERGY_FN_LEN_MISMATCH = "Weights must have the same length as energy functions"
ERR_COMPOSED_ENERGY_FN_TYPE_ENERGY_FNS = "energy_fns must be a list of energy functions"

TERM_ORDER = (
    "fene",
    "bonded_excluded_volume",
    "stacking",
    "unbonded_excluded_volume",
    "hydrogen_bonding",
    "cross_stacking",
    "coaxial_stacking",
    "debye",
)


# ---------------------------------------------------------------------------------------------
# state containers (jax_md.rigid_body.RigidBody / Quaternion stand-ins)
# ---------------------------------------------------------------------------------------------
@dc.dataclass(frozen=True)
class Quaternion:
    vec: torch.Tensor  # (..., N, 4) as [w, x, y, z]


@dc.dataclass(frozen=True)
class RigidBody:
    center: torch.Tensor  # (..., N, 3)
    orientation: Quaternion

    def __getitem__(self, key) -> "RigidBody":
        return RigidBody(self.center[key], Quaternion(self.orientation.vec[key]))


@dc.dataclass(frozen=True)
class Displacement:
    """What the reference gets from ``jax_md.space``: free space or a periodic box."""

    box: np.ndarray | None = None

    def __call__(self, a, b):
        d = a - b
        if self.box is None:
            return d
        side = torch.as_tensor(self.box, dtype=d.dtype, device=d.device)
        return torch.remainder(d + 0.5 * side, side) - 0.5 * side


class space:  # noqa: N801 - mirrors jax_md.space's module-style use
    @staticmethod
    def free():
        disp = Displacement(None)
        return disp, (lambda r, dr, **_: r + dr)

    @staticmethod
    def periodic(box):
        b = np.broadcast_to(np.asarray(box, dtype=np.float64), (3,)).copy()
        disp = Displacement(b)

        def shift(r, dr, **_):
            return torch.remainder(r + dr, torch.as_tensor(b, dtype=r.dtype, device=r.device))

        return disp, shift


DEFAULT_DISPLACEMENT = space.free()[0]


@dc.dataclass(frozen=True)
class Geometry:
    """Site offsets of a nucleotide: the kwargs of ``Nucleotide.from_rigid_body``
    (mythos/energy/dna1/nucleotide.py:29-53, dna2/nucleotide.py:30-58).  Plays the role of the
    reference's ``transform_fn``; the site algebra itself runs inside the kernels."""

    model: int
    params: dict

    def __call__(self, body: RigidBody) -> RigidBody:
        return body


# ---------------------------------------------------------------------------------------------
# HIP backend: system cache + autograd bridge
# ---------------------------------------------------------------------------------------------
_SYSTEMS: dict = {}


def _pairs_2xP(unbonded, n: int) -> np.ndarray:
    """Accept (P,2) or the reference's (2,P) layout; drop padded entries (index >= n)."""
    u = np.asarray(unbonded.detach().cpu().numpy() if isinstance(unbonded, torch.Tensor) else unbonded)
    if u.ndim != 2:
        raise ValueError("unbonded_neighbors must be 2-D")
    if u.shape[0] == 2 and u.shape[1] != 2:
        u = u.T
    elif u.shape[1] != 2:
        raise ValueError("unbonded_neighbors must have shape (P, 2) or (2, P)")
    u = u[(u[:, 0] < n) & (u[:, 1] < n)]
    return np.ascontiguousarray(u, dtype=np.int32)


def _get_system(model, seq, is_end, bonded, unbonded, box, dtype, device, is_rna=None):
    from mythos_amd.hip_system import OxdnaSystem

    seq = np.ascontiguousarray(seq, dtype=np.int32)
    bonded = np.ascontiguousarray(bonded, dtype=np.int32)
    is_end_b = None if is_end is None else np.ascontiguousarray(is_end, dtype=np.uint8)
    is_rna_b = None if is_rna is None else np.ascontiguousarray(is_rna, dtype=np.uint8)
    key = (
        model, seq.tobytes(), bonded.tobytes(), None if is_end_b is None else is_end_b.tobytes(),
        None if box is None else tuple(np.asarray(box, dtype=np.float64).tolist()), dtype, str(device),
        None if is_rna_b is None else is_rna_b.tobytes(),
    )
    entry = _SYSTEMS.get(key)
    if entry is None:
        entry = {"sys": OxdnaSystem(model, seq, is_end_b, bonded, box=box, dtype=dtype, device=device, is_rna=is_rna_b), "pairs": None,
                 "flat": None, "pseq": None}
        _SYSTEMS[key] = entry
        if len(_SYSTEMS) > 16:
            _SYSTEMS.pop(next(iter(_SYSTEMS)))
    if unbonded is not None:
        pairs = _pairs_2xP(unbonded, seq.shape[0])
        # content, never identity: ids are reused after garbage collection and arrays are mutated in place
        tag = (pairs.shape, pairs.tobytes() if pairs.size < 65_536 else hashlib.blake2b(pairs.data, digest_size=16).digest())
        if entry["pairs"] != tag:
            entry["sys"].set_neighbors(pairs)
            entry["pairs"] = tag
    return entry


def pseq_request(energy_fns):
    """The probabilistic sequence a composed function asks for: (marginals, unit, bp_probs, terms) or None.
    Stacking and hydrogen bonding each carry ``pseq`` / ``pseq_constraints`` (dna1/stacking.py:54-55,
    dna1/hydrogen_bonding.py:94-95); a system has ONE sequence distribution, so if both carry one it must be the same."""
    from mythos_amd.input.sequence_constraints import kernel_tables

    found, terms = None, 0
    for fn in energy_fns:
        bit = {"stacking": 1, "hydrogen_bonding": 2}.get(fn.term)
        if bit is None or "pseq" not in fn.params or fn.params["pseq"] is None:
            continue
        sc = fn.params["pseq_constraints"]
        if sc is None:
            raise ValueError("pseq_constraints must be provided when pseq is provided.")
        tables = kernel_tables(fn.params["pseq"], sc)
        if found is not None and not all(np.array_equal(a, b) for a, b in zip(found, tables)):
            raise ValueError("stacking and hydrogen_bonding carry different probabilistic sequences: a system has one")
        found, terms = tables, terms | bit
    return None if found is None else (*found, terms)


def pseq_tensors(energy_fns):
    """(marginals, base-pair type probabilities) as differentiable functions of the ``pseq`` a term carries, if that
    pseq holds tensors that require a gradient (sequence design: d<U>/d(distribution)); None otherwise."""
    from mythos_amd.input.sequence_constraints import kernel_tables_torch

    for fn in energy_fns:
        if fn.term in ("stacking", "hydrogen_bonding") and "pseq" in fn.params and fn.params["pseq"] is not None:
            pseq = fn.params["pseq"]
            if any(isinstance(a, torch.Tensor) and a.requires_grad for a in pseq):
                return kernel_tables_torch(pseq, fn.params["pseq_constraints"])
            return None
    return None


def na1_flat_and_types(energy_fns, weights, geom, kt_default=None):
    """The three flat vectors (one tensor, oxDNA2 | oxRNA2 | hybrid) and ``is_rna`` of a composed oxNA function, plus
    the term weights and columns: what the energy path and the simulator both hand to an oxNA system."""
    from mythos_amd.energy import terms as _terms
    from mythos_amd.input.topology import NucleotideType

    sets = {which: {"geometry": geom.params[which]} if which in geom.params else {} for which in fp.NA1_SETS}
    term_w, cols = [0.0] * 8, []
    w_user = weights if weights is not None else torch.ones(len(energy_fns), dtype=torch.float64)
    kt = salt = hce = nt_type = None
    for fn, w in zip(energy_fns, w_user):
        k = TERM_ORDER.index(fn.term)
        if k in cols:
            raise ValueError(f"term '{fn.term}' appears twice in one composed energy function")
        for which, sec in fn.params.sections().items():
            sets[which][fn.term] = sec
        term_w[k] = float(w)
        cols.append(k)
        t = np.asarray(_np(fn.params["nt_type"]))
        if nt_type is not None and not np.array_equal(nt_type, t):
            raise ValueError("the terms of an oxNA energy function carry different nt_type arrays")
        nt_type = t
        if "kt" in fn.params and kt is None:
            kt = fn.params["kt"]
        if fn.term == "debye":
            salt, hce = fn.params["salt_conc"], bool(fn.params["half_charged_ends"])
    _terms.fill_missing_sections_na1(sets)
    if kt is None:
        kt = _terms.default_kt() if kt_default is None else kt_default
    named = fp.derive_flat_na1(sets["dna"], sets["rna"], sets["drh"], kt=kt, salt_conc=0.5 if salt is None else salt,
                               half_charged_ends=False if hce is None else hce, term_weights=term_w, numbers_ok=True)
    flat = fp.pack_flat_na1(named, _lib.param_names())
    if nt_type.shape != (int(_np(energy_fns[0].seq).shape[0]),):
        raise ValueError("nt_type must have one entry per nucleotide")
    return flat, nt_type == int(NucleotideType.RNA), term_w, cols


def _apply_pseq(entry, request) -> None:
    tag = None if request is None else (request[3], *(a.tobytes() for a in request[:3]))
    if entry["pseq"] != tag:
        if request is None:
            entry["sys"].set_pseq()
        else:
            entry["sys"].set_pseq(*request)
        entry["pseq"] = tag


# merged observable sets of energy functions built with_observables(...): (signature, n, dtype, device) -> ObservableSet
_OBS_SETS: dict = {}


def _fused_observables(observables, n: int, dtype, device):
    """One ObservableSet that serves as many of ``observables`` as share its lists: at most one propeller-twist list,
    one quartet list (with its geometry, model and displacement) and one skip_ends setting.  Returns
    (set, [the observables it serves]) or (None, [])."""
    from mythos_amd.observables.base import ObservableSet

    bp_obs = next((o for o in observables if getattr(o, "base_pairs", None) is not None), None)
    q_all = [o for o in observables if getattr(o, "quartets", None) is not None]
    q_obs, served = None, []
    if q_all:
        q_obs = q_all[0]
        ref = q_obs.signature()
        same = [o for o in q_all if o.signature()[1] == ref[1] and o.signature()[3:] == ref[3:]]
        skips = {bool(o.skip_ends) for o in same if type(o).__name__ == "PersistenceLength"}
        if len(skips) > 1:
            same = [o for o in same if type(o).__name__ != "PersistenceLength" or bool(o.skip_ends) == bool(same[0].skip_ends)]
        served += same
    if bp_obs is not None:
        served.append(bp_obs)
    if not served:
        return None, []
    skip = next((bool(o.skip_ends) for o in served if type(o).__name__ == "PersistenceLength"), False)
    geo = q_obs.geometry if q_obs is not None else None
    model = q_obs.model if q_obs is not None else 2
    box = getattr(q_obs.displacement_fn, "box", None) if q_obs is not None else None
    sig = (None if bp_obs is None else bp_obs.signature()[0], None if q_obs is None else q_obs.signature()[1], skip,
           None if q_obs is None else q_obs.signature()[3:], n, dtype, str(device))
    if sig not in _OBS_SETS:
        if len(_OBS_SETS) >= 8:
            _OBS_SETS.pop(next(iter(_OBS_SETS)))
        _OBS_SETS[sig] = ObservableSet(n, model, geo, box, None if bp_obs is None else bp_obs.base_pairs,
                                       None if q_obs is None else q_obs.quartets, skip, dtype, device)
    return _OBS_SETS[sig], served


class _EnergyOp(torch.autograd.Function):
    """(center, quat, flat) -> (weighted total per frame, raw term energies)."""

    @staticmethod
    def forward(ctx, center, quat, flat, entry, weights, marg=None, bp=None):
        """``marg`` (N, 4), ``bp`` (max(n_bp, 1), 4): the probabilistic sequence the system was given, as differentiable
        tensors - passed only when a gradient with respect to the distribution is wanted (the values the kernel reads
        are the ones ``_apply_pseq`` uploaded)."""
        system = entry["sys"]
        flat_np = flat.detach().cpu().to(torch.float64)
        tag = flat_np.numpy().tobytes()
        if entry["flat"] != tag:
            system.set_params(flat_np)
            entry["flat"] = tag
        need_x = center.requires_grad or quat.requires_grad
        need_p = flat.requires_grad
        need_s = marg is not None and (marg.requires_grad or bp.requires_grad)
        fuse = entry.get("observe")  # (ObservableSet, [observables]) for this call, set by _evaluate
        gm = gb = None
        if need_s:
            e, gc, gq, gp, gm, gb = system.energy(center.detach(), quat.detach(), grads=need_x, param_grads=True, pseq_grads=True)
            if not need_p:
                gp = None
        elif fuse is not None and center.dim() == 3:
            from mythos_amd.observables.base import remember_fused

            cd, qd = center.detach().contiguous(), quat.detach().contiguous()
            e, gc, gq, gp, rows = system.energy(cd, qd, grads=need_x, param_grads=need_p, observables=fuse[0])
            for o in fuse[1]:
                remember_fused(center, quat, o.signature(), rows)
        else:
            e, gc, gq, gp = system.energy(center.detach(), quat.detach(), grads=need_x, param_grads=need_p)
        single = e.dim() == 1
        if single:
            e = e[None]
        w = torch.as_tensor(weights, dtype=torch.float64, device=e.device)
        total = e @ w
        ctx.save_for_backward(*(t for t in (gc, gq, gp, gm, gb) if t is not None))
        ctx.flags = (need_x, need_p, single, flat.device, flat.dtype, need_s)
        ctx.mark_non_differentiable(e)
        return (total[0] if single else total), (e[0] if single else e)

    @staticmethod
    def backward(ctx, g_total, _g_terms):
        need_x, need_p, single, fdev, fdt, need_s = ctx.flags
        saved = list(ctx.saved_tensors)
        gc = gq = gf = g_marg = g_bp = None
        if need_x:
            dc_, dq_ = saved[0], saved[1]
            saved = saved[2:]
            scale = g_total.to(dc_.dtype)
            gc = dc_ * (scale if single else scale[:, None, None])
            gq = dq_ * (scale if single else scale[:, None, None])
        if need_p:
            dp = saved[0]
            gf = (dp * g_total.to(dp.dtype)) if single else (dp * g_total.to(dp.dtype)[:, None]).sum(0)
            gf = gf.to(device=fdev, dtype=fdt)
            saved = saved[1:]
        if need_s:
            dm, db = saved[0], saved[1]
            w = g_total.to(dm.dtype)
            g_marg = ((dm * w) if single else (dm * w[:, None, None]).sum(0)).cpu()
            g_bp = ((db * w) if single else (db * w[:, None, None]).sum(0)).cpu()
        return gc, gq, gf, None, None, g_marg, g_bp


# ---------------------------------------------------------------------------------------------
# protocol
# ---------------------------------------------------------------------------------------------
class EnergyFunction(ABC):
    """Callable body -> scalar energy (mythos/energy/base.py:24-93)."""

    map_batch_size: int | None = 100  # kept for signature parity; all frames go in one launch
    map_checkpoint: bool = True

    @abstractmethod
    def __call__(self, body: RigidBody) -> torch.Tensor: ...

    @abstractmethod
    def with_params(self, *repl_dicts: dict, **repl_kwargs: Any) -> "EnergyFunction": ...

    @abstractmethod
    def with_props(self, **kwargs) -> "EnergyFunction": ...

    @abstractmethod
    def with_noopt(self, *params: str) -> "EnergyFunction": ...

    @abstractmethod
    def params_dict(self, *, include_dependent: bool = True, exclude_non_optimizable: bool = False) -> dict: ...

    @abstractmethod
    def opt_params(self) -> dict: ...

    def map(self, body_sequence: RigidBody) -> torch.Tensor:
        """Energies of every frame, shape (n_states,) (base.py:90-93: lax.map over frames)."""
        return self(body_sequence)


def _is_probably_topology(obj) -> bool:
    return all(hasattr(obj, a) for a in ("seq", "bonded_neighbors", "unbonded_neighbors"))


class BaseEnergyFunction(EnergyFunction):
    """One term (mythos/energy/base.py:115-212).  Subclasses set ``term``, ``section`` and
    ``model`` and are constructed as in the reference:
    ``Fene(params=cfg.init_params(), displacement_fn=..., topology=..., transform_fn=...)``."""

    term: str = ""
    model: int = 1

    def __init__(self, *, params: BaseConfiguration, displacement_fn: Callable = DEFAULT_DISPLACEMENT, seq=None,
                 bonded_neighbors=None, unbonded_neighbors=None, topology=None, transform_fn: Geometry | None = None,
                 is_end=None):
        self.params = params
        self.displacement_fn = displacement_fn
        self.transform_fn = transform_fn
        if topology is not None:
            seq = topology.seq
            bonded_neighbors = topology.bonded_neighbors
            unbonded_neighbors = np.asarray(topology.unbonded_neighbors).T
            is_end = topology.is_end
        elif any(x is None for x in (seq, bonded_neighbors, unbonded_neighbors)):
            raise ValueError("Missing topology information")
        self.seq = seq
        self.bonded_neighbors = bonded_neighbors
        self.unbonded_neighbors = unbonded_neighbors
        self.is_end = is_end

    # -- functional updates -------------------------------------------------------------------------
    def replace(self, **kw) -> "BaseEnergyFunction":
        new = object.__new__(type(self))
        new.__dict__.update(self.__dict__)
        new.__dict__.update(kw)
        return new

    @classmethod
    def create_from(cls, other: "BaseEnergyFunction", **kwargs) -> "BaseEnergyFunction":
        new = object.__new__(cls)
        new.__dict__.update(other.__dict__)
        new.__dict__.update(kwargs)
        return new

    def with_props(self, **kwargs: Any) -> "BaseEnergyFunction":
        return self.replace(**kwargs)

    def with_noopt(self, *params: str) -> "BaseEnergyFunction":
        updated = set(self.params.non_optimizable_required_params) | set(params)
        return self.replace(params=self.params.replace(non_optimizable_required_params=tuple(sorted(updated))))

    def opt_params(self) -> dict:
        return self.params.opt_params

    def with_params(self, *repl_dicts: dict, **repl_kwargs: Any) -> "BaseEnergyFunction":
        new = self.params
        for d in repl_dicts:
            new = new | d
        new = new | repl_kwargs
        return self.replace(params=new.init_params())

    def params_dict(self, include_dependent: bool = True, exclude_non_optimizable: bool = False) -> dict:
        return self.params.to_dictionary(include_dependent=include_dependent, exclude_non_optimizable=exclude_non_optimizable)

    # -- algebra --------------------------------------------------------------------------------------
    def __add__(self, other):
        if not isinstance(other, BaseEnergyFunction):
            return NotImplemented
        return ComposedEnergyFunction(energy_fns=[self, other])

    def __mul__(self, other):
        if not isinstance(other, (float, int)):
            return NotImplemented
        return ComposedEnergyFunction(energy_fns=[self], weights=torch.tensor([float(other)], dtype=torch.float64))

    # -- evaluation -----------------------------------------------------------------------------------
    def __call__(self, body: RigidBody) -> torch.Tensor:
        return ComposedEnergyFunction(energy_fns=[self])(body)

    def compute_energy(self, nucleotide: RigidBody) -> torch.Tensor:
        return self(nucleotide)


class ComposedEnergyFunction(EnergyFunction):
    """Linear combination of terms with a shared parameter namespace (base.py:215-434)."""

    def __init__(self, energy_fns: list, weights=None, strict_params: bool = True):
        if not isinstance(energy_fns, list) or not all(isinstance(f, BaseEnergyFunction) for f in energy_fns):
            raise TypeError(ERR_COMPOSED_ENERGY_FN_TYPE_ENERGY_FNS)
        if weights is not None and len(weights) != len(energy_fns):
            raise ValueError(ERR_COMPOSED_ENERGY_FN_LEN_MISMATCH)
        self.energy_fns = energy_fns
        self.weights = None if weights is None else torch.as_tensor(weights, dtype=torch.float64)
        self.strict_params = strict_params

    def replace(self, **kw) -> "ComposedEnergyFunction":
        new = object.__new__(type(self))
        new.__dict__.update(self.__dict__)
        new.__dict__.update(kw)
        return new

    # hooks the qualified variant overrides
    def _param_in_fn(self, param: str, fn: BaseEnergyFunction) -> bool:
        return param in fn.params

    def _rename_param_for_fn(self, param: str, _fn) -> str:
        return param

    def _rename_param_from_fn(self, param: str, _fn) -> str:
        return param

    def with_props(self, **kwargs: Any) -> "ComposedEnergyFunction":
        return self.replace(energy_fns=[fn.with_props(**kwargs) for fn in self.energy_fns])

    def with_observables(self, *observables) -> "ComposedEnergyFunction":
        """The same energy function whose ``map`` / ``__call__`` on a batch of frames also evaluates these structural
        observables in the same call (mythos_oxdna_energy_obs: the observables kernel queued behind the energy launch): a later ``observable(trajectory)`` on the
        same frames returns those rows instead of launching again.  What DiffTRe needs per iteration - energies,
        dU/dtheta, the observable - then costs one read of the stored trajectory.  (No counterpart in the reference,
        whose observables are separate jitted functions over the trajectory, mythos/observables/*.py.)"""
        return self.replace(observables=tuple(observables))

    def with_noopt(self, *params: str) -> "ComposedEnergyFunction":
        fns = []
        for fn in self.energy_fns:
            mine = [self._rename_param_for_fn(p, fn) for p in params if self._param_in_fn(p, fn)]
            fns.append(fn.with_noopt(*mine))
        return self.replace(energy_fns=fns)

    def opt_params(self, from_fns: list | None = None) -> dict:
        fns = self.energy_fns if from_fns is None else [f for f in self.energy_fns if type(f) in from_fns]
        return {self._rename_param_from_fn(k, fn): v for fn in fns for k, v in fn.opt_params().items()}

    def with_params(self, *repl_dicts: dict, **repl_kwargs: Any) -> "ComposedEnergyFunction":
        all_repl = set(repl_kwargs) | {k for d in repl_dicts for k in d}
        used = set()
        fns = []
        for fn in self.energy_fns:
            new = {k: v for d in repl_dicts for k, v in d.items() if self._param_in_fn(k, fn)}
            new.update({k: v for k, v in repl_kwargs.items() if self._param_in_fn(k, fn)})
            used.update(new.keys())
            new = {self._rename_param_for_fn(k, fn): v for k, v in new.items()}
            fns.append(fn.with_params(**new))
        if self.strict_params and (unused := all_repl - used):
            raise ValueError(f"Some parameters were not used in any energy function: {unused}.")
        return self.replace(energy_fns=fns)

    def params_dict(self, *, include_dependent: bool = True, exclude_non_optimizable: bool = False) -> dict:
        out = {}
        for fn in self.energy_fns:
            d = fn.params_dict(include_dependent=include_dependent, exclude_non_optimizable=exclude_non_optimizable)
            out.update({self._rename_param_from_fn(k, fn): v for k, v in d.items()})
        return out

    def without_terms(self, *terms) -> "ComposedEnergyFunction":
        keep, w = [], []
        for i, fn in enumerate(self.energy_fns):
            if type(fn) in terms or type(fn).__name__ in terms:
                continue
            keep.append(fn)
            if self.weights is not None:
                w.append(self.weights[i])
        return self.replace(energy_fns=keep, weights=None if self.weights is None else torch.stack(w))

    def add_energy_fn(self, energy_fn: BaseEnergyFunction, weight: float = 1.0) -> "ComposedEnergyFunction":
        if self.weights is None:
            weights = None if weight == 1.0 else torch.tensor([1.0] * len(self.energy_fns) + [weight], dtype=torch.float64)
        else:
            weights = torch.cat([self.weights, torch.tensor([weight], dtype=torch.float64)])
        return ComposedEnergyFunction(energy_fns=[*self.energy_fns, energy_fn], weights=weights)

    def add_composable_energy_fn(self, other: "ComposedEnergyFunction") -> "ComposedEnergyFunction":
        if self.weights is None and other.weights is None:
            weights = None
        else:
            a = self.weights if self.weights is not None else torch.ones(len(self.energy_fns), dtype=torch.float64)
            b = other.weights if other.weights is not None else torch.ones(len(other.energy_fns), dtype=torch.float64)
            weights = torch.cat([a, b])
        return ComposedEnergyFunction(energy_fns=self.energy_fns + other.energy_fns, weights=weights)

    def __add__(self, other):
        if isinstance(other, BaseEnergyFunction):
            return self.add_energy_fn(other)
        if isinstance(other, ComposedEnergyFunction):
            return self.add_composable_energy_fn(other)
        return NotImplemented

    __radd__ = __add__

    @classmethod
    def from_lists(cls, energy_fns: list, energy_configs: list, weights=None, **kwargs) -> "ComposedEnergyFunction":
        weights = weights if weights is not None else torch.ones(len(energy_fns), dtype=torch.float64)
        if len(energy_fns) != len(energy_configs):
            raise ValueError("energy_fns and energy_configs differ in length")
        return cls(energy_fns=[ef(**kwargs, params=ec.init_params()) for ef, ec in zip(energy_fns, energy_configs)], weights=weights)

    # -- evaluation -----------------------------------------------------------------------------------
    def _evaluate(self, body: RigidBody):
        """One launch: (weighted total per frame, raw (.., 8) term energies, per-fn term columns)."""
        if not self.energy_fns:
            raise ValueError("ComposedEnergyFunction has no energy functions")
        first = self.energy_fns[0]
        geom = next((fn.transform_fn for fn in self.energy_fns if fn.transform_fn is not None), None)
        if geom is None:
            raise ValueError("transform_fn (site geometry) must be provided")
        model = geom.model  # the site geometry decides oxDNA1 vs oxDNA2 (shared term classes exist in both)
        from mythos_amd.energy import terms as _terms

        _terms.check_term_models(model, self.energy_fns)
        if model == 4:
            return self._evaluate_na1(body, geom)
        sections = {"geometry": geom.params}
        term_w = [0.0] * 8
        cols = []
        w_user = self.weights if self.weights is not None else torch.ones(len(self.energy_fns), dtype=torch.float64)

        for fn, w in zip(self.energy_fns, w_user):
            k = TERM_ORDER.index(fn.term)
            if fn.term in sections:
                raise ValueError(f"term '{fn.term}' appears twice in one composed energy function")
            sections[fn.term] = {n: fn.params[n] for n in (*type(fn.params).required_params, *type(fn.params).optional_params)
                                 if n not in ("pseq", "pseq_constraints")}
            term_w[k] = float(w)
            cols.append(k)
        kt = salt = hce = None
        for fn in self.energy_fns:
            if "kt" in fn.params:
                kt = fn.params["kt"] if kt is None else kt
            if fn.term == "debye":
                salt, hce = fn.params["salt_conc"], bool(fn.params["half_charged_ends"])
        _terms.fill_missing_sections(model, sections)
        if kt is None:
            kt = _terms.default_kt()
        flat_named = fp.derive_flat(model, sections, kt=kt, salt_conc=0.5 if salt is None else salt,
                                    half_charged_ends=True if hce is None else hce, term_weights=term_w, numbers_ok=True)
        flat = fp.pack_flat(flat_named, _lib.param_names())
        center, quat = body.center, body.orientation.vec
        box = getattr(first.displacement_fn, "box", None)
        entry = _get_system(model, _np(first.seq), _np(first.is_end) if first.is_end is not None else None,
                            _np(first.bonded_neighbors), first.unbonded_neighbors, box, center.dtype, center.device)
        _apply_pseq(entry, pseq_request(self.energy_fns))
        entry["observe"] = None
        if getattr(self, "observables", None) and center.dim() == 3 and center.device.type == "cuda":
            oset, served = _fused_observables(self.observables, int(center.shape[1]), center.dtype, center.device)
            entry["observe"] = None if oset is None else (oset, served)
        pseq_leaves = pseq_tensors(self.energy_fns)
        try:
            if pseq_leaves is not None:
                entry["observe"] = None  # the distribution gradient comes from its own entry point, without the observables
                total, terms = _EnergyOp.apply(center, quat, flat, entry, term_w, *pseq_leaves)
            else:
                total, terms = _EnergyOp.apply(center, quat, flat, entry, term_w)
        finally:
            entry["observe"] = None
        return total, terms, cols

    def _evaluate_na1(self, body: RigidBody, geom):
        """oxNA (mythos/energy/na1/): every term carries three parameter sets and the types of the nucleotides; the
        kernels take three flat vectors - oxDNA2, oxRNA2, hybrid - and ``is_rna``."""
        first = self.energy_fns[0]
        flat, is_rna, term_w, cols = na1_flat_and_types(self.energy_fns, self.weights, geom)
        center, quat = body.center, body.orientation.vec
        entry = _get_system(4, _np(first.seq), _np(first.is_end) if first.is_end is not None else None, _np(first.bonded_neighbors),
                            first.unbonded_neighbors, getattr(first.displacement_fn, "box", None), center.dtype, center.device,
                            is_rna=is_rna)
        _apply_pseq(entry, pseq_request(self.energy_fns))  # hydrogen bonding only (na1/hydrogen_bonding.py:127-128)
        entry["observe"] = None
        pseq_leaves = pseq_tensors(self.energy_fns)  # d<U>/d(distribution) wanted: mythos_oxdna_energy_dpseq
        if pseq_leaves is not None:
            total, terms = _EnergyOp.apply(center, quat, flat, entry, term_w, *pseq_leaves)
        else:
            total, terms = _EnergyOp.apply(center, quat, flat, entry, term_w)
        return total, terms, cols

    def compute_terms(self, body: RigidBody) -> torch.Tensor:
        """Energy of each composed function, shape (n_fns,) or (n_states, n_fns) (base.py:312-314)."""
        _, terms, cols = self._evaluate(body)
        return terms[..., cols]

    def __call__(self, body: RigidBody) -> torch.Tensor:
        total, _, _ = self._evaluate(body)
        return total


class QualifiedComposedEnergyFunction(ComposedEnergyFunction):
    """Parameters addressed as ``ClassName.param`` (base.py:437-462)."""

    def _param_in_fn(self, param: str, fn: BaseEnergyFunction) -> bool:
        cls, name = param.split(".", 1)
        return name in fn.params and type(fn).__qualname__ == cls

    def _rename_param_for_fn(self, param: str, fn) -> str:
        return param.split(".", 1)[1]

    def _rename_param_from_fn(self, param: str, fn) -> str:
        return f"{type(fn).__qualname__}.{param}"


def _np(x):
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy()
    return np.asarray(x)
