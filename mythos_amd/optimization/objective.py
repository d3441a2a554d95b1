"""DiffTRe reweighting math (mythos/optimization/objective.py:139-235) on torch tensors, plus the
replica-sharded form that combines per-GPU partial sums with RCCL all-reduces.

``compute_loss_and_grad`` is the counterpart of ``jax.value_and_grad(compute_loss, has_aux=True)``
(:235): the energies of the stored frames are differentiated with respect to the optimisation
parameters by the HIP kernels (dU/dparam) and ``torch.autograd`` carries the chain rule through the
softmax weights and the loss.
"""

from __future__ import annotations

from typing import Any, Callable

import torch


def compute_weights_and_neff(beta, new_energies, ref_energies):
    """w = softmax(-beta (E_new - E_ref)), n_eff = exp(-sum w ln w) / n   (:139-163)."""
    new_energies = torch.as_tensor(new_energies, dtype=torch.float64)
    ref_energies = torch.as_tensor(ref_energies, dtype=torch.float64, device=new_energies.device)
    beta = torch.as_tensor(beta, dtype=torch.float64, device=new_energies.device)
    diffs = new_energies - ref_energies
    boltz = torch.exp(-beta * diffs)
    weights = boltz / torch.sum(boltz)
    n_eff = torch.exp(-torch.sum(weights * torch.log(weights)))
    return weights, n_eff / weights.shape[0]


def compute_min_segment_neff(temperature, new_energies, ref_energies) -> float:
    """Minimum normalised n_eff over temperature segments (:166-195)."""
    temperature = torch.as_tensor(temperature)
    out = []
    for t in torch.unique(temperature):
        mask = temperature == t
        _, neff = compute_weights_and_neff(1.0 / t, new_energies[mask], ref_energies[mask])
        out.append(float(neff))
    return min(out)


def compute_loss(opt_params: dict, energy_fn, beta, loss_fn: Callable, ref_states, ref_energies, observables: list):
    """(:198-232) -> loss, (neff, measured_value, new_energies)."""
    energy_fn = energy_fn.with_params(opt_params)
    new_energies = energy_fn.map(ref_states)
    weights, neff = compute_weights_and_neff(beta, new_energies, ref_energies)
    loss, (measured_value, _) = loss_fn(ref_states, weights, energy_fn, opt_params, observables)
    return loss, (neff, measured_value, new_energies)


def compute_loss_and_grad(opt_params: dict, energy_fn, beta, loss_fn, ref_states, ref_energies, observables):
    """((loss, aux), grads) like jax.value_and_grad(compute_loss, has_aux=True) (:235)."""
    leaves = {k: torch.as_tensor(v, dtype=torch.float64).detach().clone().requires_grad_(True) for k, v in opt_params.items()}
    loss, aux = compute_loss(leaves, energy_fn, beta, loss_fn, ref_states, ref_energies, observables)
    grads = torch.autograd.grad(loss, list(leaves.values()), allow_unused=True)
    gdict = {k: (torch.zeros_like(leaves[k]) if g is None else g) for k, g in zip(leaves, grads)}
    return (loss.detach(), tuple(a.detach() if isinstance(a, torch.Tensor) else a for a in aux)), gdict


# ------------------------------------------------------------------------------------------------
# replica-sharded reweighting: every rank holds the frames of its own replicas
# ------------------------------------------------------------------------------------------------
def _reduce(t, op, group):
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=op, group=group)
    return t


def distributed_weights_and_neff(beta, new_energies, ref_energies, group=None):
    """Globally normalised weights of THIS rank's frames and the global n_eff.

    Two collectives: all-reduce(MAX) of the exponent for a stable softmax, then
    all-reduce(SUM) of [sum exp, sum exp * x, count] from which Z and the entropy follow.

    The weights come back DETACHED: their normaliser was all-reduced, so autograd through them would see the
    numerator only and silently drop the -<O><g> term of the DiffTRe gradient.  Gradients of reweighted means across
    ranks come from ``distributed_reweighted_mean_and_grad`` / ``distributed_compute_loss_and_grad`` (closed form,
    every sum all-reduced before the means are combined).
    """
    import torch.distributed as dist

    with torch.no_grad():
        x = -torch.as_tensor(beta, dtype=torch.float64, device=new_energies.device) * (
            new_energies.to(torch.float64) - torch.as_tensor(ref_energies, dtype=torch.float64, device=new_energies.device)
        )
        m = x.max().reshape(1) if x.numel() else torch.full((1,), -float("inf"), dtype=torch.float64, device=x.device)
        _reduce(m, dist.ReduceOp.MAX, group)
        ex = torch.exp(x - m)
        sums = torch.stack([ex.sum(), (ex * (x - m)).sum(), torch.tensor(float(x.numel()), dtype=torch.float64, device=x.device)])
        _reduce(sums, dist.ReduceOp.SUM, group)
        z, sx, n = sums[0], sums[1], sums[2]
        weights = ex / z
        entropy = torch.log(z) - sx / z  # -sum w ln w over ALL frames
        return weights, torch.exp(entropy) / n


def reweighted_mean_and_grad(observable, energies_grad, weights, beta):
    """<O>_w and d<O>_w/dtheta = -beta ( <O g>_w - <O>_w <g>_w ) for per-frame dU/dtheta rows g (S, K)
    (closed form of the DiffTRe gradient, SURVEY.md section 8a last row).  ``weights`` must be normalised over the
    frames passed in: this is the single-process form (see ``distributed_reweighted_mean_and_grad`` across ranks)."""
    w = weights.to(torch.float64)
    o = observable.to(torch.float64)
    g = energies_grad.to(torch.float64)
    mean_o = (w * o).sum()
    mean_g = (w[:, None] * g).sum(0)
    mean_og = ((w * o)[:, None] * g).sum(0)
    b = torch.as_tensor(beta, dtype=torch.float64, device=w.device)
    return mean_o, -b * (mean_og - mean_o * mean_g)


def distributed_reweighted_mean_and_grad(observable, weighted_grad_fn, beta, new_energies, ref_energies, group=None):
    """<O>_w, d<O>_w/dtheta and n_eff over the frames of ALL ranks.

    ``observable`` (S_local,), ``new_energies`` / ``ref_energies`` (S_local,), ``beta`` scalar or (S_local,);
    ``weighted_grad_fn(c)`` returns sum_s c_s dU(x_s)/dtheta as a (K,) tensor for per-frame coefficients c (one
    vector-Jacobian product over this rank's frames: the HIP kernel's dU/dtheta rows times c).
    Collectives: all-reduce(MAX) of the softmax exponent, then ONE all-reduce(SUM) of
    [sum w~, sum w~ O, sum w~ ln w~, count, sum w~ beta g (K), sum w~ O beta g (K)]  (SURVEY.md 8e), w~ = exp(x - max).
    Means are formed only after the reduction:  d<O>/dtheta = -( <O beta g> - <O> <beta g> ).
    """
    import torch.distributed as dist

    with torch.no_grad():
        dev = new_energies.device
        b = torch.as_tensor(beta, dtype=torch.float64, device=dev).expand(new_energies.shape)
        x = -b * (new_energies.to(torch.float64) - torch.as_tensor(ref_energies, dtype=torch.float64, device=dev))
        m = x.max().reshape(1) if x.numel() else torch.full((1,), -float("inf"), dtype=torch.float64, device=dev)
        _reduce(m, dist.ReduceOp.MAX, group)
        wt = torch.exp(x - m)
        o = observable.to(torch.float64).to(dev)
    g1 = weighted_grad_fn(wt * b).detach().to(torch.float64).reshape(-1)
    g2 = weighted_grad_fn(wt * b * o).detach().to(torch.float64).reshape(-1)
    with torch.no_grad():
        k = g1.numel()
        head = torch.stack([wt.sum(), (wt * o).sum(), (wt * (x - m)).sum(), torch.tensor(float(x.numel()), dtype=torch.float64, device=dev)])
        buf = torch.cat([head, g1.to(dev), g2.to(dev)])
        _reduce(buf, dist.ReduceOp.SUM, group)
        z, so, sx, n = buf[0], buf[1], buf[2], buf[3]
        mean_o = so / z
        mean_g = buf[4 : 4 + k] / z
        mean_og = buf[4 + k : 4 + 2 * k] / z
        neff = torch.exp(torch.log(z) - sx / z) / n
        return mean_o, -(mean_og - mean_o * mean_g), neff


def distributed_compute_loss_and_grad(opt_params: dict, energy_fn, beta, observable_fn: Callable, loss_of_mean: Callable,
                                      ref_states, ref_energies, group=None):
    """The multi-rank counterpart of ``compute_loss_and_grad`` for losses of ONE reweighted mean, L(<O>_w): every rank
    passes the stored frames of its own replicas (replica r lives on rank r mod world, mythos_amd.distributed), the
    result - loss, gradients, n_eff, <O>_w - is identical on all ranks and equal to the single-process value on the
    concatenated frames.  Replaces the gather of whole trajectories into one objective task that the reference does
    through Ray's object store (mythos/optimization/optimization.py:151-169, 225-247, objective.py:277-389).

    ``observable_fn(ref_states) -> (S_local,)``; ``loss_of_mean(mean) -> scalar`` differentiable torch function.
    Returns ((loss, (neff, mean_o, new_energies_local)), grads) like ``compute_loss_and_grad``.
    """
    names = list(opt_params)
    leaves = {k: torch.as_tensor(opt_params[k], dtype=torch.float64).detach().clone().requires_grad_(True) for k in names}
    new_energies = energy_fn.with_params(leaves).map(ref_states)

    def weighted_grad(c):
        gs = torch.autograd.grad(new_energies, [leaves[k] for k in names], grad_outputs=c.to(new_energies), retain_graph=True,
                                 allow_unused=True)
        return torch.stack([torch.zeros((), dtype=torch.float64, device=new_energies.device) if g is None else g.to(new_energies.device).reshape(())
                            for g in gs])

    obs = observable_fn(ref_states).detach()
    mean_o, dmean, neff = distributed_reweighted_mean_and_grad(obs, weighted_grad, beta, new_energies.detach(), ref_energies, group)
    mean_leaf = mean_o.detach().clone().requires_grad_(True)
    loss = loss_of_mean(mean_leaf)
    (dl,) = torch.autograd.grad(loss, [mean_leaf])
    grads = {k: (dl * dmean[i]).detach() for i, k in enumerate(names)}
    return (loss.detach(), (neff, mean_o, new_energies.detach())), grads


# ------------------------------------------------------------------------------------------------
# objective protocol (mythos/optimization/objective.py:33-136, 239-389)
# ------------------------------------------------------------------------------------------------
import dataclasses as _dc
import math as _math


@_dc.dataclass(frozen=True, kw_only=True)
class ObjectiveOutput:
    """What ``Objective.calculate`` hands back (objective.py:33-50): gradients when ready, otherwise the names of
    the observables that have to be produced again; ``state`` travels to the next call."""

    is_ready: bool
    grads: dict | None = None
    observables: dict = _dc.field(default_factory=dict)
    state: dict = _dc.field(default_factory=dict)
    needs_update: tuple = ()


@_dc.dataclass(frozen=True, kw_only=True)
class Objective:
    """Immutable objective: ``grad_or_loss_fn(*observables) -> (grads, [(name, value), ...])`` (objective.py:53-136)."""

    name: str
    required_observables: tuple
    grad_or_loss_fn: Callable = _dc.field(repr=False)
    logging_observables: tuple = ()

    def __post_init__(self):
        for field in ("name", "required_observables", "grad_or_loss_fn"):
            if getattr(self, field) is None:
                raise ValueError(f"Missing required argument: {field}.")

    def _missing(self, observables) -> tuple:
        return tuple(o for o in self.required_observables if o not in observables)

    def calculate(self, observables: dict, opt_params=None, **_kwargs) -> ObjectiveOutput:  # noqa: ARG002
        missing = self._missing(observables)
        if missing:
            return ObjectiveOutput(is_ready=False, needs_update=missing)
        ordered = [observables[k] for k in self.required_observables]
        grads, aux = self.grad_or_loss_fn(*ordered)
        out = dict(aux)
        out.update(dict(zip(self.required_observables, ordered)))
        return ObjectiveOutput(is_ready=True, grads=grads, observables=out)

    def get_logging_observables(self, observables: dict) -> list:
        return [(n, observables[n]) for n in self.logging_observables if n in observables]


@_dc.dataclass(frozen=True, kw_only=True)
class DiffTReObjective(Objective):
    """Gradients by trajectory reweighting (objective.py:239-389).

    ``grad_or_loss_fn(ref_states, weights, energy_fn, opt_params, observables) -> (loss, ((name, value), aux))``.
    The stored trajectory stays valid while the normalised effective sample size of the reweighting from
    ``reference_opt_params`` to ``opt_params`` is at least ``min_n_eff_factor`` (minimum over temperature segments)
    and fewer than ``max_valid_opt_steps`` optimisation steps used it; otherwise new trajectories are requested.
    Energies and dU/dtheta of all stored frames come from one launch of the HIP energy kernel per evaluation.
    """

    energy_fn: Any = _dc.field(repr=False)
    n_equilibration_steps: int = 0
    min_n_eff_factor: float = 0.95
    max_valid_opt_steps: float = _math.inf

    def __post_init__(self):
        Objective.__post_init__(self)
        if self.energy_fn is None:
            raise ValueError("Missing required argument: energy_fn.")
        if self.n_equilibration_steps is None or self.n_equilibration_steps < 0:
            raise ValueError(f"n_equilibration_steps must be non-negative, got {self.n_equilibration_steps}.")
        if self.max_valid_opt_steps <= 0:
            raise ValueError("max_valid_opt_steps must be positive or infinity.")

    def calculate(self, observables: dict, opt_params: dict, opt_steps: int = 0, reference_opt_params: dict | None = None,
                  **_kwargs) -> ObjectiveOutput:
        from mythos_amd.simulators.io import SimulatorTrajectory

        if opt_steps >= self.max_valid_opt_steps:
            return ObjectiveOutput(is_ready=False, needs_update=tuple(self.required_observables), state={"opt_steps": 0})
        missing = self._missing(observables)
        if missing:
            return ObjectiveOutput(is_ready=False, needs_update=missing)
        ordered = [observables[k] for k in self.required_observables]
        trajectories = [o for o in ordered if isinstance(o, SimulatorTrajectory)]
        if not trajectories:
            raise ValueError("No SimulatorTrajectory observables found in observables.")
        if self.n_equilibration_steps > 0:
            trajectories = [t.slice(slice(self.n_equilibration_steps, t.length(), None)) for t in trajectories]
        ref_states = SimulatorTrajectory.concat(trajectories)
        if ref_states.length() == 0:
            raise ValueError("Equilibration slicing yields no states! Note slicing is in number of snapshots, not timesteps.")
        if ref_states.temperature is None:
            raise ValueError("SimulatorTrajectory.temperature is None. DiffTRe requires per-state temperature (kT) on the trajectory.")
        beta = 1.0 / torch.as_tensor(ref_states.temperature, dtype=torch.float64)
        reference_opt_params = reference_opt_params or opt_params
        with torch.no_grad():
            ref_energies = self.energy_fn.with_params(reference_opt_params).map(ref_states).detach()
            new_energies = self.energy_fn.with_params(opt_params).map(ref_states).detach()
        neff = compute_min_segment_neff(ref_states.temperature, new_energies, ref_energies)
        if neff < self.min_n_eff_factor:
            return ObjectiveOutput(is_ready=False, needs_update=tuple(self.required_observables), observables={"neff": neff},
                                   state={"opt_steps": 0})
        (loss, (_, measured, _)), grads = compute_loss_and_grad(
            opt_params, self.energy_fn, beta.to(ref_energies.device), self.grad_or_loss_fn, ref_states, ref_energies, ordered)
        from mythos_amd.observables.base import clear_fused

        clear_fused()  # (observable rows computed beside the energy launches above: the loss has used them)
        return ObjectiveOutput(
            is_ready=True, grads=grads, observables={"loss": loss, "neff": neff, measured[0]: measured[1]},
            state={"opt_steps": opt_steps + 1, "reference_opt_params": reference_opt_params})
