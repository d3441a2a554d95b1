"""Single-process optimisation loop (mythos/optimization/optimization.py:37-150, 340-398) and Adam.

The reference fans simulators and objectives out as Ray tasks (``RayOptimizer``, out of scope here: replicas are
sharded one process per GPU with ``mythos_amd.distributed`` instead); ``SimpleOptimizer`` is its in-process form:
one simulator, one objective, a gradient transformation.  The optimiser is optax-shaped
(``init(params) -> state``, ``update(grads, state, params) -> (updates, state)``); :class:`Adam` follows
``optax.adam`` (bias-corrected moments, update = -lr m_hat / (sqrt(v_hat) + eps))."""

from __future__ import annotations

import dataclasses as dc
import math
from typing import Any, Callable

import torch

from mythos_amd.optimization.objective import Objective
from mythos_amd.simulators.base import Simulator


@dc.dataclass(frozen=True, kw_only=True)
class OptimizerState:
    observables: dict = dc.field(default_factory=dict)
    component_state: dict = dc.field(default_factory=dict)
    optimizer_state: Any = None


@dc.dataclass(frozen=True, kw_only=True)
class OptimizerOutput:
    grads: dict
    opt_params: dict
    state: OptimizerState
    observables: dict = dc.field(default_factory=dict)


@dc.dataclass(frozen=True)
class Adam:
    learning_rate: float
    b1: float = 0.9
    b2: float = 0.999
    eps: float = 1e-8

    def init(self, params: dict):
        zeros = {k: torch.zeros_like(torch.as_tensor(v, dtype=torch.float64)) for k, v in params.items()}
        return {"count": 0, "mu": zeros, "nu": {k: v.clone() for k, v in zeros.items()}}

    def update(self, grads: dict, state, params: dict | None = None):  # noqa: ARG002
        t = state["count"] + 1
        mu = {k: self.b1 * state["mu"][k] + (1 - self.b1) * torch.as_tensor(g, dtype=torch.float64).cpu() for k, g in grads.items()}
        nu = {k: self.b2 * state["nu"][k] + (1 - self.b2) * torch.as_tensor(g, dtype=torch.float64).cpu() ** 2 for k, g in grads.items()}
        c1, c2 = 1 - self.b1**t, 1 - self.b2**t
        updates = {k: -self.learning_rate * (mu[k] / c1) / (torch.sqrt(nu[k] / c2) + self.eps) for k in grads}
        return updates, {"count": t, "mu": mu, "nu": nu}


def apply_updates(params: dict, updates: dict) -> dict:
    return {k: torch.as_tensor(v, dtype=torch.float64).cpu() + updates[k] for k, v in params.items()}


@dc.dataclass(frozen=True, kw_only=True)
class Optimizer:
    def step(self, params: dict, state: OptimizerState | None = None) -> OptimizerOutput:
        raise NotImplementedError

    def run(self, params: dict, n_steps: int, callback: Callable | None = None) -> OptimizerOutput:
        """``callback(optimizer_output=, step=) -> (replacement or None, keep_going)``; non-finite gradients raise."""
        if n_steps < 1:
            raise ValueError("n_steps must be at least 1.")
        state = None
        for step in range(n_steps):
            output = self.step(params, state)
            keep_going = True
            if callback is not None:
                replacement, keep_going = callback(optimizer_output=output, step=step)
                output = replacement if replacement is not None else output
            if not keep_going:
                break
            if any(not bool(torch.isfinite(torch.as_tensor(g)).all()) for g in output.grads.values()):
                raise RuntimeError(f"NaN or Inf detected in gradients at step {step}. Is your learning rate too high?")
            params, state = output.opt_params, output.state
        return output


@dc.dataclass(frozen=True, kw_only=True)
class SimpleOptimizer(Optimizer):
    objective: Objective
    simulator: Simulator
    optimizer: Any

    def step(self, params: dict, state: OptimizerState | None = None) -> OptimizerOutput:
        state = state or OptimizerState()
        obj_state = state.component_state.get(self.objective.name, {})
        sim_state = state.component_state.get(self.simulator.name, {})
        out = None
        if state.observables:
            out = self.objective.calculate(state.observables, opt_params=params, **obj_state)
            obj_state = out.state
        if out is None or not out.is_ready:
            sim_out = self.simulator.run(params, **sim_state)
            sim_state = sim_out.state
            state = dc.replace(state, observables=dict(zip(self.simulator.exposes(), sim_out.observables)))
            out = self.objective.calculate(state.observables, opt_params=params, **obj_state)
            obj_state = out.state
            if not out.is_ready:
                raise ValueError("Objective readiness check failed after simulation run.")
        opt_state = state.optimizer_state or self.optimizer.init(params)
        updates, opt_state = self.optimizer.update(out.grads, opt_state, params)
        from mythos_amd.observables.base import clear_fused

        clear_fused()  # rows remembered beside this iteration's energy launches (and the trajectories they pin) are done with
        new_state = dc.replace(state, optimizer_state=opt_state, component_state={
            **state.component_state, self.objective.name: obj_state, self.simulator.name: sim_state})
        return OptimizerOutput(grads=out.grads, opt_params=apply_updates(params, updates), state=new_state,
                               observables={self.objective.name: out.observables})


__all__ = ["Adam", "Optimizer", "OptimizerOutput", "OptimizerState", "SimpleOptimizer", "apply_updates", "math"]
