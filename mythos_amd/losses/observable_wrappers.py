"""Loss functions for observables: the glue between a (reweighted) observable and a DiffTRe objective.

Mirror of the reference's ``mythos/losses/observable_wrappers.py:16-63`` on torch tensors - same classes, same call
signatures, same return shapes (``ObservableLossFn`` returns a tuple, with the observable appended on request) - so a
loss written for the reference drops into ``mythos_amd.optimization.objective.DiffTReObjective``: the weighted sum runs
over the per-state values an observable (``mythos_amd.observables``: HIP kernels) hands back, and autograd carries the
loss to the weights, i.e. to the kernel's dU/dtheta."""

from __future__ import annotations

import dataclasses as dc
from typing import Any

import torch


def _t(x) -> torch.Tensor:
    return x if isinstance(x, torch.Tensor) else torch.as_tensor(x)


@dc.dataclass
class LossFn:
    """Base class for loss functions."""

    def __call__(self, actual, target, weights=None):
        raise NotImplementedError("Subclasses must implement this method.")


@dc.dataclass
class SquaredError(LossFn):
    """(target - actual)^2, element-wise."""

    def __call__(self, actual, target, weights=None):
        return (_t(target) - _t(actual)) ** 2


@dc.dataclass
class RootMeanSquaredError(LossFn):
    """sqrt(mean((target - actual)^2))."""

    def __call__(self, actual, target, weights=None):
        d = (_t(target) - _t(actual)).to(torch.float64)
        return torch.sqrt(torch.mean(d**2))


@dc.dataclass
class ObservableLossFn:
    """``loss_fn(sum(observable(trajectory) * weights), target)`` - a tuple, with the observable appended if
    ``return_observable`` (mythos/losses/observable_wrappers.py:43-58)."""

    observable: Any
    loss_fn: Any
    return_observable: bool = False

    def __call__(self, trajectory, target, weights):
        vals_per_state = _t(self.observable(trajectory))
        w = _t(weights).to(device=vals_per_state.device)
        observable = torch.sum(vals_per_state * w)
        vals = [self.loss_fn(observable, target)]
        if self.return_observable:
            vals.append(observable)
        return tuple(vals)


def l2_loss(actual, target):
    """sum((actual - target)^2)."""
    return torch.sum((_t(actual) - _t(target)) ** 2)
