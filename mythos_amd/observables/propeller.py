"""Propeller twist (mythos/observables/propeller.py:19-71): per frame, the mean over the listed H-bonded base
pairs of 180 - acos(n_i . n_j) in degrees, n = base normal a3.  Evaluated with torch on the device the
trajectory lives on; a handful of flops per base pair, not a kernel."""

from __future__ import annotations

import math

import torch

TARGETS = {"oxDNA": 21.7}  # degrees (propeller.py:13-15)


class PropellerTwist:
    def __init__(self, h_bonded_base_pairs):
        self.h_bonded_base_pairs = torch.as_tensor(h_bonded_base_pairs, dtype=torch.long).reshape(-1, 2)

    def __call__(self, trajectory) -> torch.Tensor:
        """(n_states,) propeller twist in degrees."""
        q = trajectory.orientation.vec
        q0, q1, q2, q3 = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
        # base normal a3(q) (mythos/energy/utils.py:26-30)
        a3 = torch.stack([2 * (q1 * q3 + q0 * q2), 2 * (q2 * q3 - q0 * q1), q0 * q0 - q1 * q1 - q2 * q2 + q3 * q3], dim=-1)
        bp = self.h_bonded_base_pairs.to(a3.device)
        dots = (a3[..., bp[:, 0], :] * a3[..., bp[:, 1], :]).sum(-1).clamp(-1.0, 1.0)
        return (180.0 - torch.acos(dots) * (180.0 / math.pi)).mean(-1)
