"""Helical pitch (mythos/observables/pitch.py:13-102): per frame, the mean angle between the backbone-backbone
vectors of adjacent base pairs after projecting out the local helical axis; pitch = pi / <angle> in the
reference's convention (``compute_pitch``), target 10.5 bp per turn."""

from __future__ import annotations

import math

import torch

from mythos_amd.observables import base as B

TARGETS = {"oxDNA": 10.5}  # bp / turn


def compute_pitch(avg_pitch_angle):
    return math.pi / avg_pitch_angle


class PitchAngle:
    def __init__(self, quartets, displacement_fn, geometry: dict, model: int = 2):
        self.quartets = torch.as_tensor(quartets, dtype=torch.long).reshape(-1, 2, 2)
        self.displacement_fn, self.geometry, self.model = displacement_fn, geometry, model

    def __call__(self, trajectory) -> torch.Tensor:
        """(n_states,) mean pitch angle in radians."""
        base, back, _ = B.nucleotide_sites(trajectory, self.geometry, self.model)
        axis, _ = B.local_helical_axis(self.quartets, base, self.displacement_fn)
        q = self.quartets.to(base.device)

        def projected(bp):
            bb = self.displacement_fn(back[..., q[:, bp, 1], :], back[..., q[:, bp, 0], :])
            bb = bb - (bb * axis).sum(-1, keepdim=True) * axis
            return bb / bb.norm(dim=-1, keepdim=True)

        cos = (projected(0) * projected(1)).sum(-1).clamp(-1.0, 1.0)
        return torch.acos(cos).mean(-1)
