"""Propeller twist (mythos/observables/propeller.py:19-71): per frame, the mean over the listed H-bonded base
pairs of 180 - acos(n_i . n_j) in degrees, n = base normal a3.  Evaluated by the HIP library
(mythos_amd/csrc/observables.h), stand-alone or in the same call as the energy launch."""

from __future__ import annotations

import numpy as np
import torch

from mythos_amd.observables import base as B

TARGETS = {"oxDNA": 21.7}  # degrees (propeller.py:13-15)


class PropellerTwist(B.HipObservable):
    def __init__(self, h_bonded_base_pairs):
        self.h_bonded_base_pairs = torch.as_tensor(np.asarray(h_bonded_base_pairs), dtype=torch.long).reshape(-1, 2)
        self.base_pairs = self.h_bonded_base_pairs.numpy()

    def __call__(self, trajectory) -> torch.Tensor:
        """(n_states,) propeller twist in degrees."""
        return self.rows(trajectory)[:, B.COL_PROPELLER]
