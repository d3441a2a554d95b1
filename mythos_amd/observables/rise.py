"""Helical rise (mythos/observables/rise.py:18-80): per frame, the mean over quartets of the displacement between
the midpoints of adjacent base pairs projected on the local helical axis, in Angstrom.  Evaluated by the HIP library."""

from __future__ import annotations

import numpy as np
import torch

from mythos_amd.observables import base as B

TARGETS = {"oxDNA": 3.4}  # Angstrom


class Rise(B.HipObservable):
    def __init__(self, quartets, displacement_fn, geometry: dict, model: int = 2):
        self.quartets = np.asarray(quartets, dtype=np.int64).reshape(-1, 2, 2)
        self.displacement_fn, self.geometry, self.model = displacement_fn, geometry, model
        self.skip_ends = False

    def __call__(self, trajectory) -> torch.Tensor:
        return self.rows(trajectory)[:, B.COL_RISE]
