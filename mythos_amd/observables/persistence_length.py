"""Persistence length (mythos/observables/persistence_length.py:21-185).

Per frame (HIP, mythos_amd/csrc/observables.h): the unit vectors l_k between the midpoints of adjacent base pairs (the
local helical axis of each quartet), their autocorrelation C(d) = mean_k l_k . l_(k+d) and the mean midpoint spacing
<l0>.  Over a trajectory (optionally with DiffTRe weights): log C(d) = offset - d <l0> / Lp, fitted by least squares (the
closed form of a straight-line fit, on torch tensors: differentiable with respect to the weights).  Lp comes back in the length unit of the trajectory (oxDNA units;
x 0.8518 for nm).
"""

from __future__ import annotations

import numpy as np
import torch

from mythos_amd.observables import base as B

TARGETS = {"oxDNA": 47.5}  # nm (persistence_length.py:15-17)


def persistence_length_fit(correlations: torch.Tensor, l0_av):
    """Lp and offset of the line log C(d) = offset - d l0 / Lp (persistence_length.py:21-44)."""
    # the normal equations of a straight line, written out: n, sum d and sum d^2 are numbers known on the host, the two
    # sums over y are the only device work (differentiable with respect to the DiffTRe weights behind ``correlations``).
    # Through round 3 this was torch.linalg.lstsq on the (n_lags, 2) system: a QR solver launch chain of ~0.5 ms for a
    # two-parameter fit, on the path of every DiffTRe iteration that reweights the persistence length.
    n = int(correlations.shape[0])
    if n < 2:
        raise ValueError(f"persistence_length_fit needs at least 2 lags of the axis autocorrelation to fit a line, got {n}")
    y = torch.log(correlations)
    d = torch.arange(n, dtype=y.dtype, device=y.device)
    sd, sdd = n * (n - 1) / 2.0, (n - 1) * n * (2 * n - 1) / 6.0
    sy, sdy = y.sum(), (d * y).sum()
    slope = (n * sdy - sd * sy) / (n * sdd - sd * sd)
    offset = (sy - slope * sd) / n
    return -l0_av / slope, offset


class PersistenceLength(B.HipObservable):
    def __init__(self, quartets, displacement_fn, geometry: dict, model: int = 2, truncate: int | None = None,
                 skip_ends: bool = True):
        self.quartets = np.asarray(quartets, dtype=np.int64).reshape(-1, 2, 2)
        self.displacement_fn, self.geometry, self.model = displacement_fn, geometry, model
        self.truncate, self.skip_ends = truncate, skip_ends

    def get_all_corrs_and_l0s(self, trajectory):
        """(S, n_q') correlations and (S,) mean base-pair spacing per frame; n_q' = n_q - 4 with ``skip_ends``
        (persistence_length.py:78-91, :168-185)."""
        rows = self.rows(trajectory)
        return rows[:, B.COL_CORR:], rows[:, B.COL_L0]

    def lp_fit(self, trajectory, weights=None):
        corrs, l0s = self.get_all_corrs_and_l0s(trajectory)
        if weights is not None:
            w = torch.as_tensor(weights, dtype=corrs.dtype, device=corrs.device)
            if w.shape != l0s.shape:
                raise TypeError(f"weights must have shape {tuple(l0s.shape)}, got {tuple(w.shape)}")
            corr, l0 = w @ corrs, w @ l0s
        else:
            corr, l0 = corrs.mean(0), l0s.mean(0)
        if self.truncate:
            corr = corr[: self.truncate]
        return persistence_length_fit(corr, l0)

    def __call__(self, trajectory, weights=None) -> torch.Tensor:
        return self.lp_fit(trajectory, weights)[0]
