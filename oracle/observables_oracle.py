"""Torch restatement of the structural observables  --  TEST INFRASTRUCTURE ONLY.

The checker of mythos_amd/observables (whose product path is the HIP code in mythos_amd/csrc/observables.h): the same
quantities written with torch ops on whatever device the trajectory lives on, following the reference function by
function - mythos/observables/base.py:24-66 (local helical axis, quartets), propeller.py:19-71, pitch.py:33-102,
rise.py:21-80, persistence_length.py:21-185.  Pinned (tests/test_observables_cpu.py) by the known answers the
reference's own tests hold - observables/tests/test_rise.py:14-83 (14.753608, 11.065206), test_propeller.py:39-80 (120) -
and by closed-form geometries; test_lp.py:50-59 reads data/test-data/simple-helix-60bp/output.dat, which this snapshot of
the reference does not contain, so the persistence-length fit is pinned by a discrete worm-like chain only.
Only tests/ may import this module.
"""

from __future__ import annotations

import math

import torch

ANGSTROMS_PER_OXDNA_LENGTH = 8.518  # mythos/utils/units.py:5-8


def axes_from_quaternion(q: torch.Tensor):
    """a1 (back-base vector), a2, a3 (base normal) from [w, x, y, z] quaternions (mythos/energy/utils.py:18-36)."""
    q0, q1, q2, q3 = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    a1 = torch.stack([q0 * q0 + q1 * q1 - q2 * q2 - q3 * q3, 2 * (q1 * q2 + q0 * q3), 2 * (q1 * q3 - q0 * q2)], dim=-1)
    a2 = torch.stack([2 * (q1 * q2 - q0 * q3), q0 * q0 - q1 * q1 + q2 * q2 - q3 * q3, 2 * (q2 * q3 + q0 * q1)], dim=-1)
    a3 = torch.stack([2 * (q1 * q3 + q0 * q2), 2 * (q2 * q3 - q0 * q1), q0 * q0 - q1 * q1 - q2 * q2 + q3 * q3], dim=-1)
    return a1, a2, a3


def nucleotide_sites(trajectory, geometry: dict, model: int = 2):
    """(base_sites, back_sites, stack_sites), each (S, N, 3): the site algebra of ``Nucleotide.from_rigid_body``
    (dna1/nucleotide.py:29-53, dna2/nucleotide.py:30-58) with the TOML ``[geometry]`` values."""
    c = trajectory.center
    a1, a2, a3 = axes_from_quaternion(trajectory.orientation.vec)
    if model == 3:  # rna2/nucleotide.py:52-58: backbone site on a1 and a3
        base = c + float(geometry["pos_base"]) * a1
        stack = c + float(geometry["pos_stack"]) * a1
        back = c + float(geometry["pos_back_a1"]) * a1 + float(geometry["pos_back_a3"]) * a3
        return base, back, stack
    base = c + float(geometry["com_to_hb"]) * a1
    stack = c + float(geometry["com_to_stacking"]) * a1
    if model == 2:
        back = c + float(geometry["com_to_backbone_x"]) * a1 + float(geometry["com_to_backbone_y"]) * a2
    else:
        back = c + float(geometry["com_to_backbone"]) * a1
    return base, back, stack


def get_duplex_quartets(n_nucs_per_strand: int) -> torch.Tensor:
    """All pairs of adjacent base pairs of a duplex whose strands are stored one after the other:
    base pair k = (k, 2n - 1 - k); quartet k = (base pair k, base pair k + 1).  Shape (n - 1, 2, 2)."""
    n = int(n_nucs_per_strand)
    k = torch.arange(n)
    bps = torch.stack([k, 2 * n - 1 - k], dim=1)
    return torch.stack([bps[:-1], bps[1:]], dim=1)


def base_pair_midpoints(quartets: torch.Tensor, base_sites: torch.Tensor):
    q = quartets.to(base_sites.device)
    m1 = 0.5 * (base_sites[..., q[:, 0, 0], :] + base_sites[..., q[:, 0, 1], :])
    m2 = 0.5 * (base_sites[..., q[:, 1, 0], :] + base_sites[..., q[:, 1, 1], :])
    return m1, m2


def local_helical_axis(quartets: torch.Tensor, base_sites: torch.Tensor, displacement_fn):
    """Unit vector from the midpoint of the first base pair of every quartet to the midpoint of the second, and
    its length (base.py:24-45)."""
    m1, m2 = base_pair_midpoints(quartets, base_sites)
    dr = displacement_fn(m2, m1)
    norm = dr.norm(dim=-1, keepdim=True)
    return dr / norm, norm[..., 0]


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


def compute_pitch(avg_pitch_angle):
    return math.pi / avg_pitch_angle


class PitchAngle:
    def __init__(self, quartets, displacement_fn, geometry: dict, model: int = 2):
        self.quartets = torch.as_tensor(quartets, dtype=torch.long).reshape(-1, 2, 2)
        self.displacement_fn, self.geometry, self.model = displacement_fn, geometry, model

    def __call__(self, trajectory) -> torch.Tensor:
        """(n_states,) mean pitch angle in radians."""
        base, back, _ = nucleotide_sites(trajectory, self.geometry, self.model)
        axis, _ = local_helical_axis(self.quartets, base, self.displacement_fn)
        q = self.quartets.to(base.device)

        def projected(bp):
            bb = self.displacement_fn(back[..., q[:, bp, 1], :], back[..., q[:, bp, 0], :])
            bb = bb - (bb * axis).sum(-1, keepdim=True) * axis
            return bb / bb.norm(dim=-1, keepdim=True)

        cos = (projected(0) * projected(1)).sum(-1).clamp(-1.0, 1.0)
        return torch.acos(cos).mean(-1)


class Rise:
    def __init__(self, quartets, displacement_fn, geometry: dict, model: int = 2):
        self.quartets = torch.as_tensor(quartets, dtype=torch.long).reshape(-1, 2, 2)
        self.displacement_fn, self.geometry, self.model = displacement_fn, geometry, model

    def __call__(self, trajectory) -> torch.Tensor:
        base, _, _ = nucleotide_sites(trajectory, self.geometry, self.model)
        axis, _ = local_helical_axis(self.quartets, base, self.displacement_fn)
        m1, m2 = base_pair_midpoints(self.quartets, base)
        rise = (self.displacement_fn(m2, m1) * axis).sum(-1)
        return rise.mean(-1) * ANGSTROMS_PER_OXDNA_LENGTH


def vector_autocorrelate(vecs: torch.Tensor) -> torch.Tensor:
    """(..., n, 3) ordered vectors -> (..., n) mean of v_i . v_(i+d) over the n - d pairs at every lag d
    (persistence_length.py:47-75)."""
    n = vecs.shape[-2]
    gram = vecs @ vecs.transpose(-1, -2)
    sums = torch.stack([torch.diagonal(gram, offset=d, dim1=-2, dim2=-1).sum(-1) for d in range(n)], dim=-1)
    return sums / torch.arange(n, 0, -1, dtype=vecs.dtype, device=vecs.device)


def persistence_length_fit(correlations: torch.Tensor, l0_av):
    """Lp and offset of the line log C(d) = offset - d l0 / Lp (persistence_length.py:21-44).

    Solved as the (n, 2) least-squares system with numpy's SVD-based solver, so that it is NOT the product's arithmetic
    (mythos_amd/observables/persistence_length.py writes out the normal equations of a straight line): comparing the
    two is a check, not a self-comparison."""
    import numpy as np

    y = np.log(correlations.detach().double().cpu().numpy())
    d = np.arange(y.shape[0], dtype=np.float64)
    (offset, slope), *_ = np.linalg.lstsq(np.stack([np.ones_like(d), d], axis=1), y, rcond=None)
    l0 = l0_av.detach().double().cpu() if isinstance(l0_av, torch.Tensor) else torch.tensor(float(l0_av), dtype=torch.float64)
    return -l0 / slope, torch.tensor(offset, dtype=torch.float64)


class PersistenceLength:
    def __init__(self, quartets, displacement_fn, geometry: dict, model: int = 2, truncate: int | None = None,
                 skip_ends: bool = True):
        self.quartets = torch.as_tensor(quartets, dtype=torch.long).reshape(-1, 2, 2)
        self.displacement_fn, self.geometry, self.model = displacement_fn, geometry, model
        self.truncate, self.skip_ends = truncate, skip_ends

    def get_all_corrs_and_l0s(self, trajectory):
        """(S, n_q') correlations and (S,) mean base-pair spacing per frame; n_q' = n_q - 4 with ``skip_ends``
        (persistence_length.py:78-91, :168-185)."""
        base, _, _ = nucleotide_sites(trajectory, self.geometry, self.model)
        axis, l0 = local_helical_axis(self.quartets, base, self.displacement_fn)
        if self.skip_ends:
            axis, l0 = axis[..., 2:-2, :], l0[..., 2:-2]
        return vector_autocorrelate(axis), l0.mean(-1)

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
