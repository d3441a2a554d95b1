"""Shared pieces of the structural observables (mythos/observables/base.py:13-66): nucleotide sites from a
trajectory, adjacent base pairs ("quartets") of a duplex and the local helical axis they define.  All torch, on
the device the trajectory lives on: a few flops per base pair per frame."""

from __future__ import annotations

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
    a1, a2, _ = axes_from_quaternion(trajectory.orientation.vec)
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
