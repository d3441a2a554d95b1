"""CPU oracle for the MARTINI 2/3 energy terms  --  TEST INFRASTRUCTURE ONLY.

torch-fp64 restatement of mythos/energy/martini/m2/lj.py:55-88,137-157 (shifted-cut-off
Lennard-Jones over all unbonded i<j pairs), m2/bond.py:34-40 (harmonic bonds), m2/angle.py:35-93
(G96 cosine angles) and m3/angle.py:8-11 (harmonic angles), with the periodic minimum image of
``jax_md.space.periodic(box)`` per frame (mythos/energy/martini/base.py:15-17).

Parity pin: tests/test_oracle_martini.py checks it against the GROMACS ``gmx energy`` goldens the
reference's own tests use (tests/golden/martini/m2/{lj,bond,angle}; mythos/energy/martini/m2/tests/
test_{lj,bond,angle}.py) with their tolerance (allclose rtol 1e-5).  The MARTINI-3 harmonic angle
golden needs the binary test.tpr topology (MDAnalysis) and is NOT reproduced here: that variant is
pinned only through the shared angle geometry and finite differences.  Forces are unpinned by the
reference; autograd of this file is the stand-in.
"""

from __future__ import annotations

import numpy as np
import torch

F64 = torch.float64
LJ_CUTOFF = 1.1  # m2/lj.py:57


def _disp(a, b, box):
    d = a - b
    return torch.remainder(d + 0.5 * box, box) - 0.5 * box


def lennard_jones(r, eps, sigma, cutoff=LJ_CUTOFF):
    """m2/lj.py:55-67."""
    v = 4 * eps * ((sigma / r) ** 12 - (sigma / r) ** 6)
    v_c = 4 * eps * ((sigma / cutoff) ** 12 - (sigma / cutoff) ** 6)
    return torch.where(r < cutoff, v - v_c, torch.zeros_like(v))


def lj_energy(pos, box, types, sigma, eps, bonded, chunk=2_000_000):
    """Sum over all i<j pairs, bonded pairs masked out (m2/lj.py:70-88, 137-157)."""
    n = pos.shape[0]
    iu, ju = np.triu_indices(n, k=1)
    mask = np.ones(iu.shape[0], dtype=bool)
    if len(bonded):
        lo = np.minimum(bonded[:, 0], bonded[:, 1]).astype(np.int64)
        hi = np.maximum(bonded[:, 0], bonded[:, 1]).astype(np.int64)
        mask[lo * n - lo * (lo + 1) // 2 + (hi - lo - 1)] = False
    iu, ju = torch.as_tensor(iu[mask]), torch.as_tensor(ju[mask])
    t = torch.as_tensor(types, dtype=torch.long)
    total = torch.zeros((), dtype=F64)
    for s in range(0, iu.shape[0], chunk):
        i, j = iu[s : s + chunk], ju[s : s + chunk]
        r = _disp(pos[i], pos[j], box).norm(dim=1)
        r_safe = torch.where(r < LJ_CUTOFF, r, torch.ones_like(r))
        total = total + lennard_jones(r_safe, eps[t[i], t[j]], sigma[t[i], t[j]]).where(r < LJ_CUTOFF, torch.zeros_like(r)).sum()
    return total


def bond_energy(pos, box, bonds, k, r0):
    """m2/bond.py:34-40, 64-71."""
    b = torch.as_tensor(bonds, dtype=torch.long)
    r = _disp(pos[b[:, 0]], pos[b[:, 1]], box).norm(dim=1)
    return (0.5 * k * (r - r0) ** 2).sum()


def angle_energy(pos, box, angles, k, theta0, use_g96: bool):
    """m2/angle.py:35-93: theta = atan2(|rij x rkj|, rij . rkj) on unit vectors from the central bead."""
    a = torch.as_tensor(angles, dtype=torch.long)
    rij = _disp(pos[a[:, 1]], pos[a[:, 0]], box)
    rkj = _disp(pos[a[:, 1]], pos[a[:, 2]], box)
    rij = rij / rij.norm(dim=1, keepdim=True)
    rkj = rkj / rkj.norm(dim=1, keepdim=True)
    cr = torch.linalg.cross(rij, rkj)
    theta = torch.atan2(torch.sqrt((cr**2).sum(1)), (rij * rkj).sum(1))
    term = (torch.cos(theta) - torch.cos(theta0)) if use_g96 else (theta - theta0)
    return (0.5 * k * term**2).sum()


def energies_and_forces(pos, box, types, sigma, eps, bonds, bond_k, bond_r0, angles, angle_k, angle_t0, use_g96):
    """[lj, bond, angle] and dU/dpos of their sum for one frame."""
    x = pos.detach().clone().requires_grad_(True)
    e = torch.stack([
        lj_energy(x, box, types, sigma, eps, np.asarray(bonds)),
        bond_energy(x, box, bonds, bond_k, bond_r0),
        angle_energy(x, box, angles, angle_k, angle_t0, use_g96),
    ])
    (g,) = torch.autograd.grad(e.sum(), x)
    return e.detach(), g
