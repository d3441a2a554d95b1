"""CPU oracle of the MARTINI Langevin step  --  TEST INFRASTRUCTURE ONLY.

The reference has no MARTINI integrator (it runs GROMACS as an external process,
mythos/simulators/gromacs/), so there is nothing to pin this against: **parity unpinned**.  The file
restates the BAOAB splitting for point particles

    v += h F/m;  x += h v;  v = c1 v + sqrt(kT (1 - c1^2) / m) xi;  x += h v;  F = -dU/dx;  v += h F/m

with h = dt/2, c1 = exp(-gamma dt), the forces of oracle/martini_oracle.py (itself pinned to the GROMACS
energies the reference ships) and the device's counter-based normal stream (oracle/langevin_oracle.py), so that
tests can compare the HIP kernel step by step in fp64.
"""

from __future__ import annotations

import math

import numpy as np
import torch

from oracle import martini_oracle as mo
from oracle.langevin_oracle import normals6


class MartiniLangevinOracle:
    def __init__(self, types, sigma, eps, bonds, bond_k, bond_r0, angles, angle_k, angle_t0, use_g96, box, dt, kT, gamma,
                 mass, seed=0):
        self.args = (types, torch.as_tensor(sigma), torch.as_tensor(eps), bonds, torch.as_tensor(bond_k),
                     torch.as_tensor(bond_r0), angles, torch.as_tensor(angle_k), torch.as_tensor(angle_t0), use_g96)
        self.box = torch.as_tensor(np.asarray(box, dtype=np.float64))
        self.dt, self.kT, self.gamma, self.seed = dt, kT, gamma, seed
        self.mass = np.asarray(mass, dtype=np.float64).reshape(-1, 1)
        self.c1 = math.exp(-gamma * dt)
        self.step_index = 0

    def forces(self, x):
        types, sigma, eps, bonds, bk, br, angles, ak, at, g96 = self.args
        e, g = mo.energies_and_forces(torch.as_tensor(x), self.box, types, sigma, eps, bonds, bk, br, angles, ak, at, g96)
        return e.numpy(), -g.numpy()

    def run(self, x, v, n_steps):
        """In place; returns the list of (lj, bond, angle, kinetic) at the END of every step."""
        h = 0.5 * self.dt
        out = []
        _, F = self.forces(x)
        for _ in range(n_steps):
            v += h * F / self.mass
            x += h * v
            z = normals6(self.seed, x.shape[0], self.step_index)[:, :3]
            v[:] = self.c1 * v + np.sqrt(self.kT * (1.0 - self.c1**2) / self.mass) * z
            x += h * v
            e, F = self.forces(x)
            v += h * F / self.mass
            self.step_index += 1
            out.append((*e, 0.5 * float((self.mass * v**2).sum())))
        return np.array(out)
