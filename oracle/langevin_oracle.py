"""CPU oracle of the rigid-body Langevin step  --  TEST INFRASTRUCTURE ONLY.

Restates what the reference's MD loop does per step,
``state = step_fn(state, ...)`` with ``step_fn`` from ``jax_md.simulate.nvt_langevin``
(call site mythos/simulators/jax_md/jaxmd.py:73-92; kwargs from
mythos/simulators/jax_md/utils.py:143-154).  ``jax_md==0.2.28`` is a third-party dependency
that is NOT in the reference tree and not installable here, so this file restates its published
algorithm (BAOAB; rigid bodies advanced with the NO_SQUISH free-rotor splitting of Miller et
al., J. Chem. Phys. 116, 8649 (2002); Ornstein-Uhlenbeck kick on the body-frame angular
momentum).  **Parity unpinned**: the reference's own simulator tests use a fake
``simulator_init`` (mythos/simulators/jax_md/tests/test_jaxmd.py:100-124), so no golden
vector exists for the integrator; the tests therefore check (i) this restatement against the HIP
kernels step by step with the same counter-based random numbers and (ii) physics: NVE drift,
equipartition, <U> against oxDNA's own Langevin run (tests/golden/dna2/simple-helix/energy.dat).

The quaternion-momentum form used by jax_md (conjugate momentum Pi, kick Pi += h F_q) and the
body-angular-momentum form used here are the same map for unit quaternions:
L_k = 1/2 (P_k q) . Pi  and  tau_k = -1/2 (P_k q) . dU/dq.
"""

from __future__ import annotations

import math

import numpy as np
import torch

from oracle import oxdna_oracle as orc

M32 = np.uint64(0xFFFFFFFF)


def philox4x32(counter: np.ndarray, key: tuple[int, int]) -> np.ndarray:
    """Philox4x32-10 (Salmon et al., SC'11).  counter (..., 4) uint32 -> (..., 4) uint32."""
    c = [counter[..., k].astype(np.uint64) for k in range(4)]
    k0, k1 = np.uint64(key[0]), np.uint64(key[1])
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c[0]
        p1 = np.uint64(0xCD9E8D57) * c[2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & M32
        hi1, lo1 = p1 >> np.uint64(32), p1 & M32
        c = [(hi1 ^ c[1] ^ k0) & M32, lo1, (hi0 ^ c[3] ^ k1) & M32, lo0]
        k0 = (k0 + np.uint64(0x9E3779B9)) & M32
        k1 = (k1 + np.uint64(0xBB67AE85)) & M32
    return np.stack(c, axis=-1).astype(np.uint32)


def box_muller(u0: np.ndarray, u1: np.ndarray):
    a = (u0.astype(np.float64) + 1.0) * 2.3283064365386963e-10
    b = u1.astype(np.float64) * 2.3283064365386963e-10
    r = np.sqrt(-2.0 * np.log(a))
    return r * np.cos(6.283185307179586 * b), r * np.sin(6.283185307179586 * b)


def normals6(seed: int, n: int, step: int, stream: int = 0) -> np.ndarray:
    """(n, 6) standard normals for (particle, step): the device RNG stream, in float64."""
    key = (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    ctr = np.zeros((n, 4), dtype=np.uint32)
    ctr[:, 0] = np.arange(n, dtype=np.uint32)
    ctr[:, 1] = step & 0xFFFFFFFF
    ctr[:, 2] = (step >> 32) & 0xFFFFFFFF
    ctr[:, 3] = stream
    a = philox4x32(ctr, key)
    ctr[:, 3] = stream + 1
    b = philox4x32(ctr, key)
    z = np.empty((n, 6))
    z[:, 0], z[:, 1] = box_muller(a[:, 0], a[:, 1])
    z[:, 2], z[:, 3] = box_muller(a[:, 2], a[:, 3])
    z[:, 4], z[:, 5] = box_muller(b[:, 0], b[:, 1])
    return z


def _pk(q: np.ndarray, k: int) -> np.ndarray:
    q0, q1, q2, q3 = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    if k == 0:
        return np.stack([-q1, q0, q3, -q2], axis=1)
    if k == 1:
        return np.stack([-q2, -q3, q0, q1], axis=1)
    return np.stack([-q3, q2, -q1, q0], axis=1)


def free_rotor(q, L, k, h, inertia):
    """One NO_SQUISH factor: rotation about body axis k by phi = h L_k / I_k."""
    phi = h * L[:, k] / inertia[k]
    c, s = np.cos(0.5 * phi)[:, None], np.sin(0.5 * phi)[:, None]
    q = c * q + s * _pk(q, k)
    a, b = (k + 1) % 3, (k + 2) % 3
    cf, sf = np.cos(phi), np.sin(phi)
    L = L.copy()
    la, lb = L[:, a].copy(), L[:, b].copy()
    L[:, a] = cf * la + sf * lb
    L[:, b] = -sf * la + cf * lb
    return q, L


def drift(x, q, p, L, h, mass, inertia):
    x = x + h * p / mass
    for k, hh in ((2, 0.5 * h), (1, 0.5 * h), (0, h), (1, 0.5 * h), (2, 0.5 * h)):
        q, L = free_rotor(q, L, k, hh, inertia)
    return x, q, L


class LangevinOracle:
    def __init__(self, model, P, top_tensors, box, dt, kT, gamma_t, gamma_r, mass=1.0, inertia=(1.0, 1.0, 1.0), seed=0, is_rna=None):
        self.model, self.P, self.box = model, P, box
        self.is_rna = None if is_rna is None else torch.as_tensor(np.asarray(is_rna, dtype=bool))  # oxNA (model 4)
        self.seq, self.is_end, self.bonded, self.unbonded = top_tensors
        self.dt, self.kT, self.mass = dt, kT, mass
        self.inertia = np.asarray(inertia, dtype=np.float64)
        self.c1_t = math.exp(-gamma_t * dt)
        self.c2_t = math.sqrt(kT * (1 - self.c1_t**2) * mass)
        self.c1_r = math.exp(-gamma_r * dt)
        self.c2_r = np.sqrt(kT * (1 - self.c1_r**2) * self.inertia)
        self.seed = seed
        self.step_index = 0

    def forces(self, x, q):
        if self.model == 4:
            u, gc, gq = orc.energy_and_grads_na1(self.P, torch.as_tensor(x), torch.as_tensor(q), self.seq, self.is_rna, self.is_end,
                                                 self.bonded, self.unbonded, box=self.box)
        else:
            u, gc, gq = orc.energy_and_grads(
                self.model, self.P, torch.as_tensor(x), torch.as_tensor(q), self.seq, self.is_end, self.bonded, self.unbonded, box=self.box
            )
        tau = orc.quat_grad_to_body_torque(torch.as_tensor(q), gq)
        return float(u), -gc.numpy(), tau.numpy()

    def step(self, x, q, p, L):
        """B A O A [force] B on (x, q, p, L); returns the new state and U at the new positions."""
        h = 0.5 * self.dt
        _, F, tau = self.forces(x, q)
        p = p + h * F
        L = L + h * tau
        x, q, L = drift(x, q, p, L, h, self.mass, self.inertia)
        z = normals6(self.seed, x.shape[0], self.step_index)
        p = self.c1_t * p + self.c2_t * z[:, :3]
        L = self.c1_r * L + self.c2_r[None, :] * z[:, 3:]
        x, q, L = drift(x, q, p, L, h, self.mass, self.inertia)
        q = q / np.linalg.norm(q, axis=1, keepdims=True)
        u, F, tau = self.forces(x, q)
        p = p + h * F
        L = L + h * tau
        self.step_index += 1
        return x, q, p, L, u

    def run(self, x, q, p, L, n_steps):
        """n_steps of the same map, one force evaluation per step (the closing kick of a step and
        the opening kick of the next share F).  Arrays are updated in place; returns the U trace."""
        h = 0.5 * self.dt
        us = []
        u, F, tau = self.forces(x, q)
        for _ in range(n_steps):
            pp = p + h * F
            LL = L + h * tau
            xx, qq, LL = drift(x, q, pp, LL, h, self.mass, self.inertia)
            z = normals6(self.seed, x.shape[0], self.step_index)
            pp = self.c1_t * pp + self.c2_t * z[:, :3]
            LL = self.c1_r * LL + self.c2_r[None, :] * z[:, 3:]
            xx, qq, LL = drift(xx, qq, pp, LL, h, self.mass, self.inertia)
            qq = qq / np.linalg.norm(qq, axis=1, keepdims=True)
            u, F, tau = self.forces(xx, qq)
            x[:], q[:], p[:], L[:] = xx, qq, pp + h * F, LL + h * tau
            self.step_index += 1
            us.append(u)
        return np.array(us)

    def kinetic(self, p, L):
        return 0.5 * (p**2).sum() / self.mass, 0.5 * ((L**2) / self.inertia[None, :]).sum()
