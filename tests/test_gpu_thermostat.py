"""The time scale of the Langevin thermostat - the only thing the reference's ``diff_coef`` / ``rot_diff_coef`` configure
(mythos/input/dna2/default_simulation.toml:2-9: 2.5 and 7.5; gamma = kT / diff_coef, simulators/jax_md/utils.py:143-154).

Equipartition, NVE drift and <U> are blind to the friction convention: gamma against gamma / m, a noise amplitude that is
off by a factor, a friction that acts on the wrong variable - all of them thermalise to kT.  What is not blind:

 * the momentum autocorrelation of free particles, <p(0).p(t)> = 3 m kT exp(-gamma_t t), and of the body angular momentum
   of spherical tops, <L(0).L(t)> = 3 I kT exp(-gamma_r t): the friction is a RATE, independent of mass and inertia
   (jax_md: c1 = exp(-gamma dt) on the momentum);
 * the mean squared displacement at every time, <|x(t) - x(0)|^2> = 6 D [t - (1 - exp(-gamma t)) / gamma] with
   D = kT / (m gamma) - with the reference's gamma = kT / diff_coef and unit mass: D = diff_coef = 2.5;
 * in the overdamped regime (gamma_r >> sqrt(kT / I)) the orientation diffuses: <a1(0).a1(t)> = exp(-2 D_r t),
   D_r = kT / (I gamma_r), which with gamma_r = kT / rot_diff_coef and unit inertia is rot_diff_coef (7.5).  (At the
   reference's own gamma_r = 0.013 a nucleotide is a nearly free rotor for 1 / gamma_r = 76 time units: there the
   decay of <L.L> is the test, the orientation decorrelates inertially.)

16 384 isolated nucleotides (no bonds, empty neighbour list: zero forces), both precisions, mass and inertia away from 1
so that gamma and gamma / m cannot be confused.  Tolerances: 4 standard errors of the estimators + the 3 % the round-2
verdict asked for.
"""

import numpy as np
import pytest
import torch

from mythos_amd import _lib
from mythos_amd.energy import flat_params as fp
from mythos_amd.input import defaults

pytestmark = pytest.mark.gpu

KT = 296.15 * 0.1 / 300.0
N = 16384


def _free_nucleotides(dtype):
    from mythos_amd.hip_system import OxdnaSystem

    sim, cfg = defaults.default_configs_for("dna2")
    flat = fp.pack_flat(fp.derive_flat(2, cfg, kt=sim["kT"], salt_conc=0.5, half_charged_ends=True), _lib.param_names())
    seq = np.arange(N, dtype=np.int32) % 4
    s = OxdnaSystem(2, seq, np.ones(N, dtype=np.int32), np.zeros((0, 2), dtype=np.int32), dtype=dtype)
    s.set_params(flat)
    s.set_neighbors(np.zeros((0, 2), dtype=np.int32))
    rng = np.random.default_rng(4)
    c = rng.uniform(-50.0, 50.0, size=(N, 3))
    q = rng.standard_normal((N, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return s, torch.as_tensor(c, dtype=dtype, device=s.device).contiguous(), torch.as_tensor(q, dtype=dtype, device=s.device).contiguous()


def _a1(q):
    q0, q1, q2, q3 = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    return np.stack([q0 * q0 + q1 * q1 - q2 * q2 - q3 * q3, 2 * (q1 * q2 + q0 * q3), 2 * (q1 * q3 - q0 * q2)], axis=1)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_reference_friction_sets_the_momentum_decay_and_the_diffusion_coefficient(dtype):
    from mythos_amd.hip_system import LangevinIntegrator

    s, c, q = _free_nucleotides(dtype)
    mass, inertia = 2.0, 1.5
    gam_t, gam_r = KT / 2.5, KT / 7.5  # the reference's defaults
    integ = LangevinIntegrator(s, dt=0.005, kT=KT, gamma_t=gam_t, gamma_r=gam_r, mass=mass, inertia=(inertia,) * 3, seed=77)
    p, L = integ.init_momenta()
    x0, p0, L0 = c.double().cpu().numpy(), p.double().cpu().numpy(), L.double().cpu().numpy()
    assert abs((p0**2).sum(1).mean() / (3 * mass * KT) - 1.0) < 0.03 and abs((L0**2).sum(1).mean() / (3 * inertia * KT) - 1.0) < 0.03
    chunk, n_chunks, dt = 2000, 6, 0.005  # 60 time units: 2.4 / gamma_t, 0.8 / gamma_r
    t, cp, cl, msd = [], [], [], []
    for k in range(1, n_chunks + 1):
        integ.run(c, q, p, L, chunk)
        pk, Lk, xk = p.double().cpu().numpy(), L.double().cpu().numpy(), c.double().cpu().numpy()
        t.append(k * chunk * dt)
        cp.append((p0 * pk).sum(1).mean() / (3 * mass * KT))
        cl.append((L0 * Lk).sum(1).mean() / (3 * inertia * KT))
        msd.append(((xk - x0) ** 2).sum(1).mean())
    t, cp, cl, msd = (np.asarray(v) for v in (t, cp, cl, msd))
    se = 4.0 / np.sqrt(3 * N)  # four standard errors of a normalised correlation of 3 N unit-variance products
    np.testing.assert_allclose(cp, np.exp(-gam_t * t), rtol=0.03, atol=se)
    np.testing.assert_allclose(cl, np.exp(-gam_r * t), rtol=0.03, atol=se)
    # rate from a fit, as a number to read: gamma within 3 % (+ noise) of kT / diff_coef
    fit_t = -np.polyfit(t, np.log(cp), 1)[0]
    fit_r = -np.polyfit(t, np.log(cl), 1)[0]
    assert abs(fit_t / gam_t - 1.0) < 0.05 and abs(fit_r / gam_r - 1.0) < 0.08, (fit_t, gam_t, fit_r, gam_r)
    # mean squared displacement against the exact Ornstein-Uhlenbeck result; D = kT / (m gamma) = diff_coef / m
    D = KT / (mass * gam_t)
    assert abs(D - 2.5 / mass) < 1e-12
    want = 6.0 * D * (t - (1.0 - np.exp(-gam_t * t)) / gam_t)
    np.testing.assert_allclose(msd, want, rtol=0.03 + 4.0 * np.sqrt(2.0 / (3 * N)))


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_overdamped_rotation_diffuses_with_kT_over_inertia_times_friction(dtype):
    from mythos_amd.hip_system import LangevinIntegrator

    s, c, q = _free_nucleotides(dtype)
    inertia, gam_r = 1.5, 20.0  # sqrt(kT / I) = 0.26 << gamma_r
    integ = LangevinIntegrator(s, dt=0.005, kT=KT, gamma_t=1.0, gamma_r=gam_r, mass=1.0, inertia=(inertia,) * 3, seed=78)
    p, L = integ.init_momenta()
    a0 = _a1(q.double().cpu().numpy())
    D_r = KT / (inertia * gam_r)
    t, corr = [], []
    for k in range(1, 6):
        integ.run(c, q, p, L, 4000)
        t.append(k * 4000 * 0.005)
        corr.append((a0 * _a1(q.double().cpu().numpy())).sum(1).mean())
    t, corr = np.asarray(t), np.asarray(corr)
    # inertial correction of the exponent: 2 D_r [t - (1 - exp(-gamma t)) / gamma]; 1 / gamma = 0.05 against t >= 20
    want = np.exp(-2.0 * D_r * (t - (1.0 - np.exp(-gam_r * t)) / gam_r))
    np.testing.assert_allclose(corr, want, rtol=0.03, atol=4.0 / np.sqrt(3 * N))
    fit = -0.5 * np.polyfit(t, np.log(corr), 1)[0]
    assert abs(fit / D_r - 1.0) < 0.05, (fit, D_r)
    # the same relation at the reference's setting: gamma_r = kT / rot_diff_coef and unit inertia give D_r = rot_diff_coef
    assert abs(KT / (1.0 * (KT / 7.5)) - 7.5) < 1e-12
