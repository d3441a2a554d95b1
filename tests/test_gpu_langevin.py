"""GPU tests of the fused Langevin step kernel (mythos_langevin_run).

The reference pins nothing here (its simulator tests use a fake integrator), so:
 * step-by-step parity in fp64 against oracle/langevin_oracle.py fed with the same Philox stream,
 * physics: NVE drift, equipartition, <U> vs oxDNA's own Langevin run (golden energy.dat),
 * plumbing: determinism, save cadence, dynamic Verlet list == static all-pairs list.
"""

import numpy as np
import pytest
import torch

from mythos_amd import _lib
from mythos_amd.energy import flat_params as fp
from mythos_amd.input import defaults
from mythos_amd.utils import generators
from tests import helpers as H

pytestmark = pytest.mark.gpu

KT = 296.15 * 0.1 / 300.0


def _make(model, top, box, dtype, hce=False, salt=0.5):
    from mythos_amd.hip_system import OxdnaSystem

    sim, cfg = defaults.default_configs_for(H.model_dir(model))
    flat = fp.pack_flat(fp.derive_flat(model, cfg, kt=sim["kT"], salt_conc=salt, half_charged_ends=hce), _lib.param_names())
    s = OxdnaSystem(model, top.seq, top.is_end, top.bonded_neighbors, box=box, dtype=dtype)
    s.set_params(flat)
    return s, sim


def _state(c, q, dtype, dev):
    return torch.as_tensor(c, dtype=dtype, device=dev).contiguous(), torch.as_tensor(q, dtype=dtype, device=dev).contiguous()


@pytest.mark.parametrize(("model", "name", "salt"), [(2, "simple-helix", 0.5), (1, "simple-helix", 0.5), (3, "simple-helix-12bp", 1.0),
                                                     (3, "simple-coax", 1.0)])
def test_step_by_step_parity_with_oracle_fp64(model, name, salt, md_lanes):
    """(oxRNA2, model 3: the integrator keeps the backbone site on a1 and a3 - the offsets the frames carry)"""
    from mythos_amd.hip_system import LangevinIntegrator
    from oracle.langevin_oracle import LangevinOracle

    top, traj, _, _ = H.load_golden(model, name)
    s, sim = _make(model, top, traj.box_size, torch.float64, salt=salt)
    s.set_neighbors(top.unbonded_neighbors)
    gam_t, gam_r = KT / 2.5, KT / 7.5
    integ = LangevinIntegrator(s, dt=0.005, kT=KT, gamma_t=gam_t, gamma_r=gam_r, mass=1.0, inertia=(1.0, 1.3, 0.8), seed=0x1234ABCD5678)
    c, q = _state(traj.center[0], traj.quaternions[0], torch.float64, s.device)
    p, L = integ.init_momenta()
    x0, q0, p0, L0 = (t.cpu().numpy().copy() for t in (c, q, p, L))
    n_steps = 6
    tc, tq, et = integ.run(c, q, p, L, n_steps, save_every=1)
    orc = LangevinOracle(
        model, H.oracle_params(model, salt=salt), H.topo_tensors(top), traj.box_size, 0.005, KT, gam_t, gam_r, 1.0, (1.0, 1.3, 0.8),
        seed=0x1234ABCD5678
    )
    x, qq, pp, LL = x0, q0, p0, L0
    for k in range(n_steps):
        x, qq, pp, LL, u = orc.step(x, qq, pp, LL)
        np.testing.assert_allclose(tc[k].cpu().numpy(), x, rtol=0, atol=1e-10)
        np.testing.assert_allclose(tq[k].cpu().numpy(), qq, rtol=0, atol=1e-10)
        assert abs(et[k, :8].sum().item() - u) < 1e-8 * abs(u)
        ke_t, ke_r = orc.kinetic(pp, LL)
        assert abs(et[k, 8].item() - ke_t) < 1e-9 * ke_t
        assert abs(et[k, 9].item() - ke_r) < 1e-9 * ke_r
    np.testing.assert_allclose(c.cpu().numpy(), x, atol=1e-10)
    np.testing.assert_allclose(p.cpu().numpy(), pp, atol=1e-9)
    np.testing.assert_allclose(L.cpu().numpy(), LL, atol=1e-9)
    assert integ.step == n_steps


@pytest.mark.parametrize(("spin", "steps", "tol"), [(30.0, 3, 1e-13), (300.0, 1, 1e-12), (1200.0, 1, 3e-8)])
def test_fast_spinning_free_nucleotides_take_the_rotor_beyond_its_series(spin, steps, tol):
    """The fp64 rotor evaluates sin / cos of a sub-step's half angle by a short Taylor series (|x| < 1/32, where thermal
    motion keeps it by a factor of fifteen) and halves a larger angle until it fits.  Free asymmetric tops (no
    neighbours: zero forces, no friction, no noise) with angular momenta of 30 - 60, 300 - 600 and 1 200 - 2 400 put the
    half angle of the long factor at 0.04 - 0.08, 0.4 - 0.8 and 1.5 - 3 rad: up to two halvings, six, eight.  Against the
    oracle's drift (numpy sin / cos).  A top that turns by radians per sub-step is a chaotic map - the angle of every
    factor is the angular momentum the previous one left - so the comparison is over one step there, and the
    tolerance is what a numpy emulation of the kernel's arithmetic shows against the oracle (1e-15, 1e-14, 3e-10),
    with a margin."""
    from mythos_amd.hip_system import LangevinIntegrator, OxdnaSystem
    from oracle.langevin_oracle import drift

    n = 1024
    sim, cfg = defaults.default_configs_for("dna2")
    flat = fp.pack_flat(fp.derive_flat(2, cfg, kt=sim["kT"], salt_conc=0.5, half_charged_ends=True), _lib.param_names())
    s = OxdnaSystem(2, np.arange(n, dtype=np.int32) % 4, np.ones(n, dtype=np.int32), np.zeros((0, 2), dtype=np.int32), dtype=torch.float64)
    s.set_params(flat)
    s.set_neighbors(np.zeros((0, 2), dtype=np.int32))
    rng = np.random.default_rng(int(spin))
    x = rng.uniform(-20.0, 20.0, size=(n, 3))
    q = rng.standard_normal((n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    p0 = 0.3 * rng.standard_normal((n, 3))
    L0 = spin * (1.0 + rng.random((n, 3))) * rng.choice([-1.0, 1.0], size=(n, 3))
    inertia = np.array([1.0, 1.3, 0.8])
    integ = LangevinIntegrator(s, dt=0.005, kT=KT, gamma_t=0.0, gamma_r=0.0, mass=1.0, inertia=tuple(inertia), seed=7)
    c, qd = _state(x, q, torch.float64, s.device)
    p, L = torch.as_tensor(p0, device=s.device).contiguous(), torch.as_tensor(L0, device=s.device).contiguous()
    integ.run(c, qd, p, L, steps)
    LL = L0.copy()
    for _ in range(steps):
        x, q, LL = drift(x, q, p0, LL, 0.0025, 1.0, inertia)
        x, q, LL = drift(x, q, p0, LL, 0.0025, 1.0, inertia)
        q = q / np.linalg.norm(q, axis=1, keepdims=True)
    np.testing.assert_allclose(qd.cpu().numpy(), q, rtol=0, atol=tol)
    assert (np.abs(L.cpu().numpy() - LL).max(1) <= tol * np.linalg.norm(LL, axis=1)).all()
    np.testing.assert_allclose(c.cpu().numpy(), x, rtol=0, atol=1e-12)
    np.testing.assert_array_equal(p.cpu().numpy(), p0)


def test_fp32_tracks_fp64_over_short_run():
    from mythos_amd.hip_system import LangevinIntegrator

    top, traj, _, _ = H.load_golden(2, "simple-helix")
    out = {}
    for dtype in (torch.float64, torch.float32):
        s, _ = _make(2, top, traj.box_size, dtype)
        s.set_neighbors(top.unbonded_neighbors)
        integ = LangevinIntegrator(s, dt=0.005, kT=KT, gamma_t=KT / 2.5, gamma_r=KT / 7.5, seed=7)
        c, q = _state(traj.center[0], traj.quaternions[0], dtype, s.device)
        p = torch.zeros_like(c)
        L = torch.zeros_like(c)
        integ.run(c, q, p, L, 20)
        out[dtype] = (c.double().cpu(), q.double().cpu())
    assert (out[torch.float32][0] - out[torch.float64][0]).abs().max() < 1e-3
    assert (out[torch.float32][1] - out[torch.float64][1]).abs().max() < 1e-3


def test_determinism_and_seed_dependence():
    from mythos_amd.hip_system import LangevinIntegrator

    top, traj, _, _ = H.load_golden(2, "simple-helix")
    s, _ = _make(2, top, traj.box_size, torch.float32)
    s.set_neighbors(top.unbonded_neighbors)
    res = []
    for seed in (3, 3, 4):
        integ = LangevinIntegrator(s, dt=0.005, kT=KT, gamma_t=KT / 2.5, gamma_r=KT / 7.5, seed=seed)
        c, q = _state(traj.center[0], traj.quaternions[0], torch.float32, s.device)
        p, L = integ.init_momenta()
        integ.run(c, q, p, L, 200)
        res.append((c.clone(), q.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert not torch.equal(res[0][0], res[2][0])


def test_split_runs_equal_one_run():
    """run(a) then run(b) continues the RNG stream: same result as run(a + b)."""
    from mythos_amd.hip_system import LangevinIntegrator

    top, traj, _, _ = H.load_golden(2, "simple-helix")
    s, _ = _make(2, top, traj.box_size, torch.float64)
    s.set_neighbors(top.unbonded_neighbors)
    outs = []
    for plan in ((50,), (20, 30)):
        integ = LangevinIntegrator(s, dt=0.005, kT=KT, gamma_t=KT / 2.5, gamma_r=KT / 7.5, seed=11)
        c, q = _state(traj.center[0], traj.quaternions[0], torch.float64, s.device)
        p, L = integ.init_momenta()
        for n in plan:
            integ.run(c, q, p, L, n)
        outs.append(c.clone())
    torch.testing.assert_close(outs[0], outs[1], rtol=0, atol=1e-11)


def test_nve_energy_conservation_fp64():
    from mythos_amd.hip_system import LangevinIntegrator

    top, traj, _, _ = H.load_golden(2, "simple-helix")
    s, _ = _make(2, top, traj.box_size, torch.float64)
    s.set_neighbors(top.unbonded_neighbors)
    integ = LangevinIntegrator(s, dt=0.002, kT=KT, gamma_t=0.0, gamma_r=0.0, seed=5)
    c, q = _state(traj.center[0], traj.quaternions[0], torch.float64, s.device)
    p, L = integ.init_momenta()
    _, _, et = integ.run(c, q, p, L, 4000, save_every=100)
    tot = et.sum(1).cpu().numpy()
    n = top.n_nucleotides
    assert np.ptp(tot) / n < 2e-3, tot / n  # bounded fluctuation, no drift
    assert abs(tot[-1] - tot[0]) / n < 2e-3


def test_equipartition_and_mean_potential_energy():
    from mythos_amd.hip_system import LangevinIntegrator

    top, traj, _, energy = H.load_golden(2, "simple-helix")
    s, _ = _make(2, top, traj.box_size, torch.float32)
    s.set_neighbors(top.unbonded_neighbors)
    # oxDNA's golden run: T = 296.15 K, dt 0.003, diff_coeff 2.5 (tests/golden/dna2/simple-helix/input)
    integ = LangevinIntegrator(s, dt=0.003, kT=KT, gamma_t=KT / 2.5 * 20, gamma_r=KT / 7.5 * 60, seed=2024)
    c, q = _state(traj.center[-1], traj.quaternions[-1], torch.float32, s.device)
    p, L = integ.init_momenta()
    integ.run(c, q, p, L, 20000)
    _, _, et = integ.run(c, q, p, L, 200000, save_every=100)
    et = et.cpu().numpy()
    n = top.n_nucleotides
    ke_t, ke_r = et[:, 8].mean(), et[:, 9].mean()
    assert abs(ke_t / (1.5 * n * KT) - 1.0) < 0.05, ke_t / (1.5 * n * KT)
    assert abs(ke_r / (1.5 * n * KT) - 1.0) < 0.05, ke_r / (1.5 * n * KT)
    u = et[:, :8].sum(1).mean() / n
    assert abs(u - energy.mean()) < 0.06, (u, energy.mean())


def test_dynamic_verlet_list_matches_static_all_pairs():
    from mythos_amd.hip_system import LangevinIntegrator

    top, c0, q0 = generators.ideal_duplex(48, model=2, seed=3)
    outs = []
    for dynamic in (False, True):
        s, _ = _make(2, top, None, torch.float64, hce=True)
        integ = LangevinIntegrator(s, dt=0.003, kT=KT, gamma_t=KT / 2.5, gamma_r=KT / 7.5, seed=99)
        if dynamic:
            integ.set_neighbor_policy(r_cut=3.3, skin=0.6, every=10)
        else:
            s.set_neighbors(top.unbonded_neighbors)
        c, q = _state(c0, q0, torch.float64, s.device)
        p, L = integ.init_momenta()
        integ.run(c, q, p, L, 60)
        outs.append(c.clone())
        if dynamic:
            mx, mean = s.neighbor_stats()
            assert mean < top.n_nucleotides - 3
    torch.testing.assert_close(outs[0], outs[1], rtol=0, atol=1e-9)


def test_a_site_leaving_its_skin_halts_rebuilds_and_resumes_exactly():
    """A skin far too small for the rebuild interval: the step that moves a site out of its skin halts the launches
    queued behind it, the run rebuilds at that state and carries on.  The trajectory must equal the one on a static
    all-pairs list (fp64, to summation order), with saved frames and energies at the right steps; the number of
    recoveries is reported."""
    from mythos_amd.hip_system import LangevinIntegrator

    top, c0, q0 = generators.ideal_duplex(48, model=2, seed=3)
    outs, trajs, ens = [], [], []
    for dynamic in (False, True):
        s, _ = _make(2, top, None, torch.float64, hce=True)
        integ = LangevinIntegrator(s, dt=0.003, kT=KT, gamma_t=KT / 2.5, gamma_r=KT / 7.5, seed=99)
        if dynamic:
            integ.set_neighbor_policy(r_cut=3.3, skin=0.05, every=1000)  # skin / 2 = 0.025: a few steps
        else:
            s.set_neighbors(top.unbonded_neighbors)
        c, q = _state(c0, q0, torch.float64, s.device)
        p, L = integ.init_momenta()
        tc, _, et = integ.run(c, q, p, L, 90, save_every=15)
        outs.append(torch.cat([c.reshape(-1), q.reshape(-1), p.reshape(-1), L.reshape(-1)]))
        trajs.append(tc.clone())
        ens.append(et.clone())
        if dynamic:
            assert 3 <= integ.last_recoveries() <= 64, integ.last_recoveries()
        else:
            assert integ.last_recoveries() == 0
    torch.testing.assert_close(outs[0], outs[1], rtol=0, atol=1e-9)
    torch.testing.assert_close(trajs[0], trajs[1], rtol=0, atol=1e-9)
    torch.testing.assert_close(ens[0], ens[1], rtol=1e-9, atol=1e-9)


def test_hopeless_neighbour_policy_is_an_error_not_a_crawl():
    from mythos_amd.hip_system import LangevinIntegrator

    top, c0, q0 = generators.ideal_duplex(16, model=2, seed=3)
    s, _ = _make(2, top, None, torch.float32, hce=True)
    integ = LangevinIntegrator(s, dt=0.005, kT=KT, gamma_t=KT / 2.5, gamma_r=KT / 7.5, seed=1)
    integ.set_neighbor_policy(r_cut=3.3, skin=0.002, every=1000)  # violated by every step
    c, q = _state(c0, q0, torch.float32, s.device)
    p, L = integ.init_momenta()
    with pytest.raises(_lib.MythosHipError, match="skin"):
        integ.run(c, q, p, L, 200)


def test_fp32_dynamics_do_not_depend_on_where_the_system_sits():
    """fp32 runs carry the centre as hi + lo: a duplex 4 096 length units from the origin (one fp32 ulp of a
    coordinate there is 4.9e-4, a third of the thermal displacement per step) must evolve like the same duplex at
    the origin.  Momenta are compared: they feel every force error, and no output rounding of large coordinates."""
    from mythos_amd.hip_system import LangevinIntegrator

    top, traj, _, _ = H.load_golden(2, "simple-helix")
    s, sim = _make(2, top, None, torch.float32)
    s.set_neighbors(top.unbonded_neighbors)
    grid = 2.0**-11  # representable at both places
    c0 = np.round(traj.center[0] / grid) * grid
    q0 = traj.quaternions[0]
    shift = np.array([4096.0, 0.0, 0.0])

    def run(offset, chunks):
        integ = LangevinIntegrator(s, dt=0.005, kT=KT, gamma_t=KT / 2.5, gamma_r=KT / 7.5, seed=9)
        c, q = _state(c0 + offset, q0, torch.float32, s.device)
        p, L = integ.init_momenta()
        for n in chunks:
            integ.run(c, q, p, L, n)
        return c.cpu().numpy() - offset, p.cpu().numpy(), L.cpu().numpy()

    x_a, p_a, l_a = run(np.zeros(3), [60])
    x_b, p_b, l_b = run(shift, [60])
    scale = np.abs(p_a).max()
    assert np.abs(p_b - p_a).max() < 2e-4 * scale and np.abs(l_b - l_a).max() < 2e-4 * np.abs(l_a).max()
    assert np.abs(x_b - x_a).max() <= 0.5 * 4.9e-4 + 1e-5  # the caller's fp32 copy is rounded, the state is not
    # a run cut in two keeps the low parts across the seam (the caller's array is unchanged in between); what is
    # left is the re-normalisation of the quaternions on entry, one ulp - dropping the low parts would show up
    # at 1e-3
    x_c, p_c, l_c = run(shift, [30, 30])
    assert np.abs(p_c - p_b).max() < 2e-5 * scale and np.abs(l_c - l_b).max() < 2e-5 * np.abs(l_a).max()
