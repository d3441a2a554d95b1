"""The MD step kernel (md_step_kernel, mythos_amd/csrc/langevin_core.inc) against a second implementation at the sizes it is
benchmarked at, and for every instantiation that ships.

 * 1.1 kbp, fp64, device-built cell list + chunk order: five steps against oracle/langevin_oracle.py on the same
   Philox stream (1e-9);
 * 12 kbp, fp32 (hi + lo centres, Morton chunk order, XCD remap, close / far segments by site distance): one
   frictionless step from rest against the SAME integrator arithmetic in numpy fp64 fed with the forces and torques
   of mythos_oxdna_energy (which tests/test_gpu_full_size.py holds to the oracle at 1 kbp) - 1e-3;
 * oxDNA1 (md_step_kernel<., 1, .>: dna1 backbone, coaxial stacking with f5(cos phi3/4) in the angular pass): six
   steps in fp64 against the oracle on dna1/simple-helix and dna1/simple-coax, then BASELINE configs[0] as written
   (oxDNA1 16 bp = 32 nt duplex, 1 000 NVT steps) held to equipartition;
 * BASELINE configs[4] at size: oxDNA2 32 bp (64 nt) x 64 replicas in one launch per step, then map + dU/dtheta of
   frames of the stored trajectory against the oracle;
 * halt-and-resume on a grid several times what is resident at once (12 kbp, fp64, energy trace: one workgroup per
   CU) against a static list;
 * the resident form load / advance / advance / store == one run, bit for bit.
"""

import numpy as np
import pytest
import torch

from mythos_amd import _lib
from mythos_amd.energy import flat_params as fp
from mythos_amd.input import defaults
from mythos_amd.simulators.neighbors import verlet_pairs_numpy
from mythos_amd.utils import generators
from tests import helpers as H

pytestmark = pytest.mark.gpu

KT = 296.15 * 0.1 / 300.0
R_CUT = 3.25


def _system(model, top, dtype, box=None, hce=True):
    from mythos_amd.hip_system import OxdnaSystem

    sim, cfg = defaults.default_configs_for(f"dna{model}")
    flat = fp.pack_flat(fp.derive_flat(model, cfg, kt=sim["kT"], salt_conc=0.5, half_charged_ends=hce), _lib.param_names())
    s = OxdnaSystem(model, top.seq, top.is_end, top.bonded_neighbors, box=box, dtype=dtype)
    s.set_params(flat)
    return s


def _dev(a, dtype, s):
    return torch.as_tensor(a, dtype=dtype, device=s.device).contiguous()


def _top_tensors(top, pairs):
    return (torch.as_tensor(top.seq, dtype=torch.long), torch.as_tensor(top.is_end, dtype=torch.long),
            torch.as_tensor(top.bonded_neighbors, dtype=torch.long), torch.as_tensor(pairs, dtype=torch.long))


def test_1kbp_fp64_device_list_steps_match_oracle(md_lanes):
    from mythos_amd.hip_system import LangevinIntegrator
    from oracle.langevin_oracle import LangevinOracle

    top, c0, q0 = generators.ideal_duplex(1100, model=2, seed=1234)  # 2 200 nt = 69 chunks: the chunk order is in use
    rng = np.random.default_rng(5)
    c0 = c0 + 0.03 * rng.standard_normal(c0.shape)
    q0 = q0 + 0.015 * rng.standard_normal(q0.shape)
    q0 /= np.linalg.norm(q0, axis=1, keepdims=True)
    s = _system(2, top, torch.float64)
    gam_t, gam_r, seed = KT / 2.5, KT / 7.5, 0xABCDEF12345
    integ = LangevinIntegrator(s, dt=0.005, kT=KT, gamma_t=gam_t, gamma_r=gam_r, mass=1.0, inertia=(1.0, 1.2, 0.9), seed=seed)
    integ.set_neighbor_policy(R_CUT, 0.6, 3)  # rebuilds inside the five steps
    c, q = _dev(c0, torch.float64, s), _dev(q0, torch.float64, s)
    p, L = integ.init_momenta()
    x, qq, pp, LL = (t.cpu().numpy().copy() for t in (c, q, p, L))
    n_steps = 5
    tc, tq, et = integ.run(c, q, p, L, n_steps, save_every=1)
    pairs = verlet_pairs_numpy(c0, top.bonded_neighbors, R_CUT + 0.6)  # a superset of every pair inside a cut-off
    orc = LangevinOracle(2, H.oracle_params(2, half_charged_ends=True), _top_tensors(top, pairs), None, 0.005, KT, gam_t, gam_r, 1.0,
                         (1.0, 1.2, 0.9), seed=seed)
    for k in range(n_steps):
        x, qq, pp, LL, u = orc.step(x, qq, pp, LL)
        np.testing.assert_allclose(tc[k].cpu().numpy(), x, rtol=0, atol=1e-9)
        np.testing.assert_allclose(tq[k].cpu().numpy(), qq, rtol=0, atol=1e-9)
        assert abs(et[k, :8].sum().item() - u) < 1e-9 * abs(u)
    np.testing.assert_allclose(p.cpu().numpy(), pp, atol=1e-9)
    np.testing.assert_allclose(L.cpu().numpy(), LL, atol=1e-9)


def _body_torque(q, gq):
    from oracle import oxdna_oracle as orc

    return orc.quat_grad_to_body_torque(torch.as_tensor(q), torch.as_tensor(gq)).numpy()


@pytest.mark.parametrize("bp", [12000, 17000])
def test_12kbp_fp32_step_matches_energy_kernel_forces(bp):
    """One step from rest, no friction, no noise: p1 = dt/2 (F(x0) + F(x1)), x1 = x0 + dt^2/2 F(x0), the rotor likewise.
    The reference arithmetic: oracle/langevin_oracle.py's drift on forces from the fp64 energy kernel.
    17 kbp: 1 063 workgroups, more than four per CU of a 256-CU device - the DENSE instantiation of the step kernel."""
    from mythos_amd.hip_system import LangevinIntegrator
    from oracle.langevin_oracle import drift

    top, c0, q0 = generators.ideal_duplex(bp, model=2, seed=1234)
    rng = np.random.default_rng(11)
    c0 = c0 + 0.015 * rng.standard_normal(c0.shape)  # off the symmetric point: forces of order 10, a few of order 100
    q0 = q0 + 0.0075 * rng.standard_normal(q0.shape)
    q0 /= np.linalg.norm(q0, axis=1, keepdims=True)
    c0 = c0.astype(np.float32).astype(np.float64)   # the configuration both precisions see
    q0 = q0.astype(np.float32).astype(np.float64)
    s64 = _system(2, top, torch.float64)

    def forces(x, q):
        cd, qd = _dev(x, torch.float64, s64), _dev(q, torch.float64, s64)
        s64.build_neighbors(cd, R_CUT, 0.0)
        _, gc, gq, _ = s64.energy(cd, qd, grads=True)
        qn = q / np.linalg.norm(q, axis=1, keepdims=True)
        return -gc.cpu().numpy().reshape(-1, 3), _body_torque(qn, gq.cpu().numpy().reshape(-1, 4))

    dt, inertia = 0.005, np.array([1.0, 1.0, 1.0])
    # the MD kernel normalises the quaternion on entry
    qn = q0 / np.linalg.norm(q0, axis=1, keepdims=True)
    F0, t0 = forces(c0, qn)
    p, L = 0.5 * dt * F0, 0.5 * dt * t0
    x, q, L = drift(c0, qn, p, L, 0.5 * dt, 1.0, inertia)
    x, q, L = drift(x, q, p, L, 0.5 * dt, 1.0, inertia)
    q = q / np.linalg.norm(q, axis=1, keepdims=True)
    F1, t1 = forces(x, q)
    p_ref, L_ref = p + 0.5 * dt * F1, L + 0.5 * dt * t1

    s32 = _system(2, top, torch.float32)
    integ = LangevinIntegrator(s32, dt=dt, kT=KT, gamma_t=0.0, gamma_r=0.0, mass=1.0, inertia=(1.0, 1.0, 1.0), seed=1)
    integ.set_neighbor_policy(R_CUT, 0.6, 25)
    c, qd = _dev(c0, torch.float32, s32), _dev(q0, torch.float32, s32)
    pz, Lz = torch.zeros_like(c), torch.zeros_like(c)
    integ.run(c, qd, pz, Lz, 1)
    pg, Lg = pz.cpu().double().numpy(), Lz.cpu().double().numpy()
    # per nucleotide: 1e-3 of its own momentum plus 1e-3 of the typical one (not of the largest: the steepest
    # excluded-volume contact of 24 000 nucleotides would hide everything else)
    for got, ref in ((pg, p_ref), (Lg, L_ref)):
        rms = np.sqrt((ref**2).mean())
        assert rms > 0.005 and np.abs(ref).max() < 200 * rms  # there is something to compare, and no singular contact
        err = np.abs(got - ref).max(1)
        assert (err <= 1e-3 * (np.abs(ref).max(1) + rms)).all(), (err.max(), rms, np.abs(ref).max())
    # positions moved by dt^2/2 F ~ 1e-4: compare the DISPLACEMENT, which the hi + lo centres resolve
    # (the caller's fp32 copy of a centre is rounded to its ulp: look where that ulp is small, |coordinate| < 32)
    near = np.abs(c0).max(1) < 32.0
    dx = (c.cpu().double().numpy() - c0)[near]
    assert near.sum() > 100 and np.abs(dx - (x - c0)[near]).max() <= 1e-3 * np.abs(x - c0).max() + 2e-6
    assert np.abs(qd.cpu().double().numpy() - q).max() <= 1e-3 * np.abs(q - qn).max() + 3e-7


def test_12kbp_fp64_step_matches_energy_kernel_forces():
    """The same step in fp64: 750 workgroups, more than two per CU, so this is the stepping instantiation with the
    three-per-CU register bound (168 VGPRs + scratch) and the pooled result rows - the one `config.f64` of the bench
    measures.  Held to 1e-9 of the forces of the fp64 energy kernel (itself held to the oracle at 1 kbp)."""
    from mythos_amd.hip_system import LangevinIntegrator
    from oracle.langevin_oracle import drift

    top, c0, q0 = generators.ideal_duplex(12000, model=2, seed=1234)
    rng = np.random.default_rng(12)
    c0 = c0 + 0.015 * rng.standard_normal(c0.shape)
    q0 = q0 + 0.0075 * rng.standard_normal(q0.shape)
    q0 /= np.linalg.norm(q0, axis=1, keepdims=True)
    s = _system(2, top, torch.float64)

    def forces(x, q):
        cd, qd = _dev(x, torch.float64, s), _dev(q, torch.float64, s)
        s.build_neighbors(cd, R_CUT, 0.0)
        _, gc, gq, _ = s.energy(cd, qd, grads=True)
        return -gc.cpu().numpy().reshape(-1, 3), _body_torque(q, gq.cpu().numpy().reshape(-1, 4))

    dt, inertia = 0.005, np.array([1.0, 1.1, 0.9])
    F0, t0 = forces(c0, q0)
    p, L = 0.5 * dt * F0, 0.5 * dt * t0
    x, q, L = drift(c0, q0, p, L, 0.5 * dt, 1.0, inertia)
    x, q, L = drift(x, q, p, L, 0.5 * dt, 1.0, inertia)
    q = q / np.linalg.norm(q, axis=1, keepdims=True)
    F1, t1 = forces(x, q)
    p_ref, L_ref = p + 0.5 * dt * F1, L + 0.5 * dt * t1
    integ = LangevinIntegrator(s, dt=dt, kT=KT, gamma_t=0.0, gamma_r=0.0, mass=1.0, inertia=tuple(inertia), seed=1)
    integ.set_neighbor_policy(R_CUT, 0.6, 25)
    c, qd = _dev(c0, torch.float64, s), _dev(q0, torch.float64, s)
    pz, Lz = torch.zeros_like(c), torch.zeros_like(c)
    integ.run(c, qd, pz, Lz, 1)
    assert integ.last_recoveries() == 0  # the pool of 320 rows holds a thermal duplex: no abort, this IS the small instantiation
    for got, ref in ((pz, p_ref), (Lz, L_ref)):
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=0, atol=1e-9 * max(1.0, np.abs(ref).max()))
    np.testing.assert_allclose(c.cpu().numpy(), x, rtol=0, atol=1e-11)
    np.testing.assert_allclose(qd.cpu().numpy(), q, rtol=0, atol=1e-11)


def test_12kbp_two_fp64_steps_and_forces_match_the_oracle_directly():
    """the 12 kbp checks above go through the HIP energy kernel; this one does not.  Two thermostatted fp64
    steps of the 24 000-nt duplex against LangevinOracle on the same Philox stream (positions, quaternions, potential
    energy: 1e-9), and dU/dcentre, dU/dquat of mythos_oxdna_energy against oracle autograd at the same size (fp64 1e-5,
    fp32 1e-3 of the largest component; per-term energies likewise)."""
    from mythos_amd.hip_system import LangevinIntegrator
    from oracle import oxdna_oracle as orc
    from oracle.langevin_oracle import LangevinOracle

    top, c0, q0 = generators.ideal_duplex(12000, model=2, seed=1234)
    rng = np.random.default_rng(21)
    c0 = c0 + 0.02 * rng.standard_normal(c0.shape)
    q0 = q0 + 0.01 * rng.standard_normal(q0.shape)
    q0 /= np.linalg.norm(q0, axis=1, keepdims=True)
    # the configuration every precision sees: fp32-representable (a 12 kbp duplex is 4 800 units long, where an fp32
    # coordinate resolves 5e-4 - rounding the INPUT differently for the two sides would be an error of the test)
    c0 = c0.astype(np.float32).astype(np.float64)
    q0 = q0.astype(np.float32).astype(np.float64)
    pairs = verlet_pairs_numpy(c0, top.bonded_neighbors, R_CUT + 0.6)
    P = H.oracle_params(2, half_charged_ends=True)
    tt = _top_tensors(top, pairs)
    # ---- energies and gradients of the energy kernel, both precisions
    e_ref = orc.energy_terms(2, P, torch.as_tensor(c0), torch.as_tensor(q0), *tt, box=None).numpy()
    _, gc_ref, gq_ref = orc.energy_and_grads(2, P, torch.as_tensor(c0), torch.as_tensor(q0), *tt, box=None)
    gc_ref, gq_ref = gc_ref.numpy(), gq_ref.numpy()
    for dtype, tol in ((torch.float64, 1e-5), (torch.float32, 1e-3)):
        s = _system(2, top, dtype)
        cd, qd = _dev(c0, dtype, s), _dev(q0, dtype, s)
        s.build_neighbors(cd, R_CUT, 0.0)
        e, gc, gq, _ = s.energy(cd, qd, grads=True)
        e = e.cpu().numpy().reshape(-1)[:8]
        assert np.abs(e - e_ref).max() <= tol * np.abs(e_ref).max(), (dtype, e, e_ref)
        assert np.abs(gc.cpu().double().numpy().reshape(-1, 3) - gc_ref).max() <= tol * np.abs(gc_ref).max()
        gq = gq.cpu().double().numpy().reshape(-1, 4)
        if dtype == torch.float64:
            assert np.abs(gq - gq_ref).max() <= tol * np.abs(gq_ref).max()
        else:
            # fp32: the part of dU/dq that acts - the body torque -1/2 (P_k q) . dU/dq.  The component ALONG q (a change
            # of |q|, no torque) is not comparable at this size: the reference's clamp in front of acos has zero slope
            # where it clips, and among 24 000 nucleotides a few pairs of nearly (anti)parallel axes have a cosine that
            # rounds to exactly +-1 in fp32 but not in fp64 - their whole contribution there is along q (measured:
            # differences of 5 in components of 35, the torques agreeing to 1e-5)
            tq_got, tq_ref = _body_torque(q0, gq), _body_torque(q0, gq_ref)
            assert np.abs(tq_got - tq_ref).max() <= tol * np.abs(tq_ref).max(), (np.abs(tq_got - tq_ref).max(), np.abs(tq_ref).max())
    # ---- two steps of the stepping kernel (fp64, the three-per-CU instantiation: 750 workgroups)
    s = _system(2, top, torch.float64)
    gam_t, gam_r, seed = KT / 2.5, KT / 7.5, 0x5EED12
    integ = LangevinIntegrator(s, dt=0.005, kT=KT, gamma_t=gam_t, gamma_r=gam_r, mass=1.0, inertia=(1.0, 1.1, 0.9), seed=seed)
    integ.set_neighbor_policy(R_CUT, 0.6, 25)
    c, q = _dev(c0, torch.float64, s), _dev(q0 / np.linalg.norm(q0, axis=1, keepdims=True), torch.float64, s)
    p, L = integ.init_momenta()
    x, qq, pp, LL = (t.cpu().numpy().copy() for t in (c, q, p, L))
    tc, tq, et = integ.run(c, q, p, L, 2, save_every=1)
    lo = LangevinOracle(2, P, tt, None, 0.005, KT, gam_t, gam_r, 1.0, (1.0, 1.1, 0.9), seed=seed)
    for k in range(2):
        x, qq, pp, LL, u = lo.step(x, qq, pp, LL)
        np.testing.assert_allclose(tc[k].cpu().numpy(), x, rtol=0, atol=1e-9)
        np.testing.assert_allclose(tq[k].cpu().numpy(), qq, rtol=0, atol=1e-9)
        assert abs(et[k, :8].sum().item() - u) < 1e-9 * abs(u)
    np.testing.assert_allclose(p.cpu().numpy(), pp, atol=1e-9 * max(1.0, np.abs(pp).max()))
    np.testing.assert_allclose(L.cpu().numpy(), LL, atol=1e-9 * max(1.0, np.abs(LL).max()))


def test_12kbp_fp32_step_matches_oracle_forces():
    """The fp32 stepping kernel at 12 kbp against forces and torques of the ORACLE (autograd of the pinned energies), not
    of the HIP energy kernel: one frictionless step from rest, per nucleotide 1e-3 of its own momentum + 1e-3 of the rms."""
    from mythos_amd.hip_system import LangevinIntegrator
    from oracle import oxdna_oracle as orc
    from oracle.langevin_oracle import drift

    top, c0, q0 = generators.ideal_duplex(12000, model=2, seed=1234)
    rng = np.random.default_rng(13)
    c0 = c0 + 0.015 * rng.standard_normal(c0.shape)
    q0 = q0 + 0.0075 * rng.standard_normal(q0.shape)
    q0 /= np.linalg.norm(q0, axis=1, keepdims=True)
    c0 = c0.astype(np.float32).astype(np.float64)
    q0 = q0.astype(np.float32).astype(np.float64)
    qn = q0 / np.linalg.norm(q0, axis=1, keepdims=True)
    P = H.oracle_params(2, half_charged_ends=True)
    tt = _top_tensors(top, verlet_pairs_numpy(c0, top.bonded_neighbors, R_CUT + 0.3))

    def forces(x, q):
        _, gc, gq = orc.energy_and_grads(2, P, torch.as_tensor(x), torch.as_tensor(q), *tt, box=None)
        return -gc.numpy(), _body_torque(q, gq.numpy())

    dt, inertia = 0.005, np.array([1.0, 1.0, 1.0])
    F0, t0 = forces(c0, qn)
    p, L = 0.5 * dt * F0, 0.5 * dt * t0
    x, q, L = drift(c0, qn, p, L, 0.5 * dt, 1.0, inertia)
    x, q, L = drift(x, q, p, L, 0.5 * dt, 1.0, inertia)
    q = q / np.linalg.norm(q, axis=1, keepdims=True)
    F1, t1 = forces(x, q)
    p_ref, L_ref = p + 0.5 * dt * F1, L + 0.5 * dt * t1
    s32 = _system(2, top, torch.float32)
    integ = LangevinIntegrator(s32, dt=dt, kT=KT, gamma_t=0.0, gamma_r=0.0, mass=1.0, inertia=(1.0, 1.0, 1.0), seed=1)
    integ.set_neighbor_policy(R_CUT, 0.6, 25)
    c, qd = _dev(c0, torch.float32, s32), _dev(q0, torch.float32, s32)
    pz, Lz = torch.zeros_like(c), torch.zeros_like(c)
    integ.run(c, qd, pz, Lz, 1)
    for got, ref in ((pz.cpu().double().numpy(), p_ref), (Lz.cpu().double().numpy(), L_ref)):
        rms = np.sqrt((ref**2).mean())
        assert rms > 0.005 and np.abs(ref).max() < 200 * rms
        err = np.abs(got - ref).max(1)
        assert (err <= 1e-3 * (np.abs(ref).max(1) + rms)).all(), (err.max(), rms, np.abs(ref).max())
    assert np.abs(qd.cpu().double().numpy() - q).max() <= 1e-3 * np.abs(q - qn).max() + 3e-7


@pytest.mark.parametrize("bp", [50, 1100])
def test_rebuild_overflow_in_front_of_the_first_launch_of_a_later_segment_resumes_there(bp):
    """the device's progress word is cleared after every segment of queued launches, so when the FIRST
    launch of a later segment halts (the scheduled rebuild in front of it overflowed) the word says 0 - and the host used
    to take that for "nothing ran" and start the call again from launch 0 on the wrong frame.  Forced with the library's
    test switches: segments of 8 launches, rebuild every 8 steps, the rebuild in front of launch 16 claims an overflow.
    The recovered run must be the undisturbed run, bit for bit, with one recovery and the full step count."""
    from mythos_amd.hip_system import LangevinIntegrator

    top, c0, q0 = generators.ideal_duplex(bp, model=2, seed=77)
    rng = np.random.default_rng(8)
    c0 = c0 + 0.02 * rng.standard_normal(c0.shape)
    q0 = q0 + 0.01 * rng.standard_normal(q0.shape)
    q0 /= np.linalg.norm(q0, axis=1, keepdims=True)
    results = []
    for inject in (False, True):
        s = _system(2, top, torch.float64)
        integ = LangevinIntegrator(s, dt=0.005, kT=KT, gamma_t=KT / 2.5, gamma_r=KT / 7.5, seed=5)
        integ.set_neighbor_policy(R_CUT, 0.6, 8)
        c, q = _dev(c0, torch.float64, s), _dev(q0, torch.float64, s)
        p, L = integ.init_momenta()
        _lib.debug_set("md_segment", 8)
        if inject:
            _lib.debug_set("md_overflow_at", 17)
        try:
            tc, tq, et = integ.run(c, q, p, L, 40, save_every=4)
        finally:
            _lib.debug_set("md_segment", 0)
            _lib.debug_set("md_overflow_at", 0)
        results.append((c.clone(), q.clone(), p.clone(), L.clone(), tc.clone(), et.clone(), integ.last_recoveries(), integ.step))
    plain, hit = results
    assert plain[6] == 0 and hit[6] == 1 and plain[7] == hit[7] == 40
    for a, b in zip(plain[:6], hit[:6]):
        assert torch.equal(a, b)


def _nicked_duplex(model):
    """16 bp duplex whose second strand is two 8-mers: the two bases either side of the nick stack coaxially.  (The
    reference's dna1/simple-coax trajectory has a non-zero coaxial term only in its initial configuration, which is
    not among the 100 stored frames.)"""
    from mythos_amd.input import topology as jd_top

    top0, c, q = generators.ideal_duplex(16, model=model, seed=8)
    top = jd_top.from_arrays(np.asarray(top0.seq, dtype=np.int32), [16, 8, 8])
    rng = np.random.default_rng(2)
    c = c + 0.02 * rng.standard_normal(c.shape)
    q = q + 0.01 * rng.standard_normal(q.shape)
    return top, c, q / np.linalg.norm(q, axis=1, keepdims=True), None


@pytest.mark.parametrize("name", ["simple-helix", "simple-coax", "nicked-duplex"])
def test_oxdna1_md_steps_match_oracle_fp64(name):
    from mythos_amd.hip_system import LangevinIntegrator
    from oracle.langevin_oracle import LangevinOracle

    if name == "nicked-duplex":
        top, c0, q0, box = _nicked_duplex(1)
    else:
        top, traj, _, _ = H.load_golden(1, name)
        c0, q0, box = traj.center[3], traj.quaternions[3], traj.box_size
    s = _system(1, top, torch.float64, box=box, hce=False)
    s.set_neighbors(top.unbonded_neighbors)
    gam_t, gam_r, seed = KT / 2.5, KT / 7.5, 0x51DE
    integ = LangevinIntegrator(s, dt=0.005, kT=KT, gamma_t=gam_t, gamma_r=gam_r, mass=1.0, inertia=(1.0, 1.1, 0.95), seed=seed)
    c, q = _dev(c0, torch.float64, s), _dev(q0, torch.float64, s)
    p, L = integ.init_momenta()
    x, qq, pp, LL = (t.cpu().numpy().copy() for t in (c, q, p, L))
    tc, tq, et = integ.run(c, q, p, L, 6, save_every=1)
    orc = LangevinOracle(1, H.oracle_params(1), H.topo_tensors(top), box, 0.005, KT, gam_t, gam_r, 1.0, (1.0, 1.1, 0.95), seed=seed)
    for k in range(6):
        x, qq, pp, LL, u = orc.step(x, qq, pp, LL)
        np.testing.assert_allclose(tc[k].cpu().numpy(), x, rtol=0, atol=1e-10)
        np.testing.assert_allclose(tq[k].cpu().numpy(), qq, rtol=0, atol=1e-10)
        assert abs(et[k, :8].sum().item() - u) < 1e-8 * abs(u)
        assert et[k, 7].item() == 0.0  # no Debye-Hueckel term in oxDNA1
    if name == "nicked-duplex":
        assert et[:, 6].abs().max().item() > 0.1  # coaxial stacking across the nick (f5 of cos phi3 / phi4 live)
    np.testing.assert_allclose(p.cpu().numpy(), pp, atol=1e-9)
    np.testing.assert_allclose(L.cpu().numpy(), LL, atol=1e-9)


def test_config0_oxdna1_16bp_1000_nvt_steps():
    """BASELINE configs[0]: oxDNA1 16 bp duplex, 1 000 NVT steps at dt 0.005, gamma = kT / 2.5 and kT / 7.5 (the
    reference's JAX-MD CPU case; here on the GPU in both precisions).  The ideal helix is thermalised first with a
    strong bath (the configured friction needs 5 000 - 15 000 steps to do that), then the 1 000 steps are held to
    what the thermostat guarantees: equipartition of both kinetic energies, bounded potential, intact helix."""
    from mythos_amd.hip_system import LangevinIntegrator

    top, c0, q0 = generators.ideal_duplex(16, model=1, seed=3)
    assert top.n_nucleotides == 32
    for dtype in (torch.float64, torch.float32):
        s = _system(1, top, dtype, hce=False)
        s.set_neighbors(top.unbonded_neighbors)
        ke, us = [], []
        for seed in range(12):  # twelve independent runs: 32 nucleotides are a small thermometer
            c, q = _dev(c0, dtype, s), _dev(q0, dtype, s)
            warm = LangevinIntegrator(s, dt=0.005, kT=KT, gamma_t=2.0, gamma_r=2.0, seed=1000 + seed)
            p, L = warm.init_momenta()
            warm.run(c, q, p, L, 3000)
            integ = LangevinIntegrator(s, dt=0.005, kT=KT, gamma_t=KT / 2.5, gamma_r=KT / 7.5, seed=seed)
            _, _, et = integ.run(c, q, p, L, 1000, save_every=50)
            et = et.cpu().numpy()
            assert np.isfinite(et).all() and integ.step == 1000
            ke.append(et[:, 8:].mean(0))
            us.append(et[:, :8].sum(1).mean() / 32)
            assert et[:, 4].max() < -4.0  # H-bond energy: the helix stays closed (sixteen pairs, about -0.7 each)
            assert np.allclose(q.norm(dim=1).cpu().numpy(), 1.0, atol=1e-5)
        ke = np.mean(ke, axis=0) / (1.5 * 32 * KT)
        # 12 runs x 96 degrees of freedom: one sigma of the mean is 0.042 even if the 20 frames of a run were one sample
        assert abs(ke[0] - 1.0) < 0.13 and abs(ke[1] - 1.0) < 0.13, ke
        assert -1.60 < np.mean(us) < -1.15, us  # oxDNA1 duplex at 296 K (the 16-nt golden run: -1.35 ... -1.44)


def test_config4_64_replicas_of_32bp_map_and_param_grads_match_oracle():
    """BASELINE configs[4] at size: 64 replicas of an oxDNA2 32 bp duplex advance in one launch per step; the stored
    frames then go through map + dU/dtheta (the DiffTRe data path) and are held to the oracle frame by frame."""
    from mythos_amd.energy import dna2
    from mythos_amd.energy.base import Quaternion, RigidBody, space
    from mythos_amd.simulators.hip_md import HipMDSimulator, StaticSimulatorParams, nvt_langevin
    from mythos_amd.simulators.neighbors import NoNeighborList
    from oracle import oxdna_oracle as orc

    top, c0, q0 = generators.ideal_duplex(32, model=2, seed=21)
    n = top.n_nucleotides
    assert n == 64
    disp, shift = space.free()
    ef = dna2.create_default_energy_fn(topology=top, displacement_fn=disp)
    dev = torch.device("cuda", 0)
    init = RigidBody(center=torch.as_tensor(c0, device=dev), orientation=Quaternion(vec=torch.as_tensor(q0, device=dev)))
    sp = StaticSimulatorParams(seq=top.seq, mass=(1.0, (1.0, 1.0, 1.0)), gamma=(KT / 2.5, KT / 7.5),
                               bonded_neighbors=top.bonded_neighbors, checkpoint_every=0, dt=0.005, kT=KT)
    sim = HipMDSimulator(energy_fn=ef, simulator_params=sp, space=(disp, shift), simulator_init=nvt_langevin,
                         neighbors=NoNeighborList(unbonded_nbrs=top.unbonded_neighbors), save_every=100, dtype=torch.float64,
                         n_replicas=64)
    out = sim.run({}, init, 300, key=17)
    traj = out.observables[0]
    assert traj.center.shape == (64 * 3, n, 3)
    e_all = ef.map(traj)
    assert e_all.shape == (192,) and torch.isfinite(e_all).all()
    assert e_all.std() > 1e-3  # 64 different noise streams
    # three frames of three different replicas against the oracle: energy and the gradient with respect to the
    # parameters DiffTRe optimises in the reference's examples
    opt = {"eps_stack_base": 1.3523, "eps_hb": 1.0678, "k_cross": 47.5, "q_eff": 0.815}
    leaves = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in opt.items()}
    P = H.oracle_params(2, half_charged_ends=True, overrides={
        "stacking": {"eps_stack_base": leaves["eps_stack_base"]}, "hydrogen_bonding": {"eps_hb": leaves["eps_hb"]},
        "cross_stacking": {"k_cross": leaves["k_cross"]}, "debye": {"q_eff": leaves["q_eff"]}})
    tt = H.topo_tensors(top)
    frames = [2, 3 * 31 + 1, 3 * 63 + 2]
    sub = RigidBody(center=traj.center[frames], orientation=Quaternion(vec=traj.orientation.vec[frames]))
    par = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in opt.items()}
    e_hip = ef.with_params(par).map(sub)
    for j, f in enumerate(frames):
        e_o = orc.energy(2, P, traj.center[f].cpu(), traj.orientation.vec[f].cpu(), *tt, None)
        assert abs(e_hip[j].item() - e_o.item()) <= 1e-9 * abs(e_o.item())
        g_o = torch.autograd.grad(e_o, list(leaves.values()), retain_graph=True)
        g_h = torch.autograd.grad(e_hip[j], list(par.values()), retain_graph=True)
        for k, a, b in zip(opt, g_h, g_o):
            assert abs(a.item() - b.item()) <= 1e-7 * max(1.0, abs(b.item())), (f, k, a.item(), b.item())


def test_halt_and_resume_on_a_grid_larger_than_the_chip():
    """12 kbp in fp64 with an energy trace runs one workgroup per CU: 750 workgroups on 256 CUs, three rounds per launch.
    A workgroup of the launch that raised the halt word must not mistake it for its own: compared with a static list
    that never halts, the trajectory is the same (the halt word carries the launch index)."""
    from mythos_amd.hip_system import LangevinIntegrator

    top, c0, q0 = generators.ideal_duplex(12000, model=2, seed=1234)
    s = _system(2, top, torch.float64)
    outs = []
    for dynamic in (False, True):
        integ = LangevinIntegrator(s, dt=0.005, kT=KT, gamma_t=KT / 2.5, gamma_r=KT / 7.5, seed=9)
        c, q = _dev(c0, torch.float64, s), _dev(q0, torch.float64, s)
        if dynamic:
            integ.set_neighbor_policy(R_CUT, 0.05, 10_000)  # no scheduled rebuild: every rebuild is a halt
        else:
            s.set_neighbors(verlet_pairs_numpy(c0, top.bonded_neighbors, R_CUT + 1.0))
        p, L = integ.init_momenta()
        _, _, et = integ.run(c, q, p, L, 40, save_every=1)
        outs.append((c.cpu().numpy(), q.cpu().numpy(), p.cpu().numpy(), et.cpu().numpy()))
        if dynamic:
            assert integ.last_recoveries() >= 2
    for a, b in zip(outs[0][:3], outs[1][:3]):
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-9)
    np.testing.assert_allclose(outs[0][3], outs[1][3], rtol=1e-9, atol=1e-7)


def test_resident_advances_equal_one_run_bitwise():
    from mythos_amd.hip_system import LangevinIntegrator

    top, c0, q0 = generators.ideal_duplex(1500, model=2, seed=1234)
    s = _system(2, top, torch.float32)
    res = []
    for plan in ("run", "resident"):
        integ = LangevinIntegrator(s, dt=0.005, kT=KT, gamma_t=KT / 2.5, gamma_r=KT / 7.5, seed=5)
        integ.set_neighbor_policy(R_CUT, 0.6, 25)
        c, q = _dev(c0, torch.float32, s), _dev(q0, torch.float32, s)
        p, L = integ.init_momenta()
        if plan == "run":
            integ.run(c, q, p, L, 90)
        else:
            with pytest.raises(_lib.MythosHipError, match="resident"):
                integ.advance(1)
            integ.load(c, q, p, L)
            integ.advance(17)
            integ.advance(0)
            _, _, et = integ.advance(73, save_every=73)
            assert et.shape == (1, 10) and torch.isfinite(et).all()
            integ.store(c, q, p, L)
        assert integ.step == 90
        res.append([t.clone() for t in (c, q, p, L)])
    for a, b in zip(*res):
        assert torch.equal(a, b)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_advance_leaves_the_frame_open_and_store_closes_it(dtype, md_lanes):
    """mythos_langevin_advance(n) is n launches: the closing half kick of step n comes with the next force evaluation -
    the next advance's first launch, or one launch inside mythos_langevin_store.  Whatever the sequence of calls, the
    state handed back is the one mythos_langevin_run(total) hands back, bit for bit; a second store changes nothing."""
    from mythos_amd.hip_system import LangevinIntegrator

    top, c0, q0 = generators.ideal_duplex(400, model=2, seed=77)
    s = _system(2, top, dtype)

    def fresh():
        integ = LangevinIntegrator(s, dt=0.005, kT=KT, gamma_t=KT / 2.5, gamma_r=KT / 7.5, seed=9)
        integ.set_neighbor_policy(R_CUT, 0.6, 25)
        integ.set_timing(1)
        return integ, _dev(c0, dtype, s), _dev(q0, dtype, s)

    integ, c, q = fresh()
    p, L = integ.init_momenta()
    integ.run(c, q, p, L, 60)
    assert integ.last_kernel_ms()["launches"] == 61
    want = [t.clone() for t in (c, q, p, L)]

    integ, c, q = fresh()
    p, L = integ.init_momenta()
    integ.load(c, q, p, L)
    _lib.debug_set("md_segment", 7)  # segments of seven launches: the open frame crosses segment ends as well
    try:
        integ.advance(20)
    finally:
        _lib.debug_set("md_segment", 0)
    assert integ.last_kernel_ms()["launches"] == 20
    out = [torch.empty_like(t) for t in (c, q, p, L)]
    integ.store(*out)  # closes: one launch, no step
    assert integ.last_kernel_ms()["launches"] == 1 and integ.step == 20
    again = [torch.empty_like(t) for t in (c, q, p, L)]
    integ.store(*again)  # already closed: a copy
    for a, b in zip(out, again):
        assert torch.equal(a, b)
    integ.advance(15)  # from a closed frame
    integ.advance(5)   # from an open one
    assert integ.last_kernel_ms()["launches"] == 5
    _, _, et = integ.advance(20, save_every=10)  # the last step saves a row: evaluated at x_n, closed while there
    assert integ.last_kernel_ms()["launches"] == 21 and et.shape[0] == 2
    integ.store(c, q, p, L)
    assert integ.step == 60
    for a, b in zip(want, (c, q, p, L)):
        assert torch.equal(a, b)


@pytest.mark.parametrize("model", [1, 2])
def test_dense_instantiation_steps_like_the_fixed_row_one(model):
    """The DENSE fp32 instantiation (result rows out of the workgroup's pool, five workgroups per CU) serves grids of
    more than four workgroups per CU; forced on a small system through mythos_debug_set it steps the trajectory of the
    fixed-row instantiation to fp32 rounding (two compilations of the same arithmetic: not bit for bit), and a thin
    skin takes both through halts and out-of-turn rebuilds."""
    from mythos_amd.hip_system import LangevinIntegrator

    top, c0, q0 = generators.ideal_duplex(700, model=model, seed=4)
    s = _system(model, top, torch.float32)
    res = []
    try:
        for mode in (2, 1):
            _lib.debug_set("md_dense", mode)
            integ = LangevinIntegrator(s, dt=0.005, kT=KT, gamma_t=KT / 2.5, gamma_r=KT / 7.5, seed=21)
            integ.set_neighbor_policy(R_CUT, 0.25, 30)  # a thin skin: some rebuilds come out of turn
            c, q = _dev(c0, torch.float32, s), _dev(q0, torch.float32, s)
            p, L = integ.init_momenta()
            integ.load(c, q, p, L)
            integ.advance(10)
            integ.advance(15)
            integ.store(c, q, p, L)
            near = [t.clone() for t in (c, q, p, L)]
            integ.advance(200)
            rec = integ.last_recoveries()
            integ.store(c, q, p, L)
            res.append((near, rec, torch.isfinite(c).all().item() and torch.isfinite(p).all().item()))
    finally:
        _lib.debug_set("md_dense", 0)
    assert res[0][1] >= 1 and res[1][1] >= 1 and res[0][2] and res[1][2]
    for a, b in zip(res[0][0], res[1][0]):
        scale = float(a.abs().max())
        assert float((a - b).abs().max()) <= 1e-4 * scale + 2e-6


def test_load_then_store_round_trips_a_state_without_any_neighbour_setup():
    """mythos_langevin_store launches only when it has an open frame to close: a state that was merely loaded comes
    back bit for bit from an integrator that has neither a neighbour list nor a policy for one."""
    from mythos_amd.hip_system import LangevinIntegrator

    top, c0, q0 = generators.ideal_duplex(40, model=2, seed=3)
    s = _system(2, top, torch.float64)
    integ = LangevinIntegrator(s, dt=0.005, kT=KT, gamma_t=KT / 2.5, gamma_r=KT / 7.5, seed=2)
    c, q = _dev(c0, torch.float64, s), _dev(q0, torch.float64, s)
    q = q / q.norm(dim=1, keepdim=True)
    p, L = integ.init_momenta()
    integ.load(c, q, p, L)
    out = [torch.empty_like(t) for t in (c, q, p, L)]
    integ.store(*out)
    for a, b in zip((c, p, L), (out[0], out[2], out[3])):
        assert torch.equal(a, b)
    assert (out[1] - q).abs().max() < 1e-15
    with pytest.raises(_lib.MythosHipError, match="neighbour"):
        integ.advance(1)
