"""GPU parity at the BASELINE.json sizes.

cfg-2 (oxDNA2 1 kbp duplex, 2 000 nt): the oracle still finishes in seconds over the Verlet pair list,
so energies and forces are compared with it directly (fp64 1e-5 relative, fp32 1e-3).

cfg-4 (oxDNA2 12 kbp duplex, 24 000 nt): the oracle is too slow for a test, so the HIP path is held to
size-independent properties of the model in free space:
  * Newton's third law: the forces sum to zero and so does the total torque about the origin;
  * rigid-motion invariance: U(R x + t, R q) = U(x, q), forces and torques rotate with R;
  * the hashed-cell-list rows and a SciPy k-d tree list give the same energies;
  * fp32 forces stay within 1e-3 of the fp64 ones;
  * the MD kernel is reproducible bit for bit and one run of 2K steps equals two runs of K.
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

R_CUT = 3.25  # as bench.py


def _system(top, dtype):
    from mythos_amd.hip_system import OxdnaSystem

    sim, cfg = defaults.default_configs_for("dna2")
    flat = fp.pack_flat(fp.derive_flat(2, cfg, kt=sim["kT"], salt_conc=0.5, half_charged_ends=True), _lib.param_names())
    s = OxdnaSystem(2, top.seq, top.is_end, top.bonded_neighbors, box=None, dtype=dtype)
    s.set_params(flat)
    return s, sim


def _perturbed_duplex(bp, seed=7, amp=0.03):
    """Ideal duplex with small random displacements / rotations, so no term sits at a symmetric point."""
    top, c, q = generators.ideal_duplex(bp, model=2, seed=1234)
    rng = np.random.default_rng(seed)
    c = c + amp * rng.standard_normal(c.shape)
    q = q + 0.5 * amp * rng.standard_normal(q.shape)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return top, c, q


def _lab_torque(q, dq):
    """-dU/dphi for a rotation of each body about the lab axes, from dU/dq: delta q = 1/2 (0, dphi) (x) q."""
    q0, qv = q[:, :1], q[:, 1:]
    out = np.zeros((q.shape[0], 3))
    for k in range(3):
        e = np.zeros(3)
        e[k] = 1.0
        d0 = -(qv @ e)
        dv = q0 * e[None, :] + np.cross(e[None, :], qv)
        out[:, k] = -0.5 * (dq[:, 0] * d0 + (dq[:, 1:] * dv).sum(1))
    return out


def _rot(axis, ang):
    axis = np.asarray(axis, float) / np.linalg.norm(axis)
    qr = np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * axis])
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return qr, np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K


def _qmul(a, b):
    a0, av, b0, bv = a[..., :1], a[..., 1:], b[..., :1], b[..., 1:]
    return np.concatenate([a0 * b0 - (av * bv).sum(-1, keepdims=True), a0 * bv + b0 * av + np.cross(av, bv)], axis=-1)


def test_cfg2_1kbp_energy_and_forces_match_oracle():
    from oracle import oxdna_oracle as orc

    top, c, q = _perturbed_duplex(1000)
    pairs = verlet_pairs_numpy(c, top.bonded_neighbors, R_CUT)
    P = H.oracle_params(2, half_charged_ends=True)
    tt = (torch.as_tensor(top.seq, dtype=torch.long), torch.as_tensor(top.is_end, dtype=torch.long),
          torch.as_tensor(top.bonded_neighbors, dtype=torch.long), torch.as_tensor(pairs, dtype=torch.long))
    e_ref = orc.energy_terms(2, P, torch.as_tensor(c), torch.as_tensor(q), *tt, box=None).numpy()
    _, gc_ref, gq_ref = orc.energy_and_grads(2, P, torch.as_tensor(c), torch.as_tensor(q), *tt, box=None)
    gc_ref, gq_ref = gc_ref.numpy(), gq_ref.numpy()
    for dtype, tol in ((torch.float64, 1e-5), (torch.float32, 1e-3)):
        s, _ = _system(top, dtype)
        s.set_neighbors(pairs)
        cd = torch.as_tensor(c, dtype=dtype, device=s.device)
        qd = torch.as_tensor(q, dtype=dtype, device=s.device)
        e, gc, gq, _ = s.energy(cd, qd, grads=True)
        e = e.cpu().numpy().reshape(-1)[:8]
        assert np.abs(e - e_ref).max() <= tol * np.abs(e_ref).max(), (dtype, e, e_ref)
        assert abs(e.sum() - e_ref.sum()) <= tol * abs(e_ref.sum())
        assert np.abs(gc.cpu().double().numpy().reshape(-1, 3) - gc_ref).max() <= tol * np.abs(gc_ref).max()
        assert np.abs(gq.cpu().double().numpy().reshape(-1, 4) - gq_ref).max() <= tol * np.abs(gq_ref).max()
        # the device-built list (hashed cells) holds the same interacting pairs
        s.build_neighbors(cd, R_CUT, 0.0)
        e2 = s.energy(cd, qd)[0].cpu().numpy().reshape(-1)[:8]
        assert np.abs(e2 - e).max() <= (1e-10 if dtype == torch.float64 else 1e-4) * np.abs(e).max()


@pytest.fixture(scope="module")
def duplex_12k():
    top, c, q = _perturbed_duplex(12000)
    s, sim = _system(top, torch.float64)
    cd = torch.as_tensor(c, dtype=torch.float64, device=s.device)
    qd = torch.as_tensor(q, dtype=torch.float64, device=s.device)
    s.build_neighbors(cd, R_CUT, 0.0)
    e, gc, gq, _ = s.energy(cd, qd, grads=True)
    return {"top": top, "c": c, "q": q, "sys": s, "sim": sim, "e": e.cpu().numpy().reshape(-1)[:8],
            "gc": gc.cpu().numpy().reshape(-1, 3), "gq": gq.cpu().numpy().reshape(-1, 4)}


def test_cfg4_12kbp_newtons_third_law(duplex_12k):
    d = duplex_12k
    F = -d["gc"]
    tau = _lab_torque(d["q"], d["gq"])
    scale_f = np.abs(F).sum(0).max()
    assert np.abs(F.sum(0)).max() <= 1e-10 * scale_f
    total = (np.cross(d["c"], F) + tau).sum(0)
    scale_t = (np.abs(np.cross(d["c"], F)) + np.abs(tau)).sum(0).max()
    assert np.abs(total).max() <= 1e-10 * scale_t
    assert np.isfinite(d["e"]).all() and d["e"][:1] > 0  # FENE energy of a perturbed chain is positive


def test_cfg4_12kbp_rigid_motion_invariance(duplex_12k):
    d = duplex_12k
    s = d["sys"]
    qr, Rm = _rot([0.3, -1.0, 0.55], 1.234)
    c2 = d["c"] @ Rm.T + np.array([11.0, -7.5, 3.25])
    q2 = _qmul(np.broadcast_to(qr, d["q"].shape), d["q"])
    cd = torch.as_tensor(c2, dtype=torch.float64, device=s.device)
    qd = torch.as_tensor(q2, dtype=torch.float64, device=s.device)
    s.build_neighbors(cd, R_CUT, 0.0)
    e, gc, gq, _ = s.energy(cd, qd, grads=True)
    e = e.cpu().numpy().reshape(-1)[:8]
    np.testing.assert_allclose(e, d["e"], rtol=1e-9, atol=1e-9 * np.abs(d["e"]).max())
    F2 = -gc.cpu().numpy().reshape(-1, 3)
    np.testing.assert_allclose(F2, (-d["gc"]) @ Rm.T, rtol=0, atol=1e-9 * np.abs(d["gc"]).max())
    t1 = _lab_torque(d["q"], d["gq"])
    t2 = _lab_torque(q2, gq.cpu().numpy().reshape(-1, 4))
    np.testing.assert_allclose(t2, t1 @ Rm.T, rtol=0, atol=1e-9 * np.abs(t1).max())


def test_cfg4_12kbp_cell_list_equals_kdtree_list_and_fp32_tracks_fp64(duplex_12k):
    d = duplex_12k
    pairs = verlet_pairs_numpy(d["c"], d["top"].bonded_neighbors, R_CUT)
    s = d["sys"]
    s.set_neighbors(pairs)
    cd = torch.as_tensor(d["c"], dtype=torch.float64, device=s.device)
    qd = torch.as_tensor(d["q"], dtype=torch.float64, device=s.device)
    e = s.energy(cd, qd)[0].cpu().numpy().reshape(-1)[:8]
    np.testing.assert_allclose(e, d["e"], rtol=1e-11, atol=1e-11 * np.abs(d["e"]).max())
    # fp32 arithmetic against fp64 arithmetic on the SAME (fp32-representable) configuration: a straight
    # 12 kbp duplex is 4 800 length units long, where one fp32 ulp of a coordinate is 5e-4 - rounding the
    # inputs is a property of the data, not of the kernel ("fp32 state at large coordinates")
    s32, _ = _system(d["top"], torch.float32)
    c32, q32 = cd.float(), qd.float()
    s32.build_neighbors(c32, R_CUT, 0.0)
    e32, gc32, gq32, _ = s32.energy(c32, q32, grads=True)
    s.build_neighbors(c32.double(), R_CUT, 0.0)
    e64, gc64, gq64, _ = s.energy(c32.double(), q32.double(), grads=True)
    e32, e64 = e32.cpu().numpy().reshape(-1)[:8], e64.cpu().numpy().reshape(-1)[:8]
    gc64, gq64 = gc64.cpu().numpy().reshape(-1, 3), gq64.cpu().numpy().reshape(-1, 4)
    assert np.abs(e32 - e64).max() <= 1e-3 * np.abs(e64).max()
    assert np.abs(gc32.cpu().double().numpy().reshape(-1, 3) - gc64).max() <= 1e-3 * np.abs(gc64).max()
    assert np.abs(gq32.cpu().double().numpy().reshape(-1, 4) - gq64).max() <= 1e-3 * np.abs(gq64).max()


def test_cfg4_12kbp_md_is_reproducible_and_splittable():
    from mythos_amd.hip_system import LangevinIntegrator

    top, c0, q0 = generators.ideal_duplex(12000, model=2, seed=1234)
    s, sim = _system(top, torch.float32)
    kT = sim["kT"]

    def run(chunks, seed):
        integ = LangevinIntegrator(s, dt=0.005, kT=kT, gamma_t=kT / 2.5, gamma_r=kT / 7.5, mass=1.0,
                                   inertia=(1.0, 1.0, 1.0), seed=seed)
        integ.set_neighbor_policy(R_CUT, 0.5, 25)
        c = torch.as_tensor(c0, dtype=torch.float32, device=s.device).contiguous()
        q = torch.as_tensor(q0, dtype=torch.float32, device=s.device).contiguous()
        p, L = integ.init_momenta()
        et = None
        for n in chunks:
            _, _, et = integ.run(c, q, p, L, n, save_every=n)
        return c.cpu(), q.cpu(), p.cpu(), et.cpu()

    a = run([100], 3)
    b = run([100], 3)
    for x, y in zip(a, b):
        assert torch.equal(x, y)  # no atomics in the force path: bitwise reproducible
    # splitting a run moves the list rebuilds (they happen every 25 steps from the start of a run, and at
    # its start), which changes no pair inside the cut-off but the summation order of a row: the trajectories
    # agree to a few ulp of the coordinates
    cc = run([50, 50], 3)
    ulp = torch.finfo(torch.float32).eps * a[0].abs().max().item()  # of the largest coordinate (~4 800)
    assert (a[0] - cc[0]).abs().max() <= 4 * ulp
    d = run([100], 4)
    assert (a[0] - d[0]).abs().max() > 1e-3  # another seed, another trajectory
    # thermal sanity at the full size: momenta are a Maxwell draw and the ideal helix starts at its energy
    # minimum, so after 100 steps some kinetic energy has moved into the potential modes (the thermostat's
    # 1/gamma is 5 000 steps): kinetic temperature a little below kT, never above
    ke = a[3][-1, 8:].sum().item()
    assert 0.8 < ke / (3.0 * top.n_nucleotides * kT) < 1.02


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("bp", [32, 1000])
def test_the_three_modes_of_the_energy_kernel_agree_on_the_energies(bp, dtype):
    """MODE 0 (energies), 1 (+ forces), 2 (+ dU/dtheta) are three instantiations with three register allocations; their
    term energies must agree (fp64: to the rounding of differently ordered sums; since round 4 the gradient modes of the
    fp64 kernel park their radial sums in LDS) and a repeated call must return the same bits.  Round 4 shipped, for an
    hour, a forces-mode configuration whose energies were wrong and changed from run to run while every other test was
    green: this is the test that catches it."""
    from mythos_amd.hip_system import OxdnaSystem

    top, c0, q0 = generators.ideal_duplex(bp, model=2, seed=1234)
    sim, cfg = defaults.default_configs_for("dna2")
    flat = fp.pack_flat(fp.derive_flat(2, cfg, kt=sim["kT"], salt_conc=0.5, half_charged_ends=True), _lib.param_names())
    s = OxdnaSystem(2, top.seq, top.is_end, top.bonded_neighbors, box=None, dtype=dtype)
    s.set_params(flat)
    c = torch.as_tensor(c0, dtype=dtype, device=s.device)
    q = torch.as_tensor(q0, dtype=dtype, device=s.device)
    for how in ("build", "pairs"):
        if how == "build":
            s.build_neighbors(c, 3.25, 0.1)
        elif bp <= 100:
            s.set_neighbors(top.unbonded_neighbors)
        else:
            continue
        e0 = s.energy(c, q)[0].cpu().numpy()
        e1 = s.energy(c, q, grads=True)[0].cpu().numpy()
        e2 = s.energy(c, q, grads=True, param_grads=True)[0].cpu().numpy()
        scale = np.abs(e0).max()
        tol = 1e-12 if dtype == torch.float64 else 2e-6
        assert np.abs(e1 - e0).max() <= tol * scale and np.abs(e2 - e0).max() <= tol * scale, (how, e0, e1, e2)
        for kw in ({}, {"grads": True}, {"grads": True, "param_grads": True}):
            a, b = s.energy(c, q, **kw), s.energy(c, q, **kw)
            assert all(torch.equal(x, y) for x, y in zip(a, b) if x is not None), (how, kw)


@pytest.mark.parametrize("name, top_file, conf_file", [("persistence-length-500bp", "init.top", "relaxed.dat"), ("wlc-fit", "generated.top", "generated.dat")])
def test_relaxed_systems_the_reference_ships_match_the_oracle_and_step(name, top_file, conf_file):
    """data/sys-defs of the reference (a relaxed 500 bp duplex, the 110 bp WLC system; periodic boxes): not ideal helices
    but thermally distorted ones.  Terms, forces and torques against the oracle over the reference's all-pairs set - fp64
    1e-9, fp32 1e-3 - the device-built list gives the same energies, and the state steps: 300 Langevin steps on the GPU
    list stay a duplex (energy per nucleotide, every base pair)."""
    import warnings

    from mythos_amd.hip_system import LangevinIntegrator, OxdnaSystem
    from mythos_amd.input import topology, trajectory
    from oracle import oxdna_oracle as orc

    base = H.GOLDEN / "sys-defs" / name
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        top = topology.from_oxdna_file(base / top_file)
    tr = trajectory.from_file(base / conf_file, top.strand_counts, is_5p_3p=False)
    c, q, box, n = tr.center[0], tr.quaternions[0], tr.box_size, top.n_nucleotides
    P = H.oracle_params(2, half_charged_ends=True)
    seq, is_end, b, u = H.topo_tensors(top)
    e_ref = orc.energy_terms(2, P, torch.as_tensor(c), torch.as_tensor(q), seq, is_end, b, u, box=box).numpy()
    _, gc_ref, gq_ref = orc.energy_and_grads(2, P, torch.as_tensor(c), torch.as_tensor(q), seq, is_end, b, u, box=box)
    gc_ref, gq_ref = gc_ref.numpy(), gq_ref.numpy()
    sim, cfg = defaults.default_configs_for("dna2")
    flat = fp.pack_flat(fp.derive_flat(2, cfg, kt=sim["kT"], salt_conc=0.5, half_charged_ends=True), _lib.param_names())
    for dtype, tol in ((torch.float64, 1e-9), (torch.float32, 1e-3)):
        s = OxdnaSystem(2, top.seq, top.is_end, top.bonded_neighbors, box=box, dtype=dtype)
        s.set_params(flat)
        s.set_neighbors(top.unbonded_neighbors)
        cd = torch.as_tensor(c, dtype=dtype, device=s.device)
        qd = torch.as_tensor(q, dtype=dtype, device=s.device)
        e, gc, gq, _ = s.energy(cd, qd, grads=True)
        e = e.cpu().numpy().reshape(-1)[:8]
        assert np.abs(e - e_ref).max() <= tol * np.abs(e_ref).max(), (dtype, e, e_ref)
        assert np.abs(gc.cpu().double().numpy().reshape(-1, 3) - gc_ref).max() <= max(tol, 1e-8) * np.abs(gc_ref).max()
        assert np.abs(gq.cpu().double().numpy().reshape(-1, 4) - gq_ref).max() <= max(tol, 1e-8) * np.abs(gq_ref).max()
        s.build_neighbors(cd, R_CUT, 0.0)
        e2 = s.energy(cd, qd)[0].cpu().numpy().reshape(-1)[:8]
        assert np.abs(e2 - e).max() <= (1e-10 if dtype == torch.float64 else 1e-4) * np.abs(e).max()
        kT = sim["kT"]
        integ = LangevinIntegrator(s, dt=sim["dt"], kT=kT, gamma_t=kT / sim["diff_coef"], gamma_r=kT / sim["rot_diff_coef"], seed=3)
        integ.set_neighbor_policy(R_CUT, 0.6, 20)
        p, ang = integ.init_momenta()
        integ.run(cd, qd, p, ang, 300)
        assert torch.isfinite(cd).all() and np.allclose(qd.norm(dim=1).cpu().numpy(), 1.0, atol=1e-5)
        s.build_neighbors(cd, R_CUT, 0.0)
        e3 = s.energy(cd, qd)[0].cpu().numpy().reshape(-1)[:8]
        assert -1.62 < e3.sum() / n < -1.30 and e3[4] / n < -0.25, (dtype, e3 / n)
