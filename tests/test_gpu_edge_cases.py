"""Edge cases of the oxDNA path through the C ABI: empty and ragged inputs, circular strands (the bonded pair
(first, last) the reference appends, mythos/input/topology.py:178-180), one-nucleotide strands, an empty
neighbour list, zero frames, zero steps, and a ring whose closing bond is far outside the FENE well (the smoothed
branch of mythos/energy/dna1/interactions.py:16-41)."""

import numpy as np
import pytest
import torch

from mythos_amd import _lib
from mythos_amd.energy import flat_params as fp
from mythos_amd.input import defaults, topology
from mythos_amd.utils import generators
from oracle import oxdna_oracle as orc
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _system(model, top, box, dtype=torch.float64, hce=False):
    from mythos_amd.hip_system import OxdnaSystem

    sim, cfg = defaults.default_configs_for(f"dna{model}")
    flat = fp.pack_flat(fp.derive_flat(model, cfg, kt=sim["kT"], salt_conc=0.5, half_charged_ends=hce), _lib.param_names())
    s = OxdnaSystem(model, top.seq, top.is_end, top.bonded_neighbors, box=box, dtype=dtype)
    s.set_params(flat)
    return s


def _oracle(model, top, c, q, box, pairs, hce=False):
    P = H.oracle_params(model, half_charged_ends=hce)
    tt = (torch.as_tensor(top.seq, dtype=torch.long), torch.as_tensor(top.is_end, dtype=torch.long),
          torch.as_tensor(top.bonded_neighbors, dtype=torch.long).reshape(-1, 2), torch.as_tensor(pairs, dtype=torch.long).reshape(-1, 2))
    e = orc.energy_terms(model, P, torch.as_tensor(c), torch.as_tensor(q), *tt, box=box)
    _, gc, gq = orc.energy_and_grads(model, P, torch.as_tensor(c), torch.as_tensor(q), *tt, box=box)
    return e.numpy(), gc.numpy(), gq.numpy()


@pytest.mark.parametrize("model", [1, 2])
def test_circular_strands_and_the_smoothed_fene_branch(model):
    ref_top, traj, _, _ = H.load_golden(model, "simple-helix")
    top = topology.from_arrays(ref_top.seq, ref_top.strand_counts, is_circular=[True, True])
    assert len(top.bonded_neighbors) == len(ref_top.bonded_neighbors) + 2 and top.is_end.sum() == 0
    c, q = traj.center[3], traj.quaternions[3]
    s = _system(model, top, traj.box_size)
    s.set_neighbors(top.unbonded_neighbors)
    e, gc, gq, _ = s.energy(torch.as_tensor(c, device=s.device), torch.as_tensor(q, device=s.device), grads=True)
    e_ref, gc_ref, gq_ref = _oracle(model, top, c, q, traj.box_size, top.unbonded_neighbors)
    n_terms = 7 if model == 1 else 8
    np.testing.assert_allclose(e.cpu().numpy()[:n_terms], e_ref[:n_terms], rtol=1e-9, atol=1e-9)
    assert e_ref[0] > 10.0  # the two closing bonds span the helix: far beyond the FENE well, finite by smoothing
    np.testing.assert_allclose(gc.cpu().numpy(), gc_ref, rtol=0, atol=1e-8 * np.abs(gc_ref).max())
    np.testing.assert_allclose(gq.cpu().numpy(), gq_ref, rtol=0, atol=1e-8 * np.abs(gq_ref).max())


def test_ragged_strands_single_nucleotides_and_empty_lists():
    ref_top, traj, _, _ = H.load_golden(2, "simple-helix")
    # 16 nucleotides cut into strands of 1, 5, 1, 2, 7: two strands have no bond at all
    top = topology.from_arrays(ref_top.seq, [1, 5, 1, 2, 7])
    assert len(top.bonded_neighbors) == 16 - 5
    c, q = traj.center[0], traj.quaternions[0]
    s = _system(2, top, traj.box_size, hce=True)
    cd, qd = torch.as_tensor(c, device=s.device), torch.as_tensor(q, device=s.device)
    s.set_neighbors(top.unbonded_neighbors)
    e, gc, gq, _ = s.energy(cd, qd, grads=True)
    e_ref, gc_ref, gq_ref = _oracle(2, top, c, q, traj.box_size, top.unbonded_neighbors, hce=True)
    np.testing.assert_allclose(e.cpu().numpy(), e_ref, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(gc.cpu().numpy(), gc_ref, rtol=0, atol=1e-8 * np.abs(gc_ref).max())
    # empty unbonded list: only the bonded terms survive
    s.set_neighbors(np.zeros((0, 2), dtype=np.int32))
    e0 = s.energy(cd, qd)[0].cpu().numpy()
    e0_ref, _, _ = _oracle(2, top, c, q, traj.box_size, np.zeros((0, 2), dtype=np.int64), hce=True)
    np.testing.assert_allclose(e0, e0_ref, rtol=1e-9, atol=1e-9)
    assert np.all(e0[3:] == 0.0)
    # zero frames: nothing to do, shapes kept
    ez, gz, _, _ = s.energy(cd[None][:0], qd[None][:0], grads=True)
    assert ez.shape == (0, 8) and gz.shape == (0, 16, 3)


def test_zero_steps_and_one_nucleotide_system():
    from mythos_amd.hip_system import LangevinIntegrator, OxdnaSystem

    ref_top, traj, _, _ = H.load_golden(2, "simple-helix")
    s = _system(2, ref_top, traj.box_size, dtype=torch.float32)
    s.set_neighbors(ref_top.unbonded_neighbors)
    integ = LangevinIntegrator(s, dt=0.005, kT=0.1, gamma_t=0.04, gamma_r=0.013, seed=1)
    c = torch.as_tensor(traj.center[0], dtype=torch.float32, device=s.device).contiguous()
    q = torch.as_tensor(traj.quaternions[0], dtype=torch.float32, device=s.device).contiguous()
    p, L = integ.init_momenta()
    c0, p0 = c.clone(), p.clone()
    tc, tq, et = integ.run(c, q, p, L, 0)
    assert tc is None and torch.equal(c, c0) and torch.equal(p, p0) and integ.step == 0
    # a free nucleotide: no forces at all, pure Ornstein-Uhlenbeck motion, and still finite
    one = topology.from_arrays(np.array([2]), [1])
    s1 = OxdnaSystem(2, one.seq, one.is_end, one.bonded_neighbors.reshape(-1, 2), box=None, dtype=torch.float64)
    sim, cfg = defaults.default_configs_for("dna2")
    s1.set_params(fp.pack_flat(fp.derive_flat(2, cfg, kt=sim["kT"], salt_conc=0.5, half_charged_ends=True), _lib.param_names()))
    s1.set_neighbors(np.zeros((0, 2), dtype=np.int32))
    e = s1.energy(torch.zeros((1, 3), dtype=torch.float64, device=s1.device),
                  torch.tensor([[1.0, 0.0, 0.0, 0.0]], dtype=torch.float64, device=s1.device))[0]
    assert torch.all(e == 0)
    i1 = LangevinIntegrator(s1, dt=0.005, kT=0.1, gamma_t=0.04, gamma_r=0.013, seed=2)
    c1 = torch.zeros((1, 3), dtype=torch.float64, device=s1.device)
    q1 = torch.tensor([[1.0, 0.0, 0.0, 0.0]], dtype=torch.float64, device=s1.device)
    p1, L1 = i1.init_momenta()
    i1.run(c1, q1, p1, L1, 50)
    assert torch.isfinite(c1).all() and abs(float(q1.norm()) - 1.0) < 1e-12 and float(c1.abs().max()) > 0


def test_md_kernel_with_circular_strands_matches_the_oracle_step_by_step(md_lanes):
    """The MD kernel's bonded wave makes a second sweep for the ring-closing bonds (slots 2 / 3)."""
    from mythos_amd.hip_system import LangevinIntegrator
    from oracle.langevin_oracle import LangevinOracle

    ref_top, traj, _, _ = H.load_golden(2, "simple-helix")
    top = topology.from_arrays(ref_top.seq, ref_top.strand_counts, is_circular=[True, False])
    s = _system(2, top, traj.box_size)
    s.set_neighbors(top.unbonded_neighbors)
    kT = 0.0987
    integ = LangevinIntegrator(s, dt=0.0005, kT=kT, gamma_t=0.04, gamma_r=0.013, seed=77)  # small dt: the ring bond pulls hard
    c = torch.as_tensor(traj.center[0], device=s.device).contiguous()
    q = torch.as_tensor(traj.quaternions[0], device=s.device).contiguous()
    p, L = integ.init_momenta()
    x0, q0, p0, L0 = (t.cpu().numpy().copy() for t in (c, q, p, L))
    tc, tq, et = integ.run(c, q, p, L, 4, save_every=1)
    o = LangevinOracle(2, H.oracle_params(2), H.topo_tensors(top), traj.box_size, 0.0005, kT, 0.04, 0.013, 1.0, (1.0, 1.0, 1.0), seed=77)
    x, qq, pp, LL = x0, q0, p0, L0
    for k in range(4):
        x, qq, pp, LL, u = o.step(x, qq, pp, LL)
        np.testing.assert_allclose(tc[k].cpu().numpy(), x, rtol=0, atol=1e-9)
        np.testing.assert_allclose(tq[k].cpu().numpy(), qq, rtol=0, atol=1e-9)
        assert abs(et[k, :8].sum().item() - u) < 1e-8 * abs(u)


def test_md_rows_grow_for_a_dense_system():
    """A compact bundle of short duplexes has far more neighbours per nucleotide than a lone duplex: the first build
    of a run must enlarge the rows instead of failing (default capacity 64)."""
    from mythos_amd.hip_system import LangevinIntegrator
    from mythos_amd.utils import generators

    top, c0, q0 = generators.duplex_bundle(40, 16, spacing=3.0, seed=3)  # 16 duplexes of 40 bp, axes 3 units apart
    s = _system(2, top, None, dtype=torch.float32, hce=True)
    integ = LangevinIntegrator(s, dt=0.001, kT=0.0987, gamma_t=0.04, gamma_r=0.013, seed=4)
    integ.set_neighbor_policy(3.25, 2.5, 10)  # a generous skin: every backbone site within 4.8 units is listed
    c = torch.as_tensor(c0, dtype=torch.float32, device=s.device).contiguous()
    q = torch.as_tensor(q0, dtype=torch.float32, device=s.device).contiguous()
    p, L = integ.init_momenta()
    integ.run(c, q, p, L, 30)
    mx, mean = s.neighbor_stats()
    assert mx > 64 and torch.isfinite(c).all(), (mx, mean)


def test_md_rows_that_overflow_in_the_middle_of_a_run_grow_and_the_run_continues():
    """Two duplexes start 12 units apart, out of each other's list range, and one drifts onto the other (NVE, a
    centre-of-mass velocity).  Their rows, sized at the first build for a lone duplex, become too short at some
    rebuild in the middle of the run: that rebuild halts the launches behind it, the run grows the rows and carries on
    instead of failing."""
    from mythos_amd.hip_system import LangevinIntegrator
    from mythos_amd.utils import generators

    top, c0, q0 = generators.duplex_bundle(24, 2, spacing=12.0, seed=5)
    s = _system(2, top, None, dtype=torch.float32, hce=True)
    integ = LangevinIntegrator(s, dt=0.005, kT=0.0987, gamma_t=0.0, gamma_r=0.0, seed=4)
    integ.set_neighbor_policy(3.25, 3.5, 20)
    c = torch.as_tensor(c0, dtype=torch.float32, device=s.device).contiguous()
    q = torch.as_tensor(q0, dtype=torch.float32, device=s.device).contiguous()
    p = torch.zeros_like(c)
    L = torch.zeros_like(c)
    half = top.n_nucleotides // 2
    axis = torch.as_tensor(c0[half:].mean(0) - c0[:half].mean(0), dtype=torch.float32, device=s.device)
    p[:half] = 0.8 * axis / axis.norm()  # towards the other duplex: 0.004 units per step
    integ.run(c, q, p, L, 1)
    first_max, _ = s.neighbor_stats()
    integ.run(c, q, p, L, 2000)
    mx, _ = s.neighbor_stats()
    assert torch.isfinite(c).all() and mx > 64 and mx > 1.25 * first_max, (first_max, mx)
    assert integ.last_recoveries() >= 1


def test_md_partially_filled_workgroup_matches_the_oracle(md_lanes):
    """50 nucleotides = one full 32-nucleotide workgroup and one with 18 of 32 groups idle (and a padded grid)."""
    from mythos_amd.hip_system import LangevinIntegrator
    from mythos_amd.utils import generators
    from oracle.langevin_oracle import LangevinOracle

    top, c0, q0 = generators.ideal_duplex(25, model=2, seed=11)
    rng = np.random.default_rng(2)
    c0 = c0 + 0.02 * rng.standard_normal(c0.shape)
    s = _system(2, top, None, hce=True)
    s.set_neighbors(top.unbonded_neighbors)
    kT = 0.0987
    integ = LangevinIntegrator(s, dt=0.003, kT=kT, gamma_t=0.04, gamma_r=0.013, seed=5)
    c = torch.as_tensor(c0, device=s.device).contiguous()
    q = torch.as_tensor(q0, device=s.device).contiguous()
    p, L = integ.init_momenta()
    x, qq, pp, LL = (t.cpu().numpy().copy() for t in (c, q, p, L))
    tc, tq, et = integ.run(c, q, p, L, 3, save_every=1)
    o = LangevinOracle(2, H.oracle_params(2, half_charged_ends=True), H.topo_tensors(top), None, 0.003, kT, 0.04, 0.013, 1.0,
                       (1.0, 1.0, 1.0), seed=5)
    for k in range(3):
        x, qq, pp, LL, u = o.step(x, qq, pp, LL)
        np.testing.assert_allclose(tc[k].cpu().numpy(), x, rtol=0, atol=1e-10)
        assert abs(et[k, :8].sum().item() - u) < 1e-8 * abs(u)


def test_full_cell_buckets_spill_without_losing_neighbours():
    """Cell buckets of three places (mythos_debug_set, read when a list is built): nearly every cell overflows into the
    spill list, and the lists must still be complete - the 1 kbp energies against the k-d tree list of the oracle, and
    the MARTINI Verlet forces against the all-pairs kernel."""
    from mythos_amd import _lib
    from tests.test_gpu_full_size import test_cfg2_1kbp_energy_and_forces_match_oracle
    from tests.test_gpu_martini_md import test_step_by_step_parity_with_oracle_fp64, test_verlet_list_forces_equal_all_pairs_energy_kernel

    _lib.debug_set("cell_bucket_cap", 3)
    try:
        test_cfg2_1kbp_energy_and_forces_match_oracle()
        test_verlet_list_forces_equal_all_pairs_energy_kernel()
        test_step_by_step_parity_with_oracle_fp64()
    finally:
        _lib.debug_set("cell_bucket_cap", 0)
    assert _lib.debug_get("cell_bucket_cap") == 0


def test_row_builder_equals_a_kd_tree_on_random_clouds():
    """scripts/stress_rows.py: 24 random systems (600-6 000 particles, blobs in free space and periodic boxes with
    wrapped and unwrapped coordinates, fp32 and fp64, 8-45 neighbours per particle): the device-built rows hold exactly
    the pairs of scipy's k-d tree and the same longest row - with managed buckets and with buckets of five places."""
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    for args in (("11",), ("12", "5")):
        r = subprocess.run([sys.executable, str(root / "scripts" / "stress_rows.py"), *args], cwd=root,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "mismatches: 0" in r.stdout, r.stdout[-2000:] + r.stderr[-1000:]


def test_reference_all_pairs_list_of_a_1000_nt_system_goes_through_the_energy_kernel():
    """The reference's NoNeighborList (simulators/jax_md/utils.py:49-67) is every i < j pair that is not bonded:
    498 502 pairs and up to 998 entries per row for the 1 000-nt persistence-length system (data/sys-defs/
    persistence-length-500bp).  The row walk is segmented, so such rows are no overflow: same energies, forces and
    dU/dtheta as over the Verlet list (a subset that holds every interacting pair), in both precisions."""
    from mythos_amd import _lib
    from mythos_amd.energy import flat_params as fp
    from mythos_amd.hip_system import OxdnaSystem
    from mythos_amd.input import defaults
    from mythos_amd.simulators.neighbors import verlet_pairs_numpy

    top, c, q = generators.ideal_duplex(500, model=2, seed=2)
    rng = np.random.default_rng(0)
    c = c + 0.03 * rng.standard_normal(c.shape)
    # orientations off the ideal too: with exactly parallel base normals cos(theta4) = 1 sits ON the clamp of
    # acos (mythos/utils/math.py:78-81), where the last bit of a dot product decides which branch's slope is used
    q = q + 0.015 * rng.standard_normal(q.shape)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    sim, cfg = defaults.default_configs_for("dna2")
    flat = fp.pack_flat(fp.derive_flat(2, cfg, kt=sim["kT"], salt_conc=0.5, half_charged_ends=True), _lib.param_names())
    allp = np.asarray(top.unbonded_neighbors)
    assert len(allp) == 1000 * 999 // 2 - 998
    ver = verlet_pairs_numpy(c, top.bonded_neighbors, 3.25)
    for dtype, tol in ((torch.float64, 1e-11), (torch.float32, 2e-5)):
        s = OxdnaSystem(2, top.seq, top.is_end, top.bonded_neighbors, box=None, dtype=dtype)
        s.set_params(flat)
        cd = torch.as_tensor(np.stack([c, c + 0.01]), dtype=dtype, device=s.device)
        qd = torch.as_tensor(np.stack([q, q]), dtype=dtype, device=s.device)
        out = []
        for pairs in (ver, allp):
            s.set_neighbors(pairs)
            out.append(s.energy(cd, qd, grads=True, param_grads=True))
        mx, _ = s.neighbor_stats()
        assert mx == 998  # a strand end has one bonded partner: 999 - 1 entries
        for a, b in zip(out[0], out[1]):
            scale = max(1.0, float(a.abs().max()))
            assert float((a.double() - b.double()).abs().max()) <= tol * scale


def test_a_crowded_nucleotide_makes_the_step_rerun_with_wider_work_lists(md_lanes):
    """More than 16 neighbours of one nucleotide inside the range of an angular term: the launch that finds out aborts
    (nothing it wrote counts: frames and momenta ping-pong), the run switches to the 32-row instantiation, repeats
    the step and carries on.  A blob of unbonded nucleotides, tiny time step, fp64: the trajectory equals the
    oracle's.  More than 32 is an error, not a wrong answer."""
    from mythos_amd.hip_system import LangevinIntegrator, OxdnaSystem
    from mythos_amd import _lib
    from mythos_amd.energy import flat_params as fp
    from mythos_amd.input import defaults, topology
    from oracle.langevin_oracle import LangevinOracle

    def blob(n, radius, seed, min_dist=0.27):
        """n centres in a ball, no two closer than min_dist (the repulsions stay finite-sized), random orientations"""
        rng = np.random.default_rng(seed)
        pts = []
        while len(pts) < n:
            v = rng.uniform(-radius, radius, 3)
            if np.linalg.norm(v) <= radius and all(np.linalg.norm(v - w) >= min_dist for w in pts):
                pts.append(v)
        qq = rng.standard_normal((n, 4))
        return np.array(pts), qq / np.linalg.norm(qq, axis=1, keepdims=True)

    sim, cfg = defaults.default_configs_for("dna2")
    flat = fp.pack_flat(fp.derive_flat(2, cfg, kt=sim["kT"], salt_conc=0.5, half_charged_ends=False), _lib.param_names())
    kT, dt = 296.15 * 0.1 / 300.0, 1e-9  # overlapping sites push with up to 1e13: a step must stay a small perturbation
    n = 26
    top = topology.from_arrays(np.arange(n) % 4, [1] * n)
    c0, q0 = blob(n, 0.62, 4)
    s = OxdnaSystem(2, top.seq, top.is_end, top.bonded_neighbors, box=None, dtype=torch.float64)
    s.set_params(flat)
    s.set_neighbors(top.unbonded_neighbors)
    integ = LangevinIntegrator(s, dt=dt, kT=kT, gamma_t=kT / 2.5, gamma_r=kT / 7.5, seed=12)
    c = torch.as_tensor(c0, device=s.device).contiguous()
    q = torch.as_tensor(q0, device=s.device).contiguous()
    p, L = integ.init_momenta()
    x, qq, pp, LL = (t.cpu().numpy().copy() for t in (c, q, p, L))
    tc, tq, et = integ.run(c, q, p, L, 4, save_every=2)
    assert integ.last_recoveries() == 1 and integ.step == 4
    lo = LangevinOracle(2, H.oracle_params(2), H.topo_tensors(top), None, dt, kT, kT / 2.5, kT / 7.5, 1.0, (1.0, 1.0, 1.0), seed=12)
    for k in range(4):
        x, qq, pp, LL, u = lo.step(x, qq, pp, LL)
        if k % 2 == 1:
            np.testing.assert_allclose(tc[k // 2].cpu().numpy(), x, rtol=1e-11, atol=1e-12)
            assert abs(et[k // 2, :8].sum().item() - u) <= 1e-9 * abs(u)
    np.testing.assert_allclose(p.cpu().numpy(), pp, rtol=1e-8, atol=1e-9 * np.abs(pp).max())
    np.testing.assert_allclose(L.cpu().numpy(), LL, rtol=1e-8, atol=1e-9 * np.abs(LL).max())
    # a later run of the same integrator starts with the narrow lists again and recovers again
    integ.run(c, q, p, L, 1)
    assert integ.last_recoveries() == 1
    # 60 nucleotides in the same volume: more than 32 partners in range
    n2 = 60
    top2 = topology.from_arrays(np.arange(n2) % 4, [1] * n2)
    c2, q2 = blob(n2, 0.62, 5, min_dist=0.2)
    s2 = OxdnaSystem(2, top2.seq, top2.is_end, top2.bonded_neighbors, box=None, dtype=torch.float32)
    s2.set_params(flat)
    s2.set_neighbors(top2.unbonded_neighbors)
    integ2 = LangevinIntegrator(s2, dt=dt, kT=kT, gamma_t=kT / 2.5, gamma_r=kT / 7.5, seed=1)
    cc = torch.as_tensor(c2, dtype=torch.float32, device=s2.device).contiguous()
    qc = torch.as_tensor(q2, dtype=torch.float32, device=s2.device).contiguous()
    pc, Lc = integ2.init_momenta()
    before = cc.clone()
    with pytest.raises(_lib.MythosHipError, match="angular term"):
        integ2.run(cc, qc, pc, Lc, 3)
    # what comes back is the last state that counted (dt = 1e-9: indistinguishable from the start in fp32 digits)
    assert torch.isfinite(cc).all() and (cc - before).abs().max() < 1e-2 and integ2.step <= 3


def test_a_crowded_workgroup_exhausts_the_row_pool_and_the_step_reruns():
    """The result rows of the angular pass come out of one pool per workgroup (320 rows in the stepping instantiation:
    a duplex takes 2 + ~5 per nucleotide).  32 unbonded nucleotides in a ball of radius 0.9: no nucleotide has more than 14
    partners inside an angular range (so no work list is too short), but together they need 350 rows - the launch
    aborts on the pool, the run repeats the step with the big instantiation, and the trajectory equals the oracle's."""
    from mythos_amd.hip_system import LangevinIntegrator, OxdnaSystem
    from mythos_amd import _lib
    from mythos_amd.energy import flat_params as fp
    from mythos_amd.input import defaults, topology
    from oracle import oxdna_oracle as orc
    from oracle.langevin_oracle import LangevinOracle

    rng = np.random.default_rng(21)
    n, radius, pts = 32, 0.9, []
    while len(pts) < n:
        v = rng.uniform(-radius, radius, 3)
        if np.linalg.norm(v) <= radius and all(np.linalg.norm(v - w) >= 0.27 for w in pts):
            pts.append(v)
    c0 = np.array(pts)
    q0 = rng.standard_normal((n, 4))
    q0 /= np.linalg.norm(q0, axis=1, keepdims=True)
    sim, cfg = defaults.default_configs_for("dna2")
    named = fp.derive_flat(2, cfg, kt=sim["kT"], salt_conc=0.5, half_charged_ends=False)
    # the premise, counted the way the radial pass flags entries: base-base distance inside the cross-stacking support, or
    # inside the H-bond support for a Watson-Crick pair (list 0); stacking-site distance inside the coaxial support (list 1)
    a1 = orc.quat_to_axes(torch.as_tensor(q0))[0].numpy()
    g = cfg["geometry"]
    db = np.linalg.norm((c0 + g["com_to_hb"] * a1)[:, None] - (c0 + g["com_to_hb"] * a1)[None], axis=-1) + 9.0 * np.eye(n)
    ds = np.linalg.norm((c0 + g["com_to_stacking"] * a1)[:, None] - (c0 + g["com_to_stacking"] * a1)[None], axis=-1) + 9.0 * np.eye(n)
    v = lambda k: float(named[k])  # noqa: E731
    seq = np.arange(n) % 4
    wc = (seq[:, None] + seq[None]) == 3
    n0 = (((db > v("HYDR_RCLOW")) & (db < v("HYDR_RCHIGH")) & wc) | ((db > v("CRST_RCLOW")) & (db < v("CRST_RCHIGH")))).sum(1)
    n1 = ((ds > v("CXST_RCLOW")) & (ds < v("CXST_RCHIGH"))).sum(1)
    assert (n0 + n1).max() <= 14 and (2 + n0 + n1).sum() >= 340

    flat = fp.pack_flat(named, _lib.param_names())
    kT, dt = 296.15 * 0.1 / 300.0, 1e-9
    top = topology.from_arrays(seq, [1] * n)
    s = OxdnaSystem(2, top.seq, top.is_end, top.bonded_neighbors, box=None, dtype=torch.float64)
    s.set_params(flat)
    s.set_neighbors(top.unbonded_neighbors)
    integ = LangevinIntegrator(s, dt=dt, kT=kT, gamma_t=kT / 2.5, gamma_r=kT / 7.5, seed=3)
    c = torch.as_tensor(c0, device=s.device).contiguous()
    q = torch.as_tensor(q0, device=s.device).contiguous()
    p, L = integ.init_momenta()
    x, qq, pp, LL = (t.cpu().numpy().copy() for t in (c, q, p, L))
    _lib.debug_set("md_lanes", 8)  # the premise is about ONE workgroup of 32 nucleotides (a system this small would take 16 lanes per nucleotide, two workgroups)
    try:
        tc, tq, et = integ.run(c, q, p, L, 3, save_every=1)
    finally:
        _lib.debug_set("md_lanes", 0)
    assert integ.last_recoveries() == 1 and integ.step == 3
    lo = LangevinOracle(2, H.oracle_params(2), H.topo_tensors(top), None, dt, kT, kT / 2.5, kT / 7.5, 1.0, (1.0, 1.0, 1.0), seed=3)
    for k in range(3):
        x, qq, pp, LL, u = lo.step(x, qq, pp, LL)
        np.testing.assert_allclose(tc[k].cpu().numpy(), x, rtol=1e-11, atol=1e-12)
        assert abs(et[k, :8].sum().item() - u) <= 1e-9 * abs(u)
    np.testing.assert_allclose(p.cpu().numpy(), pp, rtol=1e-8, atol=1e-9 * np.abs(pp).max())
