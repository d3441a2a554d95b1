"""GPU tests of the MARTINI Langevin step kernel (mythos_martini_langevin_run).

Nothing in the reference pins a MARTINI integrator (it delegates dynamics to GROMACS), so: step-by-step parity in
fp64 against oracle/martini_langevin_oracle.py on the same Philox stream, the device-built Verlet list against the
all-pairs energy kernel, physics (NVE drift, kinetic temperature), plumbing (determinism, split runs, errors), and
the BASELINE configs[2] size (bilayer tiled 4 x 4 = 20 480 beads).
"""

import numpy as np
import pytest
import torch

from tests import martini_helpers as MH

pytestmark = pytest.mark.gpu

KB = 0.0083144626
T = 273.0


def _make(dtype, reps=1):
    from mythos_amd.hip_system import MartiniSystem

    s = MH.system()
    x, box, _ = MH.frames("lj")
    x0, b0 = x[3].copy(), box[3].copy()
    top, types = s["top"], s["types"]
    bk, br, ak, at = s["bond_k"], s["bond_r0"], s["angle_k"], s["angle_t0"]
    if reps > 1:
        for i, j in s["top"].bonded_neighbors:  # make every lipid whole before tiling (GROMACS wraps per bead)
            d = x0[j] - x0[i]
            x0[j] = x0[i] + d - b0 * np.round(d / b0)
        cells = [(i, j) for i in range(reps) for j in range(reps)]
        x0 = np.concatenate([x0 + np.array([i * b0[0], j * b0[1], 0.0]) for i, j in cells])
        b0 = b0 * np.array([reps, reps, 1.0])
        top = s["top"].tile(reps * reps)
        types = np.tile(types, reps * reps)
        bk, br, ak, at = (np.tile(a, reps * reps) for a in (bk, br, ak, at))
    sysm = MartiniSystem(types, s["sigma"], s["eps"], top.bonded_neighbors, bk, br, top.angles, ak, at, dtype=dtype)
    return sysm, s, top, types, (bk, br, ak, at), x0, b0


def test_step_by_step_parity_with_oracle_fp64():
    from mythos_amd.hip_system import MartiniLangevinIntegrator
    from oracle.martini_langevin_oracle import MartiniLangevinOracle

    sysm, s, top, types, (bk, br, ak, at), x0, b0 = _make(torch.float64)
    rng = np.random.default_rng(5)
    mass = rng.uniform(40.0, 90.0, size=sysm.n)
    integ = MartiniLangevinIntegrator(sysm, dt=0.01, kT=KB * T, gamma=2.0, mass=mass, seed=0xABCDEF012345)
    integ.set_neighbor_policy(0.25, 2)
    pos = torch.as_tensor(x0, device=sysm.device).contiguous()
    vel = integ.init_velocities()
    v0 = vel.cpu().numpy().copy()
    n_steps = 5
    traj, et = integ.run(pos, vel, b0, n_steps, save_every=1)
    orc = MartiniLangevinOracle(types, s["sigma"], s["eps"], top.bonded_neighbors, bk, br, top.angles, ak, at, True, b0,
                                0.01, KB * T, 2.0, mass, seed=0xABCDEF012345)
    xo, vo = x0.copy(), v0.copy()
    e_ref = orc.run(xo, vo, n_steps)
    # traj[k] / et[k] are the state after k + 1 steps (energies of x_{k+1}, kinetic energy after the closing kick)
    np.testing.assert_allclose(pos.cpu().numpy(), xo, rtol=0, atol=1e-10)
    np.testing.assert_allclose(vel.cpu().numpy(), vo, rtol=0, atol=1e-10)
    np.testing.assert_allclose(et.cpu().numpy(), e_ref, rtol=1e-9, atol=1e-7)
    assert traj.shape == (n_steps, sysm.n, 3)


def test_verlet_list_forces_equal_all_pairs_energy_kernel():
    """One step with dt -> 0 moves nothing: the saved LJ / bond / angle energies must equal mythos_martini_energy."""
    from mythos_amd.hip_system import MartiniLangevinIntegrator

    for dtype, tol in ((torch.float64, 1e-10), (torch.float32, 2e-5)):
        sysm, *_rest, x0, b0 = _make(dtype)
        integ = MartiniLangevinIntegrator(sysm, dt=1e-9, kT=KB * T, gamma=0.0, seed=1)
        pos = torch.as_tensor(x0, dtype=dtype, device=sysm.device).contiguous()
        vel = torch.zeros_like(pos)
        _, et = integ.run(pos, vel, b0, 1, save_every=1)
        e, _ = sysm.energy(torch.as_tensor(x0, dtype=dtype, device=sysm.device), torch.as_tensor(b0, dtype=dtype), grads=False)
        np.testing.assert_allclose(et[0, :3].cpu().numpy(), e.cpu().numpy(), rtol=tol, atol=tol * abs(e.cpu().numpy()).max())


def test_nve_energy_conservation_fp64():
    from mythos_amd.hip_system import MartiniLangevinIntegrator

    sysm, *_rest, x0, b0 = _make(torch.float64)
    integ = MartiniLangevinIntegrator(sysm, dt=0.005, kT=KB * T, gamma=0.0, seed=3)
    integ.set_neighbor_policy(0.3, 5)
    pos = torch.as_tensor(x0, device=sysm.device).contiguous()
    vel = integ.init_velocities()
    _, et = integ.run(pos, vel, b0, 400, save_every=20)
    tot = et.sum(1).cpu().numpy()
    ke = et[:, 3].cpu().numpy()
    # velocity-Verlet with a shifted (not force-switched) LJ: drift far below the kinetic energy scale
    assert np.ptp(tot) < 2e-3 * ke.mean(), (tot, ke.mean())


def test_thermostat_holds_the_temperature_fp32():
    from mythos_amd.hip_system import MartiniLangevinIntegrator

    sysm, *_rest, x0, b0 = _make(torch.float32)
    integ = MartiniLangevinIntegrator(sysm, dt=0.02, kT=KB * T, gamma=1.0, seed=11)
    integ.set_neighbor_policy(0.3, 5)
    pos = torch.as_tensor(x0, dtype=torch.float32, device=sysm.device).contiguous()
    vel = integ.init_velocities()
    integ.run(pos, vel, b0, 1000)
    _, et = integ.run(pos, vel, b0, 2000, save_every=50)
    t_kin = 2.0 * et[:, 3].cpu().numpy() / (3.0 * sysm.n * KB)
    assert abs(t_kin.mean() / T - 1.0) < 0.02, t_kin.mean()
    mx, mean = integ.neighbor_stats()
    assert 40 < mean < 120 and mx <= 160


def test_determinism_split_runs_and_errors():
    from mythos_amd.hip_system import MartiniLangevinIntegrator

    sysm, *_rest, x0, b0 = _make(torch.float32)

    def run(chunks, seed):
        integ = MartiniLangevinIntegrator(sysm, dt=0.02, kT=KB * T, gamma=1.0, seed=seed)
        integ.set_neighbor_policy(0.3, 5)
        pos = torch.as_tensor(x0, dtype=torch.float32, device=sysm.device).contiguous()
        vel = integ.init_velocities()
        for n in chunks:
            integ.run(pos, vel, b0, n)
        return pos.cpu(), vel.cpu()

    a, b = run([60], 2), run([60], 2)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    # a split moves no rebuild (every 5 steps from step 0), but the kick that closes step 30 is then computed
    # with the list built at step 25 instead of step 30: another summation order, fp32 round-off, amplified by
    # 30 more steps of chaotic dynamics
    c = run([30, 30], 2)
    assert (a[0] - c[0]).abs().max() < 1e-4
    d = run([60], 3)
    assert (a[0] - d[0]).abs().max() > 1e-3

    integ = MartiniLangevinIntegrator(sysm, dt=0.02, kT=KB * T, gamma=1.0, seed=2)
    pos = torch.as_tensor(x0, dtype=torch.float32, device=sysm.device).contiguous()
    vel = integ.init_velocities()
    with pytest.raises(ValueError, match="box is smaller"):
        integ.run(pos, vel, [2.0, 2.0, 2.0], 1)
    integ.set_neighbor_policy(0.01, 1000)  # skin far too small for 200 steps without a rebuild
    with pytest.raises(Exception, match="skin"):
        integ.run(pos, vel, b0, 200)
    with pytest.raises(ValueError):
        integ.run(pos.double(), vel, b0, 1)


def test_a_bead_leaving_its_skin_halts_rebuilds_and_resumes_exactly():
    """No scheduled rebuilds and a skin of 0.06 nm: the step that moves a bead 0.03 nm from where the list was built
    halts the launches behind it, the run rebuilds there and carries on.  In fp64 the trajectory equals the one on a
    comfortable policy (to summation order), saved frames and energies included."""
    from mythos_amd.hip_system import MartiniLangevinIntegrator

    sysm, *_rest, x0, b0 = _make(torch.float64)
    outs = []
    for skin, every in ((0.4, 5), (0.06, 1000)):
        integ = MartiniLangevinIntegrator(sysm, dt=0.02, kT=KB * T, gamma=1.0, seed=21)
        integ.set_neighbor_policy(skin, every)
        pos = torch.as_tensor(x0, dtype=torch.float64, device=sysm.device).contiguous()
        vel = integ.init_velocities()
        traj, et = integ.run(pos, vel, b0, 60, save_every=12)
        outs.append((pos.clone(), vel.clone(), traj.clone(), et.clone()))
        assert (integ.last_recoveries() >= 3) == (every == 1000), integ.last_recoveries()
    for x, y in zip(*outs):
        torch.testing.assert_close(x, y, rtol=1e-9, atol=1e-9)


def test_cfg3_bilayer_20480_beads():
    """BASELINE configs[2]: the shipped bilayer tiled 4 x 4; the tiled system must evolve like 16 copies at step 0
    (same energies per tile) and stay at temperature."""
    from mythos_amd.hip_system import MartiniLangevinIntegrator

    big, *_r, xb, bb = _make(torch.float32, reps=4)
    assert big.n == 20480
    integ = MartiniLangevinIntegrator(big, dt=0.02, kT=KB * T, gamma=1.0, seed=5)
    integ.set_neighbor_policy(0.3, 5)
    pos = torch.as_tensor(xb, dtype=torch.float32, device=big.device).contiguous()
    vel = integ.init_velocities()
    _, et0 = integ.run(pos.clone(), vel.clone(), bb, 1, save_every=1)
    e_all, _ = big.energy(torch.as_tensor(xb, dtype=torch.float32, device=big.device), torch.as_tensor(bb, dtype=torch.float32))
    small, *_r2, xs, bs = _make(torch.float32)
    e_one, _ = small.energy(torch.as_tensor(_whole(xs, bs), dtype=torch.float32, device=small.device), torch.as_tensor(bs, dtype=torch.float32))
    np.testing.assert_allclose(e_all.cpu().numpy(), 16 * e_one.cpu().numpy(), rtol=2e-5)
    _, et = integ.run(pos, vel, bb, 500, save_every=100)
    t_kin = 2.0 * et[:, 3].cpu().numpy() / (3.0 * big.n * KB)
    assert abs(t_kin[-1] / T - 1.0) < 0.05
    assert torch.isfinite(pos).all()
    assert abs(et0[0, 0].item() - e_all[0].item()) < 0.05 * abs(e_all[0].item())  # one step moved the beads a little


def _whole(x, b):
    s = MH.system()
    x = x.copy()
    for i, j in s["top"].bonded_neighbors:
        d = x[j] - x[i]
        x[j] = x[i] + d - b * np.round(d / b)
    return x


def test_rows_longer_than_the_default_stride_grow_at_the_first_build():
    """A list range of 1.1 + 1.0 nm holds ~300 beads, more than the 256 slots rows start with: the first build of the
    run must enlarge them (and the trajectory still equals the one on a comfortable list, fp64)."""
    from mythos_amd.hip_system import MartiniLangevinIntegrator

    sysm, *_rest, x0, b0 = _make(torch.float64)
    outs = []
    for skin, every in ((0.3, 5), (1.0, 5)):
        integ = MartiniLangevinIntegrator(sysm, dt=0.02, kT=KB * T, gamma=1.0, seed=8)
        integ.set_neighbor_policy(skin, every)
        pos = torch.as_tensor(x0, dtype=torch.float64, device=sysm.device).contiguous()
        vel = integ.init_velocities()
        integ.run(pos, vel, b0, 20)
        outs.append(pos.clone())
        mx, _ = integ.neighbor_stats()
        assert (mx > 256) == (skin == 1.0), mx
    torch.testing.assert_close(outs[0], outs[1], rtol=1e-9, atol=1e-9)


def test_cells_of_more_than_64_beads_build_the_same_rows():
    """A list range of 1.1 + 0.75 nm is three cells per edge of the 1 280-bead box with ~70 beads in a cell of the bilayer:
    buckets beyond the 64 entries a register sort holds (ordered by rank through LDS since round 4).  The rows are the
    pairs inside the range minus the exclusions (k-d tree on the host), in ascending order inside every cell's stretch,
    and the trajectory equals the one on a comfortable list."""
    from scipy.spatial import cKDTree

    from mythos_amd.hip_system import MartiniLangevinIntegrator

    sysm, s, top, *_rest, x0, b0 = _make(torch.float64)
    rl = 1.1 + 0.75
    nc = np.floor(b0 / rl).astype(int)
    assert (nc >= 3).all()
    xw = np.mod(x0, b0)
    cell = np.floor(xw / b0 * nc).astype(int).clip(0, nc - 1)
    assert np.bincount(cell[:, 0] * 100 + cell[:, 1] * 10 + cell[:, 2]).max() > 64
    outs = []
    for skin in (0.3, 0.75):
        integ = MartiniLangevinIntegrator(sysm, dt=0.02, kT=KB * T, gamma=1.0, seed=8)
        integ.set_neighbor_policy(skin, 5)
        pos = torch.as_tensor(x0, dtype=torch.float64, device=sysm.device).contiguous()
        vel = integ.init_velocities()
        integ.load(pos, vel, b0)
        integ.advance(1)
        if skin == 0.75:
            rows, lens = integ.rows(False)
            tree = cKDTree(xw, boxsize=b0)
            want = tree.query_ball_point(xw, rl - 1e-9)
            bonded = {(int(a), int(b)) for a, b in top.bonded_neighbors} | {(int(b), int(a)) for a, b in top.bonded_neighbors}
            for i in range(0, sysm.n, 11):
                got = rows[i, : lens[i]]
                ref = {j for j in want[i] if j != i and (i, j) not in bonded}
                near = set(tree.query_ball_point(xw[i], rl + 1e-9)) - {i}
                assert ref <= set(got.tolist()) <= near and len(set(got.tolist())) == len(got), i
        integ.advance(19)
        integ.store(pos, vel)
        outs.append(pos.clone())
    torch.testing.assert_close(outs[0], outs[1], rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_resident_advances_equal_one_run_bitwise(dtype):
    """mythos_martini_langevin_load / advance / store: the list and its rebuild schedule carry over,
    advance(n) is n launches and leaves the frame open, and whatever the sequence of calls the state handed back is the
    one mythos_martini_langevin_run(total) hands back, bit for bit."""
    from mythos_amd.hip_system import MartiniLangevinIntegrator

    sysm, *_rest, x0, b0 = _make(dtype)

    def fresh():
        integ = MartiniLangevinIntegrator(sysm, dt=0.02, kT=KB * T, gamma=1.0, seed=17)
        integ.set_neighbor_policy(0.3, 7)
        integ.set_timing(1)
        pos = torch.as_tensor(x0, dtype=dtype, device=sysm.device).contiguous()
        return integ, pos, integ.init_velocities()

    integ, pos, vel = fresh()
    integ.run(pos, vel, b0, 60)
    assert integ.last_kernel_ms()["launches"] == 61 and integ.step == 60
    want = (pos.clone(), vel.clone())

    integ, pos, vel = fresh()
    with pytest.raises(Exception, match="resident"):
        integ.advance(1)
    integ.load(pos, vel, b0)
    integ.advance(17)
    assert integ.last_kernel_ms()["launches"] == 17
    integ.advance(0)
    out = (torch.empty_like(pos), torch.empty_like(vel))
    integ.store(*out)  # closes the open frame with one launch
    assert integ.last_kernel_ms()["launches"] == 1 and integ.step == 17
    again = (torch.empty_like(pos), torch.empty_like(vel))
    integ.store(*again)  # already closed: a copy
    assert torch.equal(out[0], again[0]) and torch.equal(out[1], again[1])
    integ.advance(20)  # from a closed frame
    _, et = integ.advance(23, save_every=23)  # from an open one; the last step saves an energy row: closed while there
    assert integ.last_kernel_ms()["launches"] == 24 and et.shape == (1, 4)
    integ.store(pos, vel)
    assert integ.step == 60
    assert torch.equal(pos, want[0]) and torch.equal(vel, want[1])


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_positions_only_rows_equal_single_step_stores_and_the_energy_trace_route(dtype):
    """e_trace == NULL: the launch that produces a saved state writes its row (no energy-trace instantiation, no closing
    launch); the rows are the states store hands out after single-step advances and the rows of the energy-trace route,
    bit for bit - through out-of-turn rebuilds as well (a thin skin, segments of five launches)."""
    from mythos_amd import _lib
    from mythos_amd.hip_system import MartiniLangevinIntegrator

    sysm, *_rest, x0, b0 = _make(dtype)

    def fresh(skin=0.3, every=6):
        integ = MartiniLangevinIntegrator(sysm, dt=0.02, kT=KB * T, gamma=1.0, seed=23)
        integ.set_neighbor_policy(skin, every)
        integ.set_timing(1)
        pos = torch.as_tensor(x0, dtype=dtype, device=sysm.device).contiguous()
        vel = integ.init_velocities()
        integ.load(pos, vel, b0)
        return integ, pos, vel

    n = 20
    integ, pos, vel = fresh()
    traj, et = integ.advance(n, save_every=1, want_energy=False)
    assert et is None and traj.shape == (n, sysm.n, 3) and integ.last_kernel_ms()["launches"] == n
    integ, pos, vel = fresh()
    for k in range(n):
        integ.advance(1)
        integ.store(pos, vel)
        assert torch.equal(traj[k], pos), k
    integ, pos, vel = fresh()
    traj2, et2 = integ.advance(n, save_every=1)
    assert torch.equal(traj, traj2) and torch.isfinite(et2).all()
    integ, pos, vel = fresh()
    traj3, _ = integ.advance(n, save_every=7, want_energy=False)  # a cadence that does not divide the call
    assert traj3.shape[0] == 2 and torch.equal(traj3[0], traj[6]) and torch.equal(traj3[1], traj[13])
    # through halts: no scheduled rebuild, a skin of 0.08 nm
    integ, pos, vel = fresh(0.08, 1000)
    _lib.debug_set("md_segment", 5)
    try:
        traj4, _ = integ.advance(40, save_every=1, want_energy=False)
    finally:
        _lib.debug_set("md_segment", 0)
    assert integ.last_recoveries() >= 2
    integ, pos, vel = fresh(0.08, 1000)
    for k in range(40):
        integ.advance(1)
        integ.store(pos, vel)
        assert torch.equal(traj4[k], pos), k


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_pruned_rows_are_the_verlet_rows_inside_the_margin_and_change_no_physics(dtype):
    """mythos_martini_langevin_set_inner_list: every fourth launch writes, from the distances it computes anyway, the
    entries of each Verlet row inside r_c + margin to a second set of rows, the launches in between walk those.
    (i) the pruned rows ARE that subsequence, entry for entry, for the state they were written from; (ii) the trajectory
    equals the one without pruning up to the order of the sums; (iii) split calls, single-step stores and the
    energy-trace route stay bitwise equal with pruning on (which list a launch walks depends only on the steps since the
    Verlet rows were built); (iv) a margin the beads outrun halts and recovers like a thin skin - still the same physics."""
    from mythos_amd.hip_system import MartiniLangevinIntegrator

    sysm, *_rest, x0, b0 = _make(dtype)
    r_c, skin, margin = 1.1, 0.4, 0.2

    def fresh(inner=(margin, 4), every=8):
        integ = MartiniLangevinIntegrator(sysm, dt=0.02, kT=KB * T, gamma=1.0, seed=31)
        integ.set_neighbor_policy(skin, every)
        integ.set_inner_list(*inner)
        pos = torch.as_tensor(x0, dtype=dtype, device=sysm.device).contiguous()
        vel = integ.init_velocities()
        integ.load(pos, vel, b0)
        return integ, pos, vel

    # (i) launch 0 prunes the rows it was given, at the loaded positions
    integ, pos, vel = fresh()
    integ.advance(1)
    rows, lens = integ.rows(False)
    rows_in, lens_in = integ.rows(True)
    xw = torch.as_tensor(x0, dtype=dtype).numpy().astype(np.float64)  # the positions as the kernel saw them
    b = np.asarray(b0, dtype=np.float64)
    n_checked = 0
    for i in range(0, sysm.n, 7):
        j = rows[i, : lens[i]]
        d = xw[i] - xw[j]
        d -= b * np.round(d / b)
        r2 = (d * d).sum(1)
        edge = np.abs(np.sqrt(r2) - (r_c + margin)) < (1e-5 if dtype == torch.float32 else 1e-12)  # (rounding at the very edge)
        keep = r2 < (r_c + margin) ** 2
        got = rows_in[i, : lens_in[i]]
        want = j[keep | edge] if edge.any() and len(got) != keep.sum() else j[keep]
        if edge.any():
            assert set(j[keep & ~edge]) <= set(got) <= set(j[keep | edge])
        else:
            assert np.array_equal(got, want), i
        n_checked += 1
    assert n_checked > 100 and lens_in.mean() < 0.85 * lens.mean()
    # (ii) same physics
    n = 24
    integ, pos, vel = fresh()
    tr_on, _ = integ.advance(n, save_every=1, want_energy=False)
    integ, pos, vel = fresh(inner=(0.0, 0))
    tr_off, _ = integ.advance(n, save_every=1, want_energy=False)
    tol = 2e-4 if dtype == torch.float32 else 1e-10
    assert (tr_on - tr_off).abs().max().item() < tol and not torch.equal(tr_on, tr_off)
    # (iii) bitwise: split calls, single-step stores, the energy-trace route
    integ, pos, vel = fresh()
    for k in range(n):
        integ.advance(1)
        integ.store(pos, vel)
        assert torch.equal(tr_on[k], pos), k
    integ, pos, vel = fresh()
    integ.advance(5)
    integ.advance(7)
    integ.advance(n - 12)
    integ.store(pos, vel)
    assert torch.equal(tr_on[-1], pos)
    integ, pos, vel = fresh()
    tr_e, et = integ.advance(n, save_every=1)
    assert torch.equal(tr_e, tr_on) and torch.isfinite(et).all()
    integ, pos, vel = fresh()
    p2 = torch.as_tensor(x0, dtype=dtype, device=sysm.device).contiguous()
    v2 = integ.init_velocities()
    integ.run(p2, v2, b0, n)
    assert torch.equal(p2, tr_on[-1])
    # (iv) a margin that does not hold: 0.02 nm / 4 allows 0.0033 nm per step, a thermal bead makes 0.006
    integ, pos, vel = fresh(inner=(0.02, 4))
    tr_h, _ = integ.advance(n, save_every=1, want_energy=False)
    assert integ.last_recoveries() >= 5
    assert (tr_h - tr_off).abs().max().item() < tol
    integ, pos, vel = fresh(inner=(0.02, 4))
    with pytest.raises(Exception, match="margin of the pruned rows"):
        integ.advance(400)
