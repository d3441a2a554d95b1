"""GPU parity of the MARTINI kernels: GROMACS goldens (as mythos/energy/martini/m2/tests/test_{lj,bond,
angle}.py), the CPU oracle for forces and the MARTINI-3 harmonic angle, and a tiling property at the
benchmark size (20 480 beads)."""

import dataclasses as dc

import numpy as np
import pytest
import torch

from mythos_amd.energy import martini as M
from oracle import martini_oracle as mo
from tests import martini_helpers as MH

pytestmark = pytest.mark.gpu


@dc.dataclass
class Traj:
    center: torch.Tensor
    box_size: torch.Tensor


def _traj(which, dtype=torch.float64):
    x, box, e = MH.frames(which)
    dev = torch.device("cuda", 0)
    return Traj(torch.as_tensor(x, dtype=dtype, device=dev), torch.as_tensor(box, dtype=dtype, device=dev)), e


def test_lj_energy_against_gromacs():
    s = MH.system()
    traj, energies = _traj("lj")
    lj_fn = M.LJ.from_topology(topology=s["top"], params=M.LJConfiguration(**s["lj_params"]))
    computed = lj_fn.map(traj)
    assert energies.shape[0] == traj.center.shape[0]
    assert np.allclose(computed.cpu().numpy(), energies)


def test_bond_energy_against_gromacs():
    s = MH.system()
    traj, energies = _traj("bond")
    fn = M.Bond.from_topology(topology=s["top"], params=M.BondConfiguration(**s["bond_params"]))
    assert np.allclose(fn.map(traj).cpu().numpy(), energies)


def test_g96_angle_energy_against_gromacs():
    s = MH.system()
    traj, energies = _traj("angle")
    params = {k: (np.deg2rad(v) if k.startswith("angle_theta0_") else v) for k, v in s["angle_params"].items()}
    fn = M.Angle.from_topology(topology=s["top"], params=M.AngleConfiguration(**params))
    assert np.allclose(fn.map(traj).cpu().numpy(), energies)


def _system(dtype, angle_kind=0):
    from mythos_amd.hip_system import MartiniSystem

    s = MH.system()
    return MartiniSystem(s["types"], s["sigma"], s["eps"], s["top"].bonded_neighbors, s["bond_k"], s["bond_r0"],
                         s["top"].angles, s["angle_k"], s["angle_t0"], angle_kind=angle_kind, dtype=dtype), s


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("angle_kind", [0, 1])
def test_terms_and_forces_match_oracle(dtype, angle_kind):
    sysm, s = _system(dtype, angle_kind)
    x, box, _ = MH.frames("lj")
    frames = [0, 4, 9]
    pos = torch.as_tensor(x[frames], dtype=dtype, device=sysm.device)
    bx = torch.as_tensor(box[frames], dtype=dtype, device=sysm.device)
    e, g = sysm.energy(pos, bx, grads=True)
    tol = 1e-7 if dtype == torch.float64 else 1e-3
    for k, f in enumerate(frames):
        e_ref, g_ref = mo.energies_and_forces(
            torch.as_tensor(x[f]), torch.as_tensor(box[f]), s["types"], torch.as_tensor(s["sigma"]), torch.as_tensor(s["eps"]),
            s["top"].bonded_neighbors, torch.as_tensor(s["bond_k"]), torch.as_tensor(s["bond_r0"]), s["top"].angles,
            torch.as_tensor(s["angle_k"]), torch.as_tensor(s["angle_t0"]), angle_kind == 0,
        )
        assert np.allclose(e[k].cpu().numpy(), e_ref.numpy(), rtol=max(tol, 1e-9), atol=tol * abs(e_ref.numpy()).max())
        assert (g[k].cpu().double() - g_ref).abs().max().item() <= tol * g_ref.abs().max().item()


def test_autograd_through_the_energy_function():
    s = MH.system()
    traj, _ = _traj("lj")
    lj_fn = M.LJ.from_topology(topology=s["top"], params=M.LJConfiguration(**s["lj_params"]))
    pos = traj.center[:2].clone().requires_grad_(True)
    u = lj_fn.map(Traj(pos, traj.box_size[:2]))
    (g,) = torch.autograd.grad(u.sum(), pos)
    x, box, _ = MH.frames("lj")
    xr = torch.as_tensor(x[0]).requires_grad_(True)
    ur = mo.lj_energy(xr, torch.as_tensor(box[0]), s["types"], torch.as_tensor(s["sigma"]), torch.as_tensor(s["eps"]), s["top"].bonded_neighbors)
    (gr,) = torch.autograd.grad(ur, xr)
    assert (g[0].cpu() - gr).abs().max() < 1e-7 * gr.abs().max()


def test_tiled_bilayer_20480_beads_energy_is_extensive():
    """cfg-3 size: membrane tiled 4 x 4 in-plane; every term must be exactly 16x the single box."""
    from mythos_amd.hip_system import MartiniSystem

    s = MH.system()
    x, box, _ = MH.frames("lj")
    x0, b0 = x[3].copy(), box[3]
    # GROMACS wraps bead by bead: make every lipid whole before replicating the box (each bead's first
    # bond partner has a lower index), otherwise a bond that crossed a face would span a whole tile
    for i, j in s["top"].bonded_neighbors:
        d = x0[j] - x0[i]
        x0[j] = x0[i] + d - b0 * np.round(d / b0)
    reps =[(i, j) for i in range(4) for j in range(4)]
    xt = np.concatenate([x0 + np.array([i * b0[0], j * b0[1], 0.0]) for i, j in reps])
    bt = b0 * np.array([4.0, 4.0, 1.0])
    top = s["top"].tile(16)
    big = MartiniSystem(np.tile(s["types"], 16), s["sigma"], s["eps"], top.bonded_neighbors, np.tile(s["bond_k"], 16),
                        np.tile(s["bond_r0"], 16), top.angles, np.tile(s["angle_k"], 16), np.tile(s["angle_t0"], 16),
                        dtype=torch.float64)
    small, _ = _system(torch.float64)
    dev = big.device
    e_big, g_big = big.energy(torch.as_tensor(xt, device=dev), torch.as_tensor(bt, device=dev), grads=True)
    e_small, g_small = small.energy(torch.as_tensor(x0, device=dev), torch.as_tensor(b0, device=dev), grads=True)
    np.testing.assert_allclose(e_big.cpu().numpy(), 16 * e_small.cpu().numpy(), rtol=1e-10)
    np.testing.assert_allclose(g_big[:1280].cpu().numpy(), g_small.cpu().numpy(), rtol=0, atol=1e-8)


def test_configuration_errors():
    with pytest.raises(ValueError, match="Unexpected parameter"):
        M.AngleConfiguration(bad_param=100.0)
    with pytest.raises(ValueError, match="pairs of k and r0"):
        M.BondConfiguration(bond_k_A_B_C=1.0)
    pairs = ("A_A", "A_B", "B_B")
    cfg = M.LJConfiguration(
        couplings={"eps_all": [f"lj_epsilon_{p}" for p in pairs]}, eps_all=4.0, **{f"lj_sigma_{p}": 0.47 for p in pairs}
    )
    assert cfg.params["lj_epsilon_A_B"] == 4.0 and "eps_all" in cfg.opt_params
    with pytest.raises(ValueError, match="Missing LJ"):
        M.LJConfiguration(lj_sigma_A_B=0.47, lj_epsilon_A_B=4.0)


def _leaf(v):
    return torch.tensor(float(v), dtype=torch.float64, requires_grad=True)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_lj_parameter_gradients_match_oracle_autograd(dtype):
    """dU/d(lj_sigma_A_B), dU/d(lj_epsilon_A_B) through LJ.map, against torch autograd of the oracle with respect
    to its symmetric (T, T) tables (the reference: jax.grad over with_params, objective.py:224-235)."""
    s = MH.system()
    x, box, _ = MH.frames("lj")
    frames = [1, 6]
    params = {k: _leaf(v) for k, v in s["lj_params"].items()}
    fn = M.LJ.from_topology(topology=s["top"], params=M.LJConfiguration(**params)).with_props(dtype=dtype)
    dev = torch.device("cuda", 0)
    traj = Traj(torch.as_tensor(x[frames], dtype=dtype, device=dev), torch.as_tensor(box[frames], dtype=dtype, device=dev))
    w = torch.tensor([1.0, -0.5], dtype=torch.float64, device=dev)  # a non-trivial cotangent per frame
    (fn.map(traj).double() * w).sum().backward()

    sig = torch.as_tensor(s["sigma"]).clone().requires_grad_(True)
    eps = torch.as_tensor(s["eps"]).clone().requires_grad_(True)
    u = sum(wk * mo.lj_energy(torch.as_tensor(x[f]), torch.as_tensor(box[f]), s["types"], sig, eps, s["top"].bonded_neighbors)
            for wk, f in zip([1.0, -0.5], frames))
    gs, ge = torch.autograd.grad(u, (sig, eps))
    idx = {t: i for i, t in enumerate(s["bead_types"])}
    tol = 1e-8 if dtype == torch.float64 else 2e-3
    checked = 0
    scale_s, scale_e = gs.abs().max().item(), ge.abs().max().item()
    for name, leaf in params.items():
        _, kind, a, b = name.split("_")
        ia, ib = idx[a], idx[b]
        g = gs if kind == "sigma" else ge
        ref = g[ia, ib] + (g[ib, ia] if ia != ib else 0.0)
        got = 0.0 if leaf.grad is None else leaf.grad.item()
        assert abs(got - ref.item()) <= tol * (scale_s if kind == "sigma" else scale_e), (name, got, ref.item())
        checked += ref.item() != 0.0
    assert checked >= 20  # DMPC + water use 4 bead types: 10 unordered pairs x 2 parameters see a gradient


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_lj_parameter_gradients_are_reproducible_bit_for_bit(dtype):
    """dU/dsigma, dU/deps come from per-wavefront LDS tables added in a fixed order and a fixed-order sum over the
    workgroups' partial tables (no global atomics since round 4): five calls, on the fixture and on the 20 480-bead
    tiling (720 workgroups per frame), give the same bits."""
    from mythos_amd.hip_system import MartiniSystem

    s = MH.system()
    x, box, _ = MH.frames("lj")
    for reps in (1, 4):
        top = s["top"].tile(reps * reps) if reps > 1 else s["top"]
        tile = lambda a: np.tile(a, reps * reps)  # noqa: E731
        sysm = MartiniSystem(tile(s["types"]), s["sigma"], s["eps"], top.bonded_neighbors, tile(s["bond_k"]), tile(s["bond_r0"]),
                             top.angles, tile(s["angle_k"]), tile(s["angle_t0"]), dtype=dtype)
        frames = [0, 4, 9] if reps == 1 else [3]
        xs = np.stack([np.concatenate([x[f] + np.array([i * box[f][0], j * box[f][1], 0.0]) for i in range(reps) for j in range(reps)])
                       for f in frames])
        bs = np.stack([box[f] * np.array([reps, reps, 1.0]) for f in frames])
        pos = torch.as_tensor(xs, dtype=dtype, device=sysm.device)
        bx = torch.as_tensor(bs, dtype=dtype, device=sysm.device)
        first = sysm.param_grads(pos, bx, bonds=False, angles=False)
        assert first["sigma"].abs().max() > 0 and torch.isfinite(first["eps"]).all()
        for _ in range(4):
            again = sysm.param_grads(pos, bx, bonds=False, angles=False)
            assert torch.equal(first["sigma"], again["sigma"]) and torch.equal(first["eps"], again["eps"])


@pytest.mark.parametrize("angle_cls", [M.Angle, M.Angle3])
def test_bond_and_angle_parameter_gradients_match_oracle_autograd(angle_cls):
    s = MH.system()
    x, box, _ = MH.frames("angle")
    f = 2
    dev = torch.device("cuda", 0)
    traj = Traj(torch.as_tensor(x[f:f + 1], device=dev), torch.as_tensor(box[f:f + 1], device=dev))
    xr, br = torch.as_tensor(x[f]), torch.as_tensor(box[f])

    bp = {k: _leaf(v) for k, v in s["bond_params"].items()}
    bond = M.Bond.from_topology(topology=s["top"], params=M.BondConfiguration(**bp))
    bond.map(traj).sum().backward()
    k = torch.as_tensor(s["bond_k"]).clone().requires_grad_(True)
    r0 = torch.as_tensor(s["bond_r0"]).clone().requires_grad_(True)
    gk, gr = torch.autograd.grad(mo.bond_energy(xr, br, s["top"].bonded_neighbors, k, r0), (k, r0))
    names = s["top"].bond_names
    for pname, leaf in bp.items():
        kind, nm = ("k", pname[len("bond_k_"):]) if pname.startswith("bond_k_") else ("r0", pname[len("bond_r0_"):])
        sel = torch.tensor([n == nm for n in names])
        ref = (gk if kind == "k" else gr)[sel].sum().item()  # every bond with this name shares the parameter
        assert sel.any() and abs(leaf.grad.item() - ref) <= 1e-9 * max(1.0, abs(ref)), (pname, leaf.grad.item(), ref)

    rad = angle_cls is M.Angle
    ap = {k2: _leaf(np.deg2rad(v) if k2.startswith("angle_theta0_") else v) for k2, v in s["angle_params"].items()}
    ang = angle_cls.from_topology(topology=s["top"], params=M.AngleConfiguration(**ap))
    ang.map(traj).sum().backward()
    ak = torch.as_tensor(s["angle_k"]).clone().requires_grad_(True)
    at = torch.as_tensor(s["angle_t0"]).clone().requires_grad_(True)
    gak, gat = torch.autograd.grad(mo.angle_energy(xr, br, s["top"].angles, ak, at, rad), (ak, at))
    anames = s["top"].angle_names
    for pname, leaf in ap.items():
        kind, nm = ("k", pname[len("angle_k_"):]) if pname.startswith("angle_k_") else ("t0", pname[len("angle_theta0_"):])
        sel = torch.tensor([n == nm for n in anames])
        ref = (gak if kind == "k" else gat)[sel].sum().item()
        assert sel.any() and abs(leaf.grad.item() - ref) <= 1e-9 * max(1.0, abs(ref)), (pname, leaf.grad.item(), ref)


def test_coupled_parameter_receives_the_summed_gradient():
    """couplings (martini/base.py:135-208): one optimisable value feeding several table entries."""
    s = MH.system()
    x, box, _ = MH.frames("lj")
    dev = torch.device("cuda", 0)
    traj = Traj(torch.as_tensor(x[:1], device=dev), torch.as_tensor(box[:1], device=dev))
    eps_names = [k for k in s["lj_params"] if k.startswith("lj_epsilon_")]
    shared = _leaf(3.0)
    params = {k: v for k, v in s["lj_params"].items() if not k.startswith("lj_epsilon_")}
    fn = M.LJ.from_topology(topology=s["top"], params=M.LJConfiguration(couplings={"eps_all": eps_names}, eps_all=shared, **params))
    fn.map(traj).sum().backward()
    # U is linear in a common epsilon: dU/deps_all = U / eps_all
    u = fn.map(traj).sum().item()
    assert abs(shared.grad.item() - u / 3.0) <= 1e-9 * abs(u)


def test_m3_harmonic_angle_energy_against_gromacs():
    """mythos/energy/martini/m3/tests/test_angle_m3.py:61-77 on the GPU: DOPC bilayer + water (5 831 beads, topology
    from the .tpr), harmonic angles against gmx energy on ten frames, rtol 1e-5; both precisions."""
    import json

    from mythos_amd.input import gromacs

    base = MH.MG / "m3" / "angle"
    top = gromacs.MartiniTopology.from_tpr(base / "test.tpr")
    params = json.loads((base / "angle_params_rad.json").read_text())
    x, box, _ = gromacs.read_trr(base / "test.trr")
    e = gromacs.read_xvg(base / "angle.xvg")[1:]
    dev = torch.device("cuda", 0)
    for dtype, rtol in ((torch.float64, 1e-5), (torch.float32, 1e-5)):
        fn = M.Angle3.from_topology(topology=top, params=M.AngleConfiguration(**params)).with_props(dtype=dtype)
        traj = Traj(torch.as_tensor(x, dtype=dtype, device=dev), torch.as_tensor(box, dtype=dtype, device=dev))
        got = fn.map(traj).cpu().numpy()
        assert got.shape == (10,)
        np.testing.assert_allclose(got, e, rtol=rtol)
    assert not np.allclose(M.Angle.from_topology(topology=top, params=M.AngleConfiguration(**params)).map(traj).cpu().numpy(), e, rtol=1e-3)
