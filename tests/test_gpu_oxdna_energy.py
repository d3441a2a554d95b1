"""GPU parity: HIP energies / forces / quaternion gradients / dU/dtheta through the C ABI
vs the CPU oracle and the oxDNA golden files.

Tolerances (BASELINE.json north_star): 1e-5 relative in fp64, 1e-3 in fp32; the golden-file
comparisons use the reference's own tolerances (dna2/tests/test_integration.py:94..374).
"""

import numpy as np
import pytest
import torch

from mythos_amd import _lib
from mythos_amd.energy import flat_params as fp
from mythos_amd.input import defaults
from tests import helpers as H

pytestmark = pytest.mark.gpu

CASES = [
    (1, "simple-helix", False),
    (1, "simple-coax", False),
    (2, "simple-helix", False),
    (2, "simple-coax", False),
    (2, "simple-helix-half-charged-ends", True),
    (3, "simple-helix-12bp", False),  # oxRNA2 (rna2/tests/test_integration.py: salt 1.0, whole end charges)
    (3, "simple-coax", False),
]
SALT = {1: 0.5, 2: 0.5, 3: 1.0}


def _system(model, top, traj, hce, dtype, overrides=None):
    from mythos_amd.hip_system import OxdnaSystem

    sim, cfg = defaults.default_configs_for(H.model_dir(model))
    for sec, d in (overrides or {}).items():
        cfg[sec].update(d)
    flat = fp.pack_flat(fp.derive_flat(model, cfg, kt=sim["kT"], salt_conc=SALT[model], half_charged_ends=hce), _lib.param_names())
    s = OxdnaSystem(model, top.seq, top.is_end, top.bonded_neighbors, box=traj.box_size, dtype=dtype)
    s.set_params(flat)
    s.set_neighbors(top.unbonded_neighbors)
    return s


def _frames(traj, dtype, dev, idx=None):
    c = traj.center if idx is None else traj.center[idx]
    q = traj.quaternions if idx is None else traj.quaternions[idx]
    return torch.as_tensor(c, dtype=dtype, device=dev), torch.as_tensor(q, dtype=dtype, device=dev)


@pytest.mark.parametrize(("model", "name", "hce"), CASES)
def test_fp64_terms_match_oracle_and_golden(model, name, hce):
    top, traj, split, energy = H.load_golden(model, name)
    s = _system(model, top, traj, hce, torch.float64)
    c, q = _frames(traj, torch.float64, s.device)
    e, _, _, _ = s.energy(c, q)
    e = e.cpu().numpy() / top.n_nucleotides
    P = H.oracle_params(model, half_charged_ends=hce, salt=SALT[model])
    ref = H.oracle_terms_traj(model, P, top, traj, use_axes=False)
    nt = ref.shape[1]
    np.testing.assert_allclose(e[:, :nt], ref, rtol=1e-9, atol=1e-11)
    if model == 1:
        assert np.all(e[:, 7] == 0.0)
    for k in range(nt):
        term = H.SPLIT_COLUMNS[1 + k]
        if model == 1 and name == "simple-coax" and term == "stacking":
            continue
        # (oxRNA2: the same slack on the rounding of the sixth decimal as the oracle's own golden test)
        np.testing.assert_allclose(np.around(e[:, k], 6), split[:, 1 + k], atol=H.TERM_ATOL[term] + (5e-7 if model == 3 else 0.0), err_msg=term)
    if model == 1 and name == "simple-helix":  # the reference's own assertion (dna1/tests/test_integration.py:322-389)
        np.testing.assert_allclose(np.around(e.sum(1), 6), energy, rtol=1e-5, atol=1e-6)
    else:
        np.testing.assert_allclose(e.sum(1), energy, atol=1e-3 if model >= 2 else 1e-4)


@pytest.mark.parametrize(("model", "name", "hce"), CASES)
def test_fp32_terms_within_1e3(model, name, hce):
    top, traj, split, _ = H.load_golden(model, name)
    s = _system(model, top, traj, hce, torch.float32)
    c, q = _frames(traj, torch.float32, s.device)
    e, _, _, _ = s.energy(c, q)
    e = e.cpu().numpy()
    P = H.oracle_params(model, half_charged_ends=hce, salt=SALT[model])
    ref = H.oracle_terms_traj(model, P, top, traj, use_axes=False) * top.n_nucleotides
    tot = np.abs(ref).sum(1, keepdims=True)
    assert np.max(np.abs(e[:, : ref.shape[1]] - ref) / tot) < 1e-3
    np.testing.assert_allclose(e.sum(1), ref.sum(1), rtol=1e-3)


def _oracle_grads(model, P, top, traj, f):
    from oracle import oxdna_oracle as orc

    seq, is_end, b, u = H.topo_tensors(top)
    return orc.energy_and_grads(
        model, P, torch.as_tensor(traj.center[f]), torch.as_tensor(traj.quaternions[f]), seq, is_end, b, u, box=traj.box_size
    )


@pytest.mark.parametrize(("model", "name", "hce"), CASES)
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_forces_and_quaternion_gradients(model, name, hce, dtype):
    top, traj, _, _ = H.load_golden(model, name)
    s = _system(model, top, traj, hce, dtype)
    frames = [0, 17, 42, 99]
    c, q = _frames(traj, dtype, s.device, frames)
    e, gc, gq, _ = s.energy(c, q, grads=True)
    P = H.oracle_params(model, half_charged_ends=hce, salt=SALT[model])
    tol = 1e-5 if dtype == torch.float64 else 1e-3
    for k, f in enumerate(frames):
        u, rc, rq = _oracle_grads(model, P, top, traj, f)
        scale_c = rc.abs().max().item()
        scale_q = rq.abs().max().item()
        assert abs(e[k].sum().item() - u.item()) <= tol * abs(u.item())
        assert (gc[k].cpu().double() - rc).abs().max().item() <= tol * scale_c
        assert (gq[k].cpu().double() - rq).abs().max().item() <= tol * scale_q


@pytest.mark.parametrize("model", [1, 2, 3])
@pytest.mark.parametrize("bonded", [False, True])
def test_random_dimers(model, bonded):
    """Every pair term on random relative poses (tests/helpers.py random_dimers; the coaxial term is 0 in every stored
    frame of the oxDNA1 and oxRNA2 goldens): energies, forces, quaternion gradients and dU/dtheta against the oracle."""
    from mythos_amd.hip_system import OxdnaSystem
    from oracle import oxdna_oracle as orc

    top, c0, q0, live = H.random_dimers(model, bonded)
    assert all(v >= 5 for v in live.values()), live
    pairs = np.zeros((0, 2), np.int64) if bonded else np.arange(top.n_nucleotides).reshape(-1, 2)
    sim, cfg, leaves = _leaf_cfg(model)
    flat = fp.pack_flat(fp.derive_flat(model, cfg, kt=sim["kT"], salt_conc=SALT[model], half_charged_ends=False), _lib.param_names())
    sim2, cfg2, leaves2 = _leaf_cfg(model)
    P = orc.init_all(model, cfg2, kt=sim2["kT"], salt_conc=SALT[model], half_charged_ends=False)
    seq, is_end, b, _ = H.topo_tensors(top)
    u, rc, rq = orc.energy_and_grads(model, P, torch.as_tensor(c0), torch.as_tensor(q0), seq, is_end, b, torch.as_tensor(pairs))
    e_ref = orc.energy_terms(model, P, torch.as_tensor(c0), torch.as_tensor(q0), seq, is_end, b, torch.as_tensor(pairs)).detach().numpy()
    for dtype, tol in ((torch.float64, 1e-9), (torch.float32, 1e-3)):
        s = OxdnaSystem(model, top.seq, top.is_end, top.bonded_neighbors, box=None, dtype=dtype)
        s.set_params(flat.detach())
        s.set_neighbors(pairs)
        c = torch.as_tensor(c0[None], dtype=dtype, device=s.device)
        q = torch.as_tensor(q0[None], dtype=dtype, device=s.device)
        e, gc, gq, gflat = s.energy(c, q, grads=True, param_grads=dtype == torch.float64)
        e = e[0].cpu().numpy()
        np.testing.assert_allclose(e[: len(e_ref)], e_ref, rtol=tol, atol=tol * np.abs(e_ref).max())
        for got, ref in ((gc, rc), (gq, rq)):
            ref = ref.detach().numpy()
            got = got[0].double().cpu().numpy()
            rms = np.sqrt((ref**2).mean())
            assert np.abs(got - ref).max() <= tol * (np.abs(ref).max() if dtype == torch.float32 else rms), (dtype, np.abs(got - ref).max(), rms)
        if dtype == torch.float64:
            keys = list(leaves)
            g = torch.autograd.grad(flat, [leaves[k] for k in keys], grad_outputs=gflat[0].cpu(), allow_unused=True)
            u_leaf = orc.energy(model, P, torch.as_tensor(c0), torch.as_tensor(q0), seq, is_end, b, torch.as_tensor(pairs))
            w = torch.autograd.grad(u_leaf, [leaves2[k] for k in keys], allow_unused=True)
            got = np.array([0.0 if x is None else float(x) for x in g])
            ref = np.array([0.0 if x is None else float(x) for x in w])
            bad = np.abs(got - ref) > 1e-6 * np.maximum(np.abs(ref), 1e-3 * np.abs(ref).max())
            assert not bad.any(), [(keys[i], got[i], ref[i]) for i in np.nonzero(bad)[0]]
            assert np.count_nonzero(ref) >= (20 if bonded else 40)
            if model != 3:  # the oxRNA2-only entries of a dU/dparams row stay zero in the oxDNA instantiations
                names = _lib.param_names()
                assert float(gflat[0, names.index("GEO_STACK3_A1"):].abs().max()) == 0.0


def test_energy_is_the_same_from_all_three_kernel_modes():
    """energy-only, +gradient and +parameter-partial launches report the same energies (fp64
    round-off only: the instantiations differ in FMA contraction) and each is run-to-run bitwise."""
    top, traj, _, _ = H.load_golden(2, "simple-helix")
    s = _system(2, top, traj, False, torch.float64)
    c, q = _frames(traj, torch.float64, s.device, [3, 4])
    e0, _, _, _ = s.energy(c, q)
    e1, _, _, _ = s.energy(c, q, grads=True)
    e2, _, _, _ = s.energy(c, q, grads=True, param_grads=True)
    torch.testing.assert_close(e0, e1, rtol=1e-13, atol=1e-14)
    torch.testing.assert_close(e0, e2, rtol=1e-13, atol=1e-14)
    assert torch.equal(e0, s.energy(c, q)[0])
    assert torch.equal(e1, s.energy(c, q, grads=True)[0])


def _leaf_cfg(model):
    sim, cfg = defaults.default_configs_for(H.model_dir(model))
    leaves = {}
    for sec, d in cfg.items():
        if sec == "geometry":
            continue
        for k, v in d.items():
            t = torch.tensor(float(v), dtype=torch.float64, requires_grad=True)
            d[k] = t
            leaves[(sec, k)] = t
    return sim, cfg, leaves


@pytest.mark.parametrize(("model", "name", "hce"), [(1, "simple-helix", False), (2, "simple-helix", True), (2, "simple-coax", False), (1, "simple-coax", False),
                                                     (3, "simple-helix-12bp", False), (3, "simple-coax", False)])
def test_parameter_gradients_chain_rule(model, name, hce):
    """dU/dtheta for every independent parameter: HIP partials + host chain rule vs oracle autograd
    (the stand-in for jax.value_and_grad, mythos/optimization/objective.py:235)."""
    from mythos_amd.hip_system import OxdnaSystem
    from oracle import oxdna_oracle as orc

    top, traj, _, _ = H.load_golden(model, name)
    frames = [5, 60]
    # ---- HIP side
    sim, cfg, leaves = _leaf_cfg(model)
    kt = torch.tensor(sim["kT"], dtype=torch.float64, requires_grad=True)
    flat = fp.pack_flat(fp.derive_flat(model, cfg, kt=kt, salt_conc=SALT[model], half_charged_ends=hce), _lib.param_names())
    s = OxdnaSystem(model, top.seq, top.is_end, top.bonded_neighbors, box=traj.box_size, dtype=torch.float64)
    s.set_params(flat)
    s.set_neighbors(top.unbonded_neighbors)
    c, q = _frames(traj, torch.float64, s.device, frames)
    e, _, _, gflat = s.energy(c, q, param_grads=True)
    keys = list(leaves)
    hip = []
    for k in range(len(frames)):
        g = torch.autograd.grad(flat, [leaves[kk] for kk in keys] + [kt], grad_outputs=gflat[k].cpu(), retain_graph=True, allow_unused=True)
        hip.append([0.0 if x is None else float(x) for x in g])
    # ---- oracle side (independent restatement of init_params)
    sim2, cfg2, leaves2 = _leaf_cfg(model)
    kt2 = torch.tensor(sim2["kT"], dtype=torch.float64, requires_grad=True)
    P = orc.init_all(model, cfg2, kt=kt2, salt_conc=SALT[model], half_charged_ends=hce)
    seq, is_end, b, u = H.topo_tensors(top)
    for k, f in enumerate(frames):
        U = orc.energy(model, P, torch.as_tensor(traj.center[f]), torch.as_tensor(traj.quaternions[f]), seq, is_end, b, u, box=traj.box_size)
        g = torch.autograd.grad(U, [leaves2[kk] for kk in keys] + [kt2], retain_graph=True, allow_unused=True)
        ref = np.array([0.0 if x is None else float(x) for x in g])
        got = np.array(hip[k])
        assert abs(e[k].sum().item() - U.item()) < 1e-9 * abs(U.item())
        scale = np.abs(ref).max()
        bad = np.abs(got - ref) > 1e-5 * np.maximum(np.abs(ref), 1e-3 * scale)
        assert not bad.any(), [(keys[i] if i < len(keys) else "kt", got[i], ref[i]) for i in np.nonzero(bad)[0]]
        assert np.count_nonzero(ref) >= 15  # the comparison is not vacuous


def test_sequence_dependent_weights_fp64():
    top, traj, split, _ = H.load_golden(1, "simple-helix-seq-dep")
    ss = H.read_ss_weights(H.GOLDEN / "dna1" / "simple-helix-seq-dep" / "seq_dep.dat")
    ov = {
        "stacking": {"ss_stack_weights": torch.as_tensor(ss["ss_stack_weights"]), "eps_stack_kt_coeff": ss["eps_stack_kt_coeff"]},
        "hydrogen_bonding": {"ss_hb_weights": torch.as_tensor(ss["ss_hb_weights"])},
    }
    s = _system(1, top, traj, False, torch.float64, overrides=ov)
    c, q = _frames(traj, torch.float64, s.device)
    e = s.energy(c, q)[0].cpu().numpy() / top.n_nucleotides
    np.testing.assert_allclose(np.around(e[:, 2], 6), split[:, 3], atol=1e-6)
    np.testing.assert_allclose(np.around(e[:, 4], 6), split[:, 5], atol=1e-3)


def test_gpu_neighbor_build_reproduces_all_pairs_energy():
    """A Verlet list with the interaction cut-off must give the all-pairs energy (compact support)."""
    top, traj, _, _ = H.load_golden(2, "simple-helix")
    s = _system(2, top, traj, False, torch.float64)
    c, q = _frames(traj, torch.float64, s.device, [10])
    e_all = s.energy(c, q)[0]
    s.build_neighbors(c[0], r_cut=3.3, skin=0.2)
    mx, mean = s.neighbor_stats()
    assert 0 < mean <= mx <= 14
    e_list = s.energy(c, q)[0]
    np.testing.assert_allclose(e_list.cpu().numpy(), e_all.cpu().numpy(), rtol=1e-12, atol=1e-13)


def test_error_conventions():
    from mythos_amd.hip_system import OxdnaSystem

    top, traj, _, _ = H.load_golden(2, "simple-helix")
    s = OxdnaSystem(2, top.seq, top.is_end, top.bonded_neighbors, box=traj.box_size, dtype=torch.float64)
    c, q = _frames(traj, torch.float64, s.device, [0])
    with pytest.raises(_lib.MythosHipError):  # parameters / neighbours not set
        s.energy(c, q)
    with pytest.raises(ValueError):
        s.set_params(np.zeros(3))
    with pytest.raises(ValueError):
        s.energy(c.float(), q.float())


@pytest.mark.parametrize("periodic", [False, True])
def test_hashed_cell_list_reproduces_all_pairs_energy(periodic):
    """N >= 512 takes the hashed cell-list build; its rows must give the all-pairs energy, in free
    space (one long duplex) and in a periodic box (a bundle of short duplexes straddling the faces)."""
    from mythos_amd.hip_system import OxdnaSystem
    from mythos_amd.utils import generators

    if periodic:
        top, c0, q0 = generators.duplex_bundle(12, 36, spacing=6.5, seed=5)
        box = np.array([39.0, 39.0, 39.0])
        c0 = c0 + np.array([-3.0, 17.0, 36.5])  # unwrapped coordinates crossing the faces
    else:
        top, c0, q0 = generators.ideal_duplex(400, seed=5)
        box = None
    sim, cfg = defaults.default_configs_for("dna2")
    flat = fp.pack_flat(fp.derive_flat(2, cfg, kt=sim["kT"], half_charged_ends=True), _lib.param_names())
    s = OxdnaSystem(2, top.seq, top.is_end, top.bonded_neighbors, box=box, dtype=torch.float64)
    s.set_params(flat)
    c = torch.as_tensor(c0, dtype=torch.float64, device=s.device)
    q = torch.as_tensor(q0, dtype=torch.float64, device=s.device)
    s.set_neighbors(top.unbonded_neighbors)
    e_all, gc_all, _, _ = s.energy(c, q, grads=True)
    s.build_neighbors(c, r_cut=3.25, skin=0.4)
    mx, mean = s.neighbor_stats()
    assert 10 < mean < 80
    e_cell, gc_cell, _, _ = s.energy(c, q, grads=True)
    np.testing.assert_allclose(e_cell.cpu().numpy(), e_all.cpu().numpy(), rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(gc_cell.cpu().numpy(), gc_all.cpu().numpy(), rtol=0, atol=1e-10)
    # rebuilding gives the identical list (bitwise identical energies)
    s.build_neighbors(c, r_cut=3.25, skin=0.4)
    assert torch.equal(s.energy(c, q, grads=True)[0], e_cell)


def test_parameter_gradients_are_bitwise_reproducible():
    """Energies and forces are gathers; the dU/dtheta partials go through LDS atomics into accumulator copies that
    each belong to one wavefront, so their order of summation is fixed too: repeated calls agree bit for bit."""
    top, traj, _, _ = H.load_golden(2, "simple-helix")
    for dtype in (torch.float64, torch.float32):
        s = _system(2, top, traj, False, dtype)
        c, q = _frames(traj, dtype, s.device)
        first = s.energy(c, q, grads=True, param_grads=True)
        for _ in range(4):
            again = s.energy(c, q, grads=True, param_grads=True)
            for a, b in zip(first, again):
                assert torch.equal(a, b)


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


SMALL_SYSDEFS = [
    ("hairpins/4bp_stem_8nt_loop", "init.conf"), ("hairpins/6bp_stem_6nt_loop", "init_bound.conf"),
    ("hairpins/6bp_stem_6nt_loop", "init_unbound.conf"), ("simple-helix", "bound.conf"), ("simple-helix", "bound_relaxed.conf"),
    ("simple-helix", "unbound.conf"),
]


@pytest.mark.parametrize("model", [1, 2])
@pytest.mark.parametrize("case, conf", SMALL_SYSDEFS)
def test_small_system_definitions_of_the_reference_match_the_oracle(case, conf, model):
    """data/sys-defs of the reference's melting-temperature examples: hairpins (ONE strand; the stem's base pairs are
    unbonded pairs inside it, the loop is unpaired; one conformation clashes under oxDNA2's backbone site: excluded
    volume of 70 units), the 8 bp duplex bound, relaxed in a box of 5 units - smaller than twice the interaction range -
    and with its strands apart.  Terms, forces, torques: fp64 1e-9, fp32 1e-3 against the oracle; the device-built list
    gives the energies of the all-pairs set."""
    import warnings

    from mythos_amd import _lib
    from mythos_amd.energy import flat_params as fp
    from mythos_amd.hip_system import OxdnaSystem
    from mythos_amd.input import defaults, topology, trajectory
    from oracle import oxdna_oracle as orc
    from tests import helpers as H

    base = H.GOLDEN / "sys-defs" / case
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        top = topology.from_oxdna_file(base / "sys.top")
    tr = trajectory.from_file(base / conf, top.strand_counts, is_5p_3p=False)
    c, q, box = tr.center[0], tr.quaternions[0], tr.box_size
    hce = model == 2
    P = H.oracle_params(model, half_charged_ends=hce)
    seq, is_end, b, u = H.topo_tensors(top)
    e_ref = orc.energy_terms(model, P, torch.as_tensor(c), torch.as_tensor(q), seq, is_end, b, u, box=box).numpy()
    _, gc_ref, gq_ref = orc.energy_and_grads(model, P, torch.as_tensor(c), torch.as_tensor(q), seq, is_end, b, u, box=box)
    sim, cfg = defaults.default_configs_for("dna1" if model == 1 else "dna2")
    flat = fp.pack_flat(fp.derive_flat(model, cfg, kt=sim["kT"], salt_conc=0.5, half_charged_ends=hce), _lib.param_names())
    scale = max(np.abs(e_ref).max(), 1.0)
    a3 = np.stack([2 * (q[:, 1] * q[:, 3] + q[:, 0] * q[:, 2]), 2 * (q[:, 2] * q[:, 3] - q[:, 0] * q[:, 1]),
                   q[:, 0] ** 2 - q[:, 1] ** 2 - q[:, 2] ** 2 + q[:, 3] ** 2], axis=1)
    bn = np.asarray(top.bonded_neighbors)
    on_kink = bool((np.abs((a3[bn[:, 0]] * a3[bn[:, 1]]).sum(1)) > 1 - 1e-10).any())  # (see the torque comparison below)
    for dtype, tol in ((torch.float64, 1e-9), (torch.float32, 1e-3)):
        s = OxdnaSystem(model, top.seq, top.is_end, top.bonded_neighbors, box=box, dtype=dtype)
        s.set_params(flat)
        s.set_neighbors(top.unbonded_neighbors)
        cd = torch.as_tensor(c, dtype=dtype, device=s.device)
        qd = torch.as_tensor(q, dtype=dtype, device=s.device)
        e, gc, gq, _ = s.energy(cd, qd, grads=True)
        e = e.cpu().numpy().reshape(-1)[: len(e_ref)]
        assert np.abs(e - e_ref).max() <= tol * scale, (dtype, e, e_ref)
        gs = max(gc_ref.abs().max().item(), gq_ref.abs().max().item(), 1.0)
        assert (gc.cpu().double().reshape(-1, 3) - gc_ref).abs().max().item() <= max(tol, 1e-8) * gs
        # Torques, i.e. the part of dU/dq that moves a body.  The straight strands of the hairpin files sit EXACTLY on a kink
        # of the reference's energy: neighbouring a3 are parallel, cos(theta4) = 1 is where `clamp` cuts the gradient to
        # zero (utils/math.py:68-88) while the limit from inside is 2 a.  Which side a dot product of 1 - 1e-16 (fp64) or
        # 1 - 6e-8 (fp32) falls on is rounding.  The difference is along q itself (a change of |q|, which the axes are
        # quadratic in and the integrator projects out): the torques agree, the raw dU/dq only where nothing is on the kink.
        gq_d = gq.cpu().double().reshape(-1, 4).numpy()
        tq, tq_ref = _lab_torque(q, gq_d), _lab_torque(q, gq_ref.numpy())
        assert np.abs(tq - tq_ref).max() <= max(tol, 1e-8) * gs, (dtype, np.abs(tq - tq_ref).max())
        if dtype == torch.float64 and not on_kink:
            assert np.abs(gq_d - gq_ref.numpy()).max() <= 1e-8 * gs
        s.build_neighbors(cd, 3.25, 0.0)
        e2 = s.energy(cd, qd)[0].cpu().numpy().reshape(-1)[: len(e_ref)]
        assert np.abs(e2 - e).max() <= (1e-10 if dtype == torch.float64 else 1e-4) * scale, (dtype, e2, e)


@pytest.mark.parametrize("name", ["circle", "burns-natnano-2015"])
def test_circular_strands_on_the_gpu_against_the_oracle_and_against_oxdna(name):
    """The reference's shipped oxDNA runs with circular strands (tests/golden/regr, see tests/test_oracle_golden.py): the
    kernels against the oracle with the reference's closing pair (first, last) - the second-bond slots of the rows - fp64
    1e-9 / fp32 1e-3 on terms, forces, torques; and, with the closing pairs in strand direction, against oxDNA's own
    per-term energies to its printed decimals.  Device-built list and a short run on the ring included."""
    from mythos_amd.hip_system import LangevinIntegrator, OxdnaSystem
    from oracle import oxdna_oracle as orc

    top, traj, split, turned = H.load_regr(name)
    n, box = top.n_nucleotides, traj.box_size
    P = H.oracle_params(2, half_charged_ends=True)
    seq, is_end, b, u = H.topo_tensors(top)
    sim, cfg = defaults.default_configs_for("dna2")
    flat = fp.pack_flat(fp.derive_flat(2, cfg, kt=sim["kT"], salt_conc=0.5, half_charged_ends=True), _lib.param_names())
    frames = [0, 3, traj.center.shape[0] - 1]
    for dtype, tol in ((torch.float64, 1e-9), (torch.float32, 1e-3)):
        for bonded, against_oxdna in ((top.bonded_neighbors, False), (turned, True)):
            s = OxdnaSystem(2, top.seq, top.is_end, bonded, box=box, dtype=dtype)
            s.set_params(flat)
            s.set_neighbors(top.unbonded_neighbors)
            cd = torch.as_tensor(traj.center[frames], dtype=dtype, device=s.device)
            qd = torch.as_tensor(traj.quaternions[frames], dtype=dtype, device=s.device)
            e, gc, gq, _ = s.energy(cd, qd, grads=True)
            e = e.cpu().numpy().reshape(len(frames), -1)[:, :8]
            if against_oxdna:
                assert np.abs(e / n - split[frames]).max() <= (2.5e-6 if dtype == torch.float64 else 2e-5), (dtype, np.abs(e / n - split[frames]).max(0))
                continue
            for k, f in enumerate(frames):
                c, q = torch.as_tensor(traj.center[f]), torch.as_tensor(traj.quaternions[f])
                e_ref = orc.energy_terms(2, P, c, q, seq, is_end, b, u, box=box).numpy()
                _, gc_ref, gq_ref = orc.energy_and_grads(2, P, c, q, seq, is_end, b, u, box=box)
                assert np.abs(e[k] - e_ref).max() <= tol * np.abs(e_ref).max(), (dtype, f)
                gs = max(gc_ref.abs().max().item(), gq_ref.abs().max().item())
                assert (gc[k].cpu().double() - gc_ref).abs().max().item() <= max(tol, 1e-8) * gs
                assert (gq[k].cpu().double() - gq_ref).abs().max().item() <= max(tol, 1e-8) * gs
            s.build_neighbors(cd[0], 3.25, 0.0)
            e2 = s.energy(cd[0], qd[0])[0].cpu().numpy().reshape(-1)[:8]
            assert np.abs(e2 - e[0]).max() <= (1e-10 if dtype == torch.float64 else 1e-4) * np.abs(e[0]).max()
            kT = sim["kT"]
            integ = LangevinIntegrator(s, dt=0.003, kT=kT, gamma_t=kT / sim["diff_coef"], gamma_r=kT / sim["rot_diff_coef"], seed=5)
            integ.set_neighbor_policy(3.25, 0.6, 20)
            c0, q0 = cd[-1].clone().contiguous(), qd[-1].clone().contiguous()
            p, ang = integ.init_momenta()
            integ.run(c0, q0, p, ang, 200)
            s.build_neighbors(c0, 3.25, 0.0)
            e3 = s.energy(c0, q0)[0].cpu().numpy().reshape(-1)[:8]
            assert torch.isfinite(c0).all() and abs(e3.sum() / n - split[-1].sum()) < 0.15, (dtype, e3.sum() / n, split[-1].sum())


@pytest.mark.parametrize("name, ss_file", [("simple-helix-oxdna2-ss", "oxDNA2_sequence_dependent_parameters.txt"), ("simple-coax-oxdna2-rev", None),
                                           ("simple-helix-oxdna2-12bp", None), ("simple-helix-rna2-12bp-half-charged-ends", None)])
def test_more_oxdna2_runs_of_the_reference_on_the_gpu(name, ss_file):
    """The three further oxDNA2 runs of tests/test_oracle_golden.py through the kernels: every frame's terms per nucleotide
    against oxDNA's split energies - oxDNA2's own sequence-dependent weights among them - fp64 at the oracle's margins, fp32 1e-4."""
    top, traj, split, _ = H.load_regr(name)
    ov = None
    if ss_file is not None:
        w = H.read_ss_weights(H.GOLDEN / "regr" / name / ss_file)
        ov = {"stacking": {"ss_stack_weights": torch.as_tensor(w["ss_stack_weights"]), "eps_stack_kt_coeff": w["eps_stack_kt_coeff"]},
              "hydrogen_bonding": {"ss_hb_weights": torch.as_tensor(w["ss_hb_weights"])}}
    rna = "rna2" in name  # (oxRNA2 with half-charged strand ends, salt 1.0)
    for dtype, tight, loose in ((torch.float64, 2.5e-6, 5e-5), (torch.float32, 1e-4, 1e-4)):
        s = _system(3 if rna else 2, top, traj, rna, dtype, overrides=ov)
        c, q = _frames(traj, dtype, s.device)
        e = s.energy(c, q)[0].cpu().numpy().reshape(len(split), -1)[:, :8] / top.n_nucleotides
        err = np.abs(e - split).max(0)
        assert err[[0, 1, 2, 3, 6]].max() <= tight and err[[4, 5, 7]].max() <= loose, (dtype, err)


@pytest.mark.parametrize("name, seqdep, tol_stk", [("lammps-oxdna2-40bp-sa", False, 3e-6), ("lammps-oxdna2-40bp", True, 2e-5)])
def test_lammps_runs_at_another_temperature_and_salt_on_the_gpu(name, seqdep, tol_stk):
    """The LAMMPS runs of tests/test_oracle_golden.py through the kernels: T = 0.1, salt 0.15 - the stacking strength and
    the Debye length of conditions no other golden has - per term and dumped step against LAMMPS's own log (fp64 at the
    oracle's margins, fp32 1e-4).  LAMMPS has no bonded excluded volume; its excluded-volume column is our non-bonded one."""
    from mythos_amd.hip_system import OxdnaSystem

    top, traj, lam = H.load_lammps_regr(name)
    n = top.n_nucleotides
    sim, cfg = defaults.default_configs_for("dna2")
    if seqdep:
        w = H.read_ss_weights(H.GOLDEN / "regr" / "simple-helix-oxdna2-ss" / "oxDNA2_sequence_dependent_parameters.txt")
        cfg["stacking"].update({"ss_stack_weights": torch.as_tensor(w["ss_stack_weights"]), "eps_stack_kt_coeff": w["eps_stack_kt_coeff"]})
        cfg["hydrogen_bonding"].update({"ss_hb_weights": torch.as_tensor(w["ss_hb_weights"])})
    flat = fp.pack_flat(fp.derive_flat(2, cfg, kt=0.1, salt_conc=0.15, half_charged_ends=False), _lib.param_names())
    for dtype, tol, ts in ((torch.float64, 3e-6, tol_stk), (torch.float32, 1e-4, 1e-4)):
        s = OxdnaSystem(2, top.seq, top.is_end, top.bonded_neighbors, box=traj.box_size, dtype=dtype)
        s.set_params(flat)
        s.set_neighbors(top.unbonded_neighbors)
        c, q = _frames(traj, dtype, s.device)
        e = s.energy(c, q)[0].cpu().numpy().reshape(traj.center.shape[0], -1)[:, :8] / n
        for key, col, t in (("bond", 0, tol), ("hb", 4, tol), ("excv", 3, tol), ("stk", 2, ts), ("xstk", 5, tol), ("coax", 6, tol), ("dh", 7, tol)):
            assert np.abs(e[:, col] - lam[key]).max() <= t, (dtype, key, np.abs(e[:, col] - lam[key]).max())


@pytest.mark.parametrize("name, model, hce", [("simple-helix-60bp", 1, False), ("simple-helix-60bp-oxdna2", 2, True)])
def test_sixty_base_pairs_total_energy_on_the_gpu(name, model, hce):
    """The 120-nt duplexes of tests/test_oracle_golden.py through the kernels: oxDNA's total potential energy per nucleotide
    of every kept configuration (fp64 6e-6 - the printed digits of the configurations - fp32 1e-4), all-pairs and device-built list."""
    from tests.test_oracle_golden import _sixty_bp

    top, traj, want = _sixty_bp(name)
    for dtype, tol in ((torch.float64, 6e-6), (torch.float32, 1e-4)):
        s = _system(model, top, traj, hce, dtype)
        c, q = _frames(traj, dtype, s.device)
        e = s.energy(c, q)[0].cpu().numpy().reshape(traj.center.shape[0], -1)[:, :8].sum(1) / top.n_nucleotides
        assert np.abs(e - want).max() <= tol, (dtype, e, want)
        s.build_neighbors(c[0], 3.25, 0.0)
        e2 = s.energy(c[:1], q[:1])[0].cpu().numpy().reshape(-1)[:8].sum() / top.n_nucleotides
        assert abs(e2 - e[0]) <= (1e-10 if dtype == torch.float64 else 1e-5)
