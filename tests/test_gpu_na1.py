"""oxNA (hybrid DNA / RNA, model 4 of the C ABI) on the GPU against the oracle and the reference's seven na1 goldens
(mythos/energy/na1/tests/test_integration.py): energies per term, forces, quaternion gradients, dU/dtheta for the three
parameter vectors; random dimers of all four pair kinds; Langevin steps through the fused kernel's oxNA instantiation and through the two-launch path; what the boundary refuses."""

import numpy as np
import pytest
import torch

from mythos_amd import _lib
from mythos_amd.energy import flat_params as fp
from mythos_amd.input import defaults
from oracle import oxdna_oracle as orc
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _leaf_cfgs():
    """(sim, {set: sections}, {(set, section, name): leaf}) with every non-geometry parameter a torch leaf."""
    sim, cfg = defaults.default_configs_for("na1")
    leaves = {}
    for which, sections in cfg.items():
        for sec, d in sections.items():
            if sec == "geometry":
                continue
            for k, v in d.items():
                t = torch.tensor(float(v), dtype=torch.float64, requires_grad=True)
                d[k] = t
                leaves[(which, sec, k)] = t
    return sim, cfg, leaves


def _flat(cfg=None, kt=None):
    sim, dflt = defaults.default_configs_for("na1")
    cfg = dflt if cfg is None else cfg
    named = fp.derive_flat_na1(cfg["dna"], cfg["rna"], cfg["drh"], kt=sim["kT"] if kt is None else kt, salt_conc=0.5, half_charged_ends=False)
    return fp.pack_flat_na1(named, _lib.param_names())


def _system(top, is_rna, box, dtype, flat=None, pairs=None):
    from mythos_amd.hip_system import OxdnaSystem

    s = OxdnaSystem(4, top.seq, top.is_end, top.bonded_neighbors, box=box, dtype=dtype, is_rna=is_rna)
    s.set_params((_flat() if flat is None else flat).detach())
    s.set_neighbors(top.unbonded_neighbors if pairs is None else pairs)
    return s


@pytest.mark.parametrize("name", H.NA1_CASES)
def test_terms_forces_and_gradients(name):
    top, traj, split, is_rna = H.load_golden_na1(name)
    P = H.oracle_params_na1()
    seq, is_end, b, u = H.topo_tensors(top)
    rna = torch.as_tensor(is_rna)
    frames = [0, 33, 99]
    ref = np.array([orc.energy_terms_na1(P, torch.as_tensor(traj.center[f]), torch.as_tensor(traj.quaternions[f]), seq, rna, is_end, b, u,
                                         box=traj.box_size).numpy() for f in range(100)])
    for dtype, tol in ((torch.float64, 1e-9), (torch.float32, 1e-3)):
        s = _system(top, is_rna, traj.box_size, dtype)
        c = torch.as_tensor(traj.center, dtype=dtype, device=s.device)
        q = torch.as_tensor(traj.quaternions, dtype=dtype, device=s.device)
        e, gc, gq, _ = s.energy(c, q, grads=True)
        e = e.cpu().numpy()
        np.testing.assert_allclose(e, ref, rtol=tol, atol=tol * np.abs(ref).max())
        if dtype == torch.float64:
            for k, term in enumerate(H.SPLIT_COLUMNS[1:9]):
                if term == "coaxial_stacking" and name == "simple-coax-dna-dna-rna":
                    continue  # left out by the reference: oxNA's standalone code reads the hybrid spring constant as 0
                np.testing.assert_allclose(np.around(e[:, k] / top.n_nucleotides, 6), split[:, 1 + k], atol=H.NA1_TERM_ATOL[term] + 5e-7, err_msg=term)
        # the energy-only instantiation: the same numbers up to the rounding of a differently scheduled sum
        np.testing.assert_allclose(s.energy(c, q)[0].cpu().numpy(), e, rtol=1e-12 if dtype == torch.float64 else 1e-5, atol=1e-12 if dtype == torch.float64 else 1e-5)
        for f in frames:
            _, rc, rq = orc.energy_and_grads_na1(P, torch.as_tensor(traj.center[f]), torch.as_tensor(traj.quaternions[f]), seq, rna, is_end, b, u,
                                                 box=traj.box_size)
            for got, want in ((gc[f], rc), (gq[f], rq)):
                want = want.numpy()
                err = np.abs(got.double().cpu().numpy() - want).max()
                assert err <= (1e-8 if dtype == torch.float64 else 1e-3) * max(1.0, np.abs(want).max()), (dtype, f, err)


@pytest.mark.parametrize("name", ["simple-helix-dna-rna", "simple-coax-dna-dna-rna", "simple-coax-rna-rna-rna"])
def test_parameter_gradients_chain_rule(name):
    """dU/dtheta of every independent parameter of the three vectors (oxDNA2, oxRNA2, hybrid) and kT: HIP partials + host
    chain rule against the oracle's autograd."""
    top, traj, _, is_rna = H.load_golden_na1(name)
    frames = [7, 70]
    sim, cfg, leaves = _leaf_cfgs()
    kt = torch.tensor(sim["kT"], dtype=torch.float64, requires_grad=True)
    flat = _flat(cfg, kt)
    assert flat.shape == (3 * len(_lib.param_names()),)
    s = _system(top, is_rna, traj.box_size, torch.float64, flat=flat)
    c = torch.as_tensor(traj.center[frames], device=s.device)
    q = torch.as_tensor(traj.quaternions[frames], device=s.device)
    e, _, _, gflat = s.energy(c, q, param_grads=True)
    assert gflat.shape == (2, flat.shape[0])
    keys = list(leaves)
    sim2, cfg2, leaves2 = _leaf_cfgs()
    kt2 = torch.tensor(sim2["kT"], dtype=torch.float64, requires_grad=True)
    P = orc.init_all_na1(cfg2["dna"], cfg2["rna"], cfg2["drh"], kt=kt2, salt_conc=0.5, half_charged_ends=False)
    seq, is_end, b, u = H.topo_tensors(top)
    for k, f in enumerate(frames):
        g = torch.autograd.grad(flat, [leaves[kk] for kk in keys] + [kt], grad_outputs=gflat[k].cpu(), retain_graph=True, allow_unused=True)
        got = np.array([0.0 if x is None else float(x) for x in g])
        U = orc.energy_terms_na1(P, torch.as_tensor(traj.center[f]), torch.as_tensor(traj.quaternions[f]), seq, torch.as_tensor(is_rna), is_end, b, u,
                                 box=traj.box_size).sum()
        w = torch.autograd.grad(U, [leaves2[kk] for kk in keys] + [kt2], retain_graph=True, allow_unused=True)
        want = np.array([0.0 if x is None else float(x) for x in w])
        assert abs(e[k].sum().item() - U.item()) < 1e-9 * abs(U.item())
        bad = np.abs(got - want) > 1e-5 * np.maximum(np.abs(want), 1e-3 * np.abs(want).max())
        assert not bad.any(), [(keys[i] if i < len(keys) else "kt", got[i], want[i]) for i in np.nonzero(bad)[0]]
        live_sets = {kk[0] for kk, x in zip(keys, want) if x != 0.0}
        if name == "simple-helix-dna-rna":
            assert live_sets == {"dna", "rna", "drh"}  # bonded terms of both strands, hybrid unbonded terms


@pytest.mark.parametrize("bonded", [False, True])
def test_random_dimers(bonded):
    """All four kinds of pair in random relative poses (tests/helpers.py random_dimers): the hybrid coaxial term has no
    golden (na1/tests/test_integration.py:404-406) - here it is live in dozens of DNA-RNA dimers."""
    top, c0, q0, live = H.random_dimers(4, bonded)
    is_rna = np.asarray(top.nt_type) == 2
    assert min(live["pairs dna-dna / rna-rna / hybrid"]) >= 30
    pairs = np.zeros((0, 2), np.int64) if bonded else np.arange(top.n_nucleotides).reshape(-1, 2)
    P = H.oracle_params_na1()
    seq, is_end, b, _ = H.topo_tensors(top)
    rna = torch.as_tensor(is_rna)
    ct, qt, u = torch.as_tensor(c0), torch.as_tensor(q0), torch.as_tensor(pairs)
    e_ref = orc.energy_terms_na1(P, ct, qt, seq, rna, is_end, b, u).numpy()
    _, rc, rq = orc.energy_and_grads_na1(P, ct, qt, seq, rna, is_end, b, u)
    for dtype, tol in ((torch.float64, 1e-9), (torch.float32, 1e-3)):
        s = _system(top, is_rna, None, dtype, pairs=pairs)
        c = torch.as_tensor(c0[None], dtype=dtype, device=s.device)
        q = torch.as_tensor(q0[None], dtype=dtype, device=s.device)
        e, gc, gq, _ = s.energy(c, q, grads=True)
        np.testing.assert_allclose(e[0].cpu().numpy(), e_ref, rtol=tol, atol=tol * np.abs(e_ref).max())
        for got, want in ((gc, rc), (gq, rq)):
            want = want.numpy()
            got = got[0].double().cpu().numpy()
            rms = np.sqrt((want**2).mean())
            assert np.abs(got - want).max() <= tol * (np.abs(want).max() if dtype == torch.float32 else rms), (dtype, np.abs(got - want).max(), rms)


@pytest.mark.parametrize("unfused", [False, True])
@pytest.mark.parametrize("name", ["simple-helix-dna-rna", "simple-coax-dna-dna-rna", "simple-helix-rna-rna"])
def test_langevin_steps_match_the_oracle(name, unfused, md_lanes):
    """The fused step kernel's oxNA instantiation (per row entry: the parameter set of the pair's kind; sites by the type of
    each nucleotide), and the two-launch path behind LangevinIntegrator.set_unfused (the energy kernel's forces + an integrator
    kernel): six fp64 steps against LangevinOracle on the same Philox stream - positions, quaternions, potential and kinetic
    energies of every step, final momenta; then the same trajectory in two advances."""
    from mythos_amd.hip_system import LangevinIntegrator
    from oracle.langevin_oracle import LangevinOracle

    top, traj, _, is_rna = H.load_golden_na1(name)
    kT = 296.15 * 0.1 / 300.0
    gam_t, gam_r, seed = kT / 2.5, kT / 7.5, 0xBADC0FFEE
    s = _system(top, is_rna, traj.box_size, torch.float64)
    integ = LangevinIntegrator(s, dt=0.003, kT=kT, gamma_t=gam_t, gamma_r=gam_r, mass=1.0, inertia=(1.0, 1.2, 0.9), seed=seed)
    if unfused:
        integ.set_unfused()
    c = torch.as_tensor(traj.center[2], device=s.device).contiguous()
    q = torch.as_tensor(traj.quaternions[2], device=s.device).contiguous()
    p, L = integ.init_momenta()
    x, qq, pp, LL = (t.cpu().numpy().copy() for t in (c, q, p, L))
    start = [t.clone() for t in (c, q, p, L)]
    tc, tq, et = integ.run(c, q, p, L, 6, save_every=1)
    lo = LangevinOracle(4, H.oracle_params_na1(), H.topo_tensors(top), traj.box_size, 0.003, kT, gam_t, gam_r, 1.0, (1.0, 1.2, 0.9),
                        seed=seed, is_rna=is_rna)
    for k in range(6):
        x, qq, pp, LL, u = lo.step(x, qq, pp, LL)
        np.testing.assert_allclose(tc[k].cpu().numpy(), x, rtol=0, atol=1e-10)
        np.testing.assert_allclose(tq[k].cpu().numpy(), qq, rtol=0, atol=1e-10)
        assert abs(et[k, :8].sum().item() - u) < 1e-8 * abs(u)
        ke_t, ke_r = lo.kinetic(pp, LL)
        assert abs(et[k, 8].item() - ke_t) < 1e-9 * ke_t and abs(et[k, 9].item() - ke_r) < 1e-9 * ke_r
    np.testing.assert_allclose(p.cpu().numpy(), pp, atol=1e-9)
    np.testing.assert_allclose(L.cpu().numpy(), LL, atol=1e-9)
    assert integ.step == 6
    # resident: load; advance(2); advance(4); store == run(6), bit for bit (one force evaluation per step either way)
    integ2 = LangevinIntegrator(s, dt=0.003, kT=kT, gamma_t=gam_t, gamma_r=gam_r, mass=1.0, inertia=(1.0, 1.2, 0.9), seed=seed)
    if unfused:
        integ2.set_unfused()
    c2, q2, p2, L2 = [t.clone() for t in start]
    integ2.load(c2, q2, p2, L2)
    integ2.advance(2)
    integ2.advance(4)
    integ2.store(c2, q2, p2, L2)
    assert torch.equal(c2, c) and torch.equal(q2, q) and torch.equal(p2, p) and torch.equal(L2, L)


def test_simulator_runs_a_hybrid_duplex_with_a_dynamic_list():
    """HipMDSimulator (the fused step kernel) on the DNA-RNA golden helix, fp32, Verlet list rebuilt every 10 steps: finite, the helix holds, the
    mean potential energy stays where oxDNA's own run (interaction_type = NA) has it; and the static all-pairs list gives
    the same trajectory while no pair crosses the list range."""
    from mythos_amd.simulators.hip_md import HipMDSimulator, StaticSimulatorParams, nvt_langevin
    from mythos_amd.simulators.neighbors import NoNeighborList, VerletNeighborList

    top, traj, split, _ = H.load_golden_na1("simple-helix-dna-rna")
    disp, shift = space.periodic(20.0)
    ef = na1.create_default_energy_fn(top, disp)
    sp = StaticSimulatorParams(seq=top.seq, mass=(1.0, (1.0, 1.0, 1.0)), gamma=(KT / 2.5, KT / 7.5), bonded_neighbors=top.bonded_neighbors,
                               checkpoint_every=0, dt=0.003, kT=KT)
    init = _states(traj, torch.float32)[0]
    outs = []
    for nb in (VerletNeighborList(3.25, 0.6, 10), NoNeighborList(unbonded_nbrs=top.unbonded_neighbors)):
        sim = HipMDSimulator(energy_fn=ef, simulator_params=sp, space=(disp, shift), simulator_init=nvt_langevin, neighbors=nb,
                             save_every=50, dtype=torch.float32)
        outs.append(sim.run({}, init, 3000, key=5).observables[0])
    tr = outs[0]
    assert tr.center.shape == (60, top.n_nucleotides, 3) and torch.isfinite(tr.center).all()
    # 16 nucleotides, every pair inside the list range at all times: the two lists hold the same pairs, the sums differ in order only
    assert (outs[0].center[:5] - outs[1].center[:5]).abs().max().item() < 1e-3
    states = RigidBody(center=tr.center.double(), orientation=Quaternion(vec=tr.orientation.vec.double()))
    u = ef.map(states).cpu().numpy() / top.n_nucleotides
    u_gold = split[:, 1:9].sum(1)
    assert abs(u[15:].mean() - u_gold.mean()) < 4.0 * u_gold.std() + 0.02, (u[15:].mean(), u_gold.mean(), u_gold.std())
    hb = ef.compute_terms(states)[:, 4].cpu().numpy() / top.n_nucleotides
    assert hb[15:].mean() < -0.15  # the hybrid duplex stays hybridised (golden: about -0.3)


def test_what_the_boundary_refuses():
    from mythos_amd.hip_system import LangevinIntegrator, OxdnaSystem

    top, traj, _, is_rna = H.load_golden_na1("simple-helix-dna-rna")
    with pytest.raises(ValueError, match="is_rna"):
        OxdnaSystem(4, top.seq, top.is_end, top.bonded_neighbors, box=traj.box_size, dtype=torch.float64)
    s = OxdnaSystem(4, top.seq, top.is_end, top.bonded_neighbors, box=traj.box_size, dtype=torch.float64, is_rna=is_rna)
    with pytest.raises(ValueError, match="expected 783"):  # one vector where three are needed
        s.set_params(_flat()[: len(_lib.param_names())])
    s.set_params(_flat())
    with pytest.raises(ValueError, match="hydrogen bonding only"):  # stacking weights of an oxNA system take no distribution
        s.set_pseq(np.full((top.n_nucleotides, 4), 0.25), np.full(top.n_nucleotides, -1), np.zeros((0, 4)), terms=3)
    s.set_neighbors(top.unbonded_neighbors)
    s.set_pseq(np.full((top.n_nucleotides, 4), 0.25), np.full(top.n_nucleotides, -1), np.zeros((0, 4)), terms=2)
    c1 = torch.as_tensor(traj.center[:1], device=s.device)
    q1 = torch.as_tensor(traj.quaternions[:1], device=s.device)
    # dU/d(distribution) is there for oxNA systems too (round 3; checked against the oracle further down): shapes only here
    out = s.energy(c1, q1, param_grads=True, pseq_grads=True)
    assert tuple(out[4].shape) == (1, top.n_nucleotides, 4) and tuple(out[5].shape) == (1, 1, 4)
    s.set_pseq()
    other = OxdnaSystem(2, top.seq, top.is_end, top.bonded_neighbors, box=traj.box_size, dtype=torch.float64)
    with pytest.raises(ValueError, match="only an oxNA system"):
        t = np.ascontiguousarray(is_rna, dtype=np.uint8)
        _lib.check(other._lib.mythos_oxdna_set_nucleotide_types(other._h, t.ctypes.data_as(_lib.c_uint8_p)), "set_nucleotide_types")


# ---- through the reference-shaped classes, as mythos/energy/na1/tests/test_integration.py does ---------------------------------
from mythos_amd.energy import na1  # noqa: E402
from mythos_amd.energy.base import Quaternion, RigidBody, space  # noqa: E402

KT = 296.15 * 0.1 / 300.0
HELICES = ["simple-helix-dna-dna", "simple-helix-rna-rna", "simple-helix-dna-rna", "simple-helix-rna-dna"]
TERM_CASES = (
    [(n, "Fene", "FeneConfiguration", "fene", {}) for n in HELICES[:3]]
    + [(n, "BondedExcludedVolume", "BondedExcludedVolumeConfiguration", "bonded_excluded_volume", {}) for n in HELICES[:3]]
    + [(n, "Stacking", "StackingConfiguration", "stacking", {"kt": KT}) for n in HELICES[:3]]
    + [(n, "UnbondedExcludedVolume", "UnbondedExcludedVolumeConfiguration", "unbonded_excluded_volume", {}) for n in HELICES]
    + [(n, "CrossStacking", "CrossStackingConfiguration", "cross_stacking", {}) for n in HELICES]
    + [(n, "HydrogenBonding", "HydrogenBondingConfiguration", "hydrogen_bonding", {}) for n in HELICES]
    + [(n, "CoaxialStacking", "CoaxialStackingConfiguration", "coaxial_stacking", {}) for n in ("simple-coax-dna-dna-dna", "simple-coax-rna-rna-rna")]
    + [(n, "Debye", "DebyeConfiguration", "debye", {"kt": KT, "salt_conc": 0.5, "half_charged_ends": False}) for n in HELICES]
)


def _states(traj, dtype=torch.float64):
    dev = torch.device("cuda", 0)
    return RigidBody(center=torch.as_tensor(traj.center, dtype=dtype, device=dev),
                     orientation=Quaternion(vec=torch.as_tensor(traj.quaternions, dtype=dtype, device=dev)))


@pytest.mark.parametrize(("name", "cls", "cfg_cls", "section", "extra"), TERM_CASES)
def test_single_term_matches_split_energy(name, cls, cfg_cls, section, extra):
    """The parametrisation of na1/tests/test_integration.py:152-496, case for case."""
    top, traj, split, _ = H.load_golden_na1(name)
    default_params = na1.default_configs()[1]
    displacement_fn, _ = space.periodic(20.0)
    energy_config = getattr(na1, cfg_cls)(**(default_params[section] | {"nt_type": top.nt_type} | extra))
    energy_fn = getattr(na1, cls)(displacement_fn=displacement_fn, transform_fn=na1.default_transform_fn(), topology=top,
                                  params=energy_config.init_params())
    energy = energy_fn.map(_states(traj)).cpu().numpy()
    energy = np.around(energy / top.n_nucleotides, 6)
    np.testing.assert_allclose(energy, split[:, H.SPLIT_COLUMNS.index(section)], atol=H.NA1_TERM_ATOL[section] + 5e-7)


def test_zero_stacking_weights_give_zero_stacking():
    """na1/tests/test_integration.py:236-267, the rna-dna case: all-zero sequence weights for both strand types."""
    top, traj, _, _ = H.load_golden_na1("simple-helix-rna-dna")
    default_params = na1.default_configs()[1]
    z = torch.zeros(4, 4, dtype=torch.float64)
    cfg = na1.StackingConfiguration(**(default_params["stacking"] | {"kt": KT, "nt_type": top.nt_type}), dna_ss_stack_weights=z, rna_ss_stack_weights=z)
    fn = na1.Stacking(displacement_fn=space.periodic(20.0)[0], transform_fn=na1.default_transform_fn(), topology=top, params=cfg.init_params())
    assert float(fn.map(_states(traj)).abs().max()) == 0.0


def test_default_function_total_and_autograd():
    top, traj, split, is_rna = H.load_golden_na1("simple-helix-dna-rna")
    displacement_fn, _ = space.periodic(20.0)
    ef = na1.create_default_energy_fn(top, displacement_fn)
    states = _states(traj)
    e = ef.map(states).cpu().numpy() / top.n_nucleotides
    np.testing.assert_allclose(e, split[:, 1:9].sum(1), atol=2e-4)
    terms = ef.compute_terms(states)
    assert terms.shape == (100, 8)
    # with_params on prefixed names + autograd against the oracle: one parameter of each set, and the shared kT
    opt = {"dna_eps_stack_base": torch.tensor(1.3523, dtype=torch.float64, requires_grad=True),
           "rna_a_stack_9": torch.tensor(1.3, dtype=torch.float64, requires_grad=True),
           "drh_eps_hb": torch.tensor(1.5, dtype=torch.float64, requires_grad=True),
           "drh_k_cross": torch.tensor(44.535, dtype=torch.float64, requires_grad=True)}
    frames = [4, 44, 84]
    sub = RigidBody(center=states.center[frames], orientation=Quaternion(vec=states.orientation.vec[frames]))
    u = ef.with_params(opt).map(sub).sum()
    got = torch.autograd.grad(u, list(opt.values()))
    sim, cfg = defaults.default_configs_for("na1")
    leaves = {k: torch.tensor(float(v.detach()), dtype=torch.float64, requires_grad=True) for k, v in opt.items()}
    cfg["dna"]["stacking"]["eps_stack_base"] = leaves["dna_eps_stack_base"]
    cfg["rna"]["stacking"]["a_stack_9"] = leaves["rna_a_stack_9"]
    cfg["drh"]["hydrogen_bonding"]["eps_hb"] = leaves["drh_eps_hb"]
    cfg["drh"]["cross_stacking"]["k_cross"] = leaves["drh_k_cross"]
    P = orc.init_all_na1(cfg["dna"], cfg["rna"], cfg["drh"], kt=sim["kT"], salt_conc=0.5, half_charged_ends=False)
    seq, is_end, b, uu = H.topo_tensors(top)
    uo = sum(orc.energy_terms_na1(P, torch.as_tensor(traj.center[f]), torch.as_tensor(traj.quaternions[f]), seq, torch.as_tensor(is_rna), is_end, b, uu,
                                  box=traj.box_size).sum() for f in frames)
    want = torch.autograd.grad(uo, [leaves[k] for k in opt])
    assert abs(float(u.detach()) - float(uo.detach())) <= 1e-9 * abs(float(uo.detach()))
    for k, g, w in zip(opt, got, want):
        assert abs(float(g) - float(w)) <= 1e-7 * max(1.0, abs(float(w))), (k, float(g), float(w))
        assert abs(float(w)) > 1e-6, k


def test_probabilistic_sequence_through_the_hydrogen_bonding_term():
    """The reference's na1 HydrogenBondingConfiguration carries ``pseq`` / ``pseq_constraints`` and hands them to its
    three sub-configurations (na1/hydrogen_bonding.py:127-128, 243-304): a one-hot distribution reproduces the discrete
    energies of the hybrid golden, a soft one equals the oracle's expectation - with the hybrid weight table for the
    DNA-RNA pairs."""
    from mythos_amd.input import sequence_constraints as scm

    top, traj, _, is_rna = H.load_golden_na1("simple-helix-dna-rna")
    disp, _ = space.periodic(20.0)
    ef = na1.create_default_energy_fn(top, disp)
    frames = [0, 25, 50, 75]
    states = _states(traj)
    sub = RigidBody(center=states.center[frames], orientation=Quaternion(vec=states.orientation.vec[frames]))
    e_d = ef.compute_terms(sub).cpu().numpy()
    sc = scm.from_bps(top.n_nucleotides, np.array([[1, 14], [3, 12]]))
    e_hot = ef.with_params(pseq=scm.dseq_to_pseq(top.seq, sc), pseq_constraints=sc).compute_terms(sub).cpu().numpy()
    np.testing.assert_allclose(e_hot, e_d, rtol=0, atol=1e-12)
    rng = np.random.default_rng(3)

    def dist(rows):
        a = rng.random((rows, 4)) + 0.05
        return a / a.sum(1, keepdims=True)

    up, bp = dist(sc.n_unpaired), dist(sc.n_bp)
    e_soft = ef.with_params(pseq=(up, bp), pseq_constraints=sc).compute_terms(sub).cpu().numpy()
    sim, cfg = defaults.default_configs_for("na1")
    for which in cfg:
        cfg[which]["hydrogen_bonding"].update(pseq=(up, bp), pseq_constraints=sc)
    P = orc.init_all_na1(cfg["dna"], cfg["rna"], cfg["drh"], kt=sim["kT"], salt_conc=0.5, half_charged_ends=False)
    seq, is_end, b, u = H.topo_tensors(top)
    want = np.array([orc.energy_terms_na1(P, torch.as_tensor(traj.center[f]), torch.as_tensor(traj.quaternions[f]), seq, torch.as_tensor(is_rna),
                                          is_end, b, u, box=traj.box_size).numpy() for f in frames])
    np.testing.assert_allclose(e_soft, want, rtol=1e-10, atol=1e-10)
    assert np.abs(e_soft[:, 4] - e_d[:, 4]).max() > 1e-2   # the distribution matters ...
    np.testing.assert_allclose(np.delete(e_soft, 4, axis=1), np.delete(e_d, 4, axis=1), rtol=0, atol=1e-12)  # ... to hydrogen bonding only


def test_gradient_with_respect_to_the_distribution_of_a_hybrid_system():
    """dU/d(pseq) for oxNA (na1/hydrogen_bonding.py:127-128, 243-304: the distribution reaches the hydrogen-bonding weight of
    whichever parameter set a pair takes): the kernel's dU/d(marginals), dU/d(type probabilities) of the model-4
    instantiation (mythos_oxdna_energy_dpseq), carried to the (unpaired, base-pair) arrays by autograd, against oracle
    autograd through compute_seq_dep_weight on the DNA-RNA golden; three frames weighted differently, a parameter
    gradient in the same backward pass."""
    from mythos_amd.input import sequence_constraints as scm

    top, traj, _, is_rna = H.load_golden_na1("simple-helix-dna-rna")
    disp, _ = space.periodic(20.0)
    n = top.n_nucleotides
    sc = scm.from_bps(n, np.array([[1, 14], [3, 12], [6, 9]]))
    rng = np.random.default_rng(17)

    def dist(rows):
        a = rng.random((rows, 4)) + 0.05
        return a / a.sum(1, keepdims=True)

    up0, bp0 = dist(sc.n_unpaired), dist(sc.n_bp)
    frames = [3, 41, 88]
    states = _states(traj)
    sub = RigidBody(center=states.center[frames], orientation=Quaternion(vec=states.orientation.vec[frames]))
    up_h, bp_h = (torch.tensor(a, requires_grad=True) for a in (up0, bp0))
    eps_h = torch.tensor(0.97, dtype=torch.float64, requires_grad=True)
    ef = na1.create_default_energy_fn(top, disp)
    u_h = ef.with_params({"drh_eps_hb": eps_h}, pseq=(up_h, bp_h), pseq_constraints=sc).map(sub)
    coef = torch.tensor([1.0, -0.5, 2.0], dtype=torch.float64, device=u_h.device)
    g_h = torch.autograd.grad((u_h * coef).sum(), [up_h, bp_h, eps_h])
    up_o, bp_o = (torch.tensor(a, requires_grad=True) for a in (up0, bp0))
    eps_o = torch.tensor(0.97, dtype=torch.float64, requires_grad=True)
    sim, cfg = defaults.default_configs_for("na1")
    for which in cfg:
        cfg[which]["hydrogen_bonding"].update(pseq=(up_o, bp_o), pseq_constraints=sc)
    cfg["drh"]["hydrogen_bonding"]["eps_hb"] = eps_o
    P = orc.init_all_na1(cfg["dna"], cfg["rna"], cfg["drh"], kt=sim["kT"], salt_conc=0.5, half_charged_ends=False)
    seq, is_end, b, u = H.topo_tensors(top)
    u_o = torch.stack([orc.energy_terms_na1(P, torch.as_tensor(traj.center[f]), torch.as_tensor(traj.quaternions[f]), seq, torch.as_tensor(is_rna),
                                            is_end, b, u, box=traj.box_size).sum() for f in frames])
    np.testing.assert_allclose(u_h.detach().cpu().numpy(), u_o.detach().numpy(), rtol=1e-10)
    g_o = torch.autograd.grad((u_o * coef.cpu()).sum(), [up_o, bp_o, eps_o])
    for got, want, name in zip(g_h, g_o, ("unpaired", "base pairs", "drh_eps_hb")):
        got, want = got.detach().cpu().numpy(), want.detach().numpy()
        assert np.abs(want).max() > 1e-3, name
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-9 * max(1.0, np.abs(want).max()), err_msg=name)
