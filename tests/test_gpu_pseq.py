"""Probabilistic sequences through the HIP energy kernel (mythos_oxdna_set_pseq): the expected stacking / hydrogen-
bonding weight of every pair is formed inside the kernel from per-nucleotide marginals.

 * one-hot distribution == the discrete sequence == oxDNA's golden energies (dna1/tests/test_integration.py:192-293,
   use_pseq=True);
 * the reference's own check, brute-force enumeration over every allowed sequence with the DISCRETE kernel
   (dna1/tests/test_expected_energies.py:162-328, atol 1e-4), in both precisions;
 * energies, forces and dU/dtheta against the oracle's restatement of compute_seq_dep_weight (energy/utils.py:45-132);
 * dynamics with a sequence distribution: the stepping kernel's PSEQ instantiation against LangevinOracle, and
   through HipMDSimulator (one-hot == discrete; replicas).
"""

import numpy as np
import pytest
import torch

from mythos_amd.energy import dna1, dna2
from mythos_amd.energy.base import Quaternion, RigidBody, space
from mythos_amd.input import sequence_constraints as scm
from oracle import oxdna_oracle as orc
from tests import helpers as H
from tests.test_pseq_cpu import _helix4, enumerate_sequences

pytestmark = pytest.mark.gpu


def _states(traj, frames, dtype):
    dev = torch.device("cuda", 0)
    return RigidBody(center=torch.as_tensor(traj.center[frames], dtype=dtype, device=dev),
                     orientation=Quaternion(vec=torch.as_tensor(traj.quaternions[frames], dtype=dtype, device=dev)))


@pytest.mark.parametrize(("case", "weights"), [("simple-helix", False), ("simple-helix-seq-dep", True)])
def test_one_hot_distribution_equals_discrete_sequence_and_goldens(case, weights):
    top, traj, split, _ = H.load_golden(1, case)
    disp, _ = space.periodic(traj.box_size)
    ef = dna1.create_default_energy_fn(topology=top, displacement_fn=disp)
    if weights:
        ss = H.read_ss_weights(H.GOLDEN / "dna1" / case / "seq_dep.dat")
        ef = ef.with_params(ss_stack_weights=ss["ss_stack_weights"], eps_stack_kt_coeff=ss["eps_stack_kt_coeff"], ss_hb_weights=ss["ss_hb_weights"])
    frames = list(range(0, 100, 9))
    st = _states(traj, frames, torch.float64)
    sc = scm.from_bps(top.n_nucleotides, np.array([[0, 15], [3, 12]]) if not weights else np.zeros((0, 2), dtype=np.int32))
    soft = ef.with_params(pseq=scm.dseq_to_pseq(top.seq, sc), pseq_constraints=sc)
    e_d, e_p = ef.compute_terms(st).cpu().numpy(), soft.compute_terms(st).cpu().numpy()
    np.testing.assert_allclose(e_p, e_d, rtol=0, atol=1e-12)
    n = top.n_nucleotides
    np.testing.assert_allclose(np.around(e_p[:, 4] / n, 6), split[frames, 5], atol=1e-3)
    np.testing.assert_allclose(np.around(e_p[:, 2] / n, 6), split[frames, 3], atol=1e-6)
    # switching back: the next discrete evaluation on the same cached system is discrete again
    np.testing.assert_allclose(ef.compute_terms(st).cpu().numpy(), e_d, rtol=0, atol=0)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_expected_energy_equals_brute_force_enumeration_with_the_discrete_kernel(dtype):
    top, traj = _helix4()
    disp, _ = space.periodic(traj.box_size)
    sc = scm.from_bps(8, np.array([[0, 7], [1, 6], [2, 5]]))
    rng = np.random.default_rng(3)

    def dist(rows):
        a = rng.random((rows, 4))
        return a / a.sum(1, keepdims=True)

    w_hb, w_st, up, bp = dist(4), dist(4), dist(sc.n_unpaired), dist(sc.n_bp)
    ef = dna1.create_default_energy_fn(topology=top, displacement_fn=disp).with_params(ss_hb_weights=w_hb, ss_stack_weights=w_st)
    st = _states(traj, list(range(0, 100, 5)), dtype)
    got = ef.with_params(pseq=(up, bp), pseq_constraints=sc).compute_terms(st).cpu().numpy()
    want = np.zeros_like(got)
    for seq, prob in enumerate_sequences(sc, up, bp):
        want += prob * ef.with_props(seq=seq.astype(np.int32)).compute_terms(st).cpu().numpy()
    assert np.abs(want[:, 4]).max() > 1e-2 and np.abs(want[:, 2]).max() > 1e-2
    tol = 1e-10 if dtype == torch.float64 else 1e-4  # the reference's own tolerance is 1e-4
    np.testing.assert_allclose(got, want, rtol=0, atol=tol)


def test_energies_forces_and_parameter_gradients_match_the_oracle():
    top, traj, _, _ = H.load_golden(2, "simple-helix")
    disp, _ = space.periodic(traj.box_size)
    n = top.n_nucleotides
    sc = scm.from_bps(n, np.array([[1, 14], [2, 13], [5, 10], [7, 8]]))
    rng = np.random.default_rng(5)

    def dist(rows):
        a = rng.random((rows, 4)) + 0.05
        return a / a.sum(1, keepdims=True)

    up, bp, w_st = dist(sc.n_unpaired), dist(sc.n_bp), rng.random((4, 4)) + 0.5
    opt = {"eps_hb": 1.0678, "eps_stack_kt_coeff": 2.6717, "a_stack": 6.0, "theta0_hb_4": float(np.pi), "k_cross": 47.5}
    leaves_h = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in opt.items()}
    leaves_o = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in opt.items()}
    ef = dna2.create_default_energy_fn(topology=top, displacement_fn=disp).with_params(ss_stack_weights=w_st)
    ef = ef.with_params(pseq=(up, bp), pseq_constraints=sc)
    P = H.oracle_params(2, half_charged_ends=True, overrides={
        "stacking": {"ss_stack_weights": w_st, "eps_stack_kt_coeff": leaves_o["eps_stack_kt_coeff"], "a_stack": leaves_o["a_stack"],
                     "pseq": (up, bp), "pseq_constraints": sc},
        "hydrogen_bonding": {"eps_hb": leaves_o["eps_hb"], "theta0_hb_4": leaves_o["theta0_hb_4"], "pseq": (up, bp), "pseq_constraints": sc},
        "cross_stacking": {"k_cross": leaves_o["k_cross"]}})
    tt = H.topo_tensors(top)
    for f in (4, 61):
        dev = torch.device("cuda", 0)
        c = torch.as_tensor(traj.center[f], device=dev).requires_grad_(True)
        q = torch.as_tensor(traj.quaternions[f], device=dev).requires_grad_(True)
        body = RigidBody(center=c, orientation=Quaternion(vec=q))
        e_h = ef.with_params(leaves_h)(body)
        co = torch.as_tensor(traj.center[f]).requires_grad_(True)
        qo = torch.as_tensor(traj.quaternions[f]).requires_grad_(True)
        e_o = orc.energy(2, P, co, qo, *tt, traj.box_size)
        assert abs(e_h.item() - e_o.item()) <= 1e-10 * abs(e_o.item())
        g_h = torch.autograd.grad(e_h, [c, q, *leaves_h.values()])
        g_o = torch.autograd.grad(e_o, [co, qo, *leaves_o.values()], retain_graph=True)
        for a, b in zip(g_h, g_o):
            assert (a.cpu() - b).abs().max().item() <= 1e-8 * max(1.0, b.abs().max().item())
    # the terms each carry the distribution: only stacking soft, hydrogen bonding discrete
    only = dna2.create_default_energy_fn(topology=top, displacement_fn=disp).with_params(ss_stack_weights=w_st)
    fns = [fn.with_params(pseq=(up, bp), pseq_constraints=sc) if fn.term == "stacking" else fn for fn in only.energy_fns]
    mixed = only.replace(energy_fns=fns)
    st = _states(traj, [4], torch.float64)
    tm, tf, td = (x.compute_terms(st).cpu().numpy()[0] for x in (mixed, ef, only))
    assert abs(tm[2] - tf[2]) < 1e-12 and abs(tm[4] - td[4]) < 1e-12 and abs(tm[2] - td[2]) > 1e-3


@pytest.mark.parametrize("model", [2, 1])
def test_gradient_with_respect_to_the_distribution(model):
    """dU/d(pseq) - what jax.grad of the reference's energy function returns for the two pseq arrays (sequence design):
    the kernel's dU/d(marginals) and dU/d(type probabilities) (mythos_oxdna_energy_dpseq), carried to the (unpaired,
    base-pair) arrays by autograd, against the oracle's autograd through compute_seq_dep_weight.  Unpaired-unpaired,
    unpaired-paired, same-pair and different-pair contacts all occur; frames and a parameter gradient ride along."""
    top, traj, _, _ = H.load_golden(model, "simple-helix")
    mod = dna2 if model == 2 else dna1
    disp, _ = space.periodic(traj.box_size)
    n = top.n_nucleotides
    sc = scm.from_bps(n, np.array([[1, 14], [2, 13], [5, 10], [7, 8]]))
    rng = np.random.default_rng(11)

    def dist(rows):
        a = rng.random((rows, 4)) + 0.05
        return a / a.sum(1, keepdims=True)

    up0, bp0, w_st, w_hb = dist(sc.n_unpaired), dist(sc.n_bp), rng.random((4, 4)) + 0.5, rng.random((4, 4)) + 0.2
    frames = [4, 33, 61]
    st = _states(traj, frames, torch.float64)
    up_h, bp_h = (torch.tensor(a, requires_grad=True) for a in (up0, bp0))
    a_h = torch.tensor(6.0, dtype=torch.float64, requires_grad=True)
    ef = mod.create_default_energy_fn(topology=top, displacement_fn=disp).with_params(ss_stack_weights=w_st, ss_hb_weights=w_hb)
    u_h = ef.with_params({"a_stack": a_h}, pseq=(up_h, bp_h), pseq_constraints=sc).map(st)
    coef = torch.tensor([1.0, -0.5, 2.0], dtype=torch.float64, device=u_h.device)  # frames weigh differently: the backward pass scales per frame
    g_h = torch.autograd.grad((u_h * coef).sum(), [up_h, bp_h, a_h])
    up_o, bp_o = (torch.tensor(a, requires_grad=True) for a in (up0, bp0))
    a_o = torch.tensor(6.0, dtype=torch.float64, requires_grad=True)
    P = H.oracle_params(model, half_charged_ends=True, overrides={
        "stacking": {"ss_stack_weights": w_st, "a_stack": a_o, "pseq": (up_o, bp_o), "pseq_constraints": sc},
        "hydrogen_bonding": {"ss_hb_weights": w_hb, "pseq": (up_o, bp_o), "pseq_constraints": sc}})
    tt = H.topo_tensors(top)
    u_o = torch.stack([orc.energy(model, P, torch.as_tensor(traj.center[f]), torch.as_tensor(traj.quaternions[f]), *tt, traj.box_size) for f in frames])
    np.testing.assert_allclose(u_h.detach().cpu().numpy(), u_o.detach().numpy(), rtol=1e-10)
    g_o = torch.autograd.grad((u_o * coef.cpu()).sum(), [up_o, bp_o, a_o])
    for name, a, b in zip(("unpaired", "base pairs", "a_stack"), g_h, g_o):
        assert b.abs().max().item() > 1e-3, name
        assert (a.cpu() - b).abs().max().item() <= 1e-9 * max(1.0, b.abs().max().item()), (name, a, b)
    # no gradient asked for the distribution: the ordinary entry point serves the call, same energies
    u_plain = ef.with_params(pseq=(up0, bp0), pseq_constraints=sc).map(st)
    assert torch.equal(u_plain, u_h.detach())
    # fp32 frames: the same gradients to fp32 accuracy (the accumulators are fp64 either way)
    up_s, bp_s = (torch.tensor(a, requires_grad=True) for a in (up0, bp0))
    u_s = ef.with_params(pseq=(up_s, bp_s), pseq_constraints=sc).map(_states(traj, frames, torch.float32))
    g_s = torch.autograd.grad((u_s * coef.to(u_s.dtype)).sum(), [up_s, bp_s])
    for a, b in zip(g_s, g_o[:2]):
        assert (a.cpu().double() - b).abs().max().item() <= 2e-3 * b.abs().max().item()


@pytest.fixture(params=["short", "wide"])
def work_lists(request):
    """The PSEQ step kernels exist with both work-list widths since round 4 (ITEMS = 16, and 32 / 22 after an aborted
    launch); "wide" starts a run on the wide ones through mythos_debug_set."""
    from mythos_amd import _lib

    _lib.debug_set("md_items_big", 1 if request.param == "wide" else 0)
    yield request.param
    _lib.debug_set("md_items_big", 0)


@pytest.mark.parametrize(("model", "save_every"), [(1, 1), (2, 1), (2, 0)])
def test_langevin_steps_with_a_soft_distribution_match_the_oracle(model, save_every, work_lists, md_lanes):
    """A probabilistic sequence inside the dynamics (the reference's stacking / hydrogen-bonding configurations
    carry pseq into whatever energy function a simulator steps with, dna1/stacking.py:261-285, hydrogen_bonding.py:310-331):
    md_step_kernel's PSEQ instantiation, six fp64 steps on dna1/helix-4bp with a soft distribution - three constrained
    base pairs, two free nucleotides, random weight tables - against LangevinOracle on the same Philox stream (the
    oracle's energy takes the expectation through compute_seq_dep_weight); with and without stored steps (the two
    instantiations), and the run split in two advances."""
    from mythos_amd.hip_system import LangevinIntegrator
    from oracle.langevin_oracle import LangevinOracle
    from tests.test_gpu_md_at_size import _system

    top, traj = _helix4()
    sc = scm.from_bps(8, np.array([[0, 7], [1, 6], [2, 5]]))
    rng = np.random.default_rng(31)

    def dist(rows):
        a = rng.random((rows, 4)) + 0.05
        return a / a.sum(1, keepdims=True)

    up, bp, w_st, w_hb = dist(sc.n_unpaired), dist(sc.n_bp), rng.random((4, 4)) + 0.5, rng.random((4, 4)) + 0.2
    kT = 296.15 * 0.1 / 300.0
    from mythos_amd import _lib
    from mythos_amd.energy import flat_params as fp
    from mythos_amd.hip_system import OxdnaSystem
    from mythos_amd.input import defaults

    sim, cfg = defaults.default_configs_for(f"dna{model}")
    cfg["stacking"]["ss_stack_weights"] = torch.as_tensor(w_st)
    cfg["hydrogen_bonding"]["ss_hb_weights"] = torch.as_tensor(w_hb)
    flat = fp.pack_flat(fp.derive_flat(model, cfg, kt=sim["kT"], salt_conc=0.5, half_charged_ends=False), _lib.param_names())
    s = OxdnaSystem(model, top.seq, top.is_end, top.bonded_neighbors, box=traj.box_size, dtype=torch.float64)
    s.set_params(flat)
    s.set_neighbors(top.unbonded_neighbors)
    s.set_pseq(*scm.kernel_tables((up, bp), sc), 3)
    gam_t, gam_r, seed = kT / 2.5, kT / 7.5, 0xFACE
    integ = LangevinIntegrator(s, dt=0.004, kT=kT, gamma_t=gam_t, gamma_r=gam_r, mass=1.0, inertia=(1.0, 1.2, 0.9), seed=seed)
    c = torch.as_tensor(traj.center[3], device=s.device).contiguous()
    q = torch.as_tensor(traj.quaternions[3], device=s.device).contiguous()
    p, L = integ.init_momenta()
    x, qq, pp, LL = (t.cpu().numpy().copy() for t in (c, q, p, L))
    start = [t.clone() for t in (c, q, p, L)]
    tc, tq, et = integ.run(c, q, p, L, 6, save_every=save_every)
    P = H.oracle_params(model, half_charged_ends=False, overrides={
        "stacking": {"ss_stack_weights": w_st, "pseq": (up, bp), "pseq_constraints": sc},
        "hydrogen_bonding": {"ss_hb_weights": w_hb, "pseq": (up, bp), "pseq_constraints": sc}})
    lo = LangevinOracle(model, P, H.topo_tensors(top), traj.box_size, 0.004, kT, gam_t, gam_r, 1.0, (1.0, 1.2, 0.9), seed=seed)
    P_d = H.oracle_params(model, half_charged_ends=False, overrides={"stacking": {"ss_stack_weights": w_st}, "hydrogen_bonding": {"ss_hb_weights": w_hb}})
    lo_d = LangevinOracle(model, P_d, H.topo_tensors(top), traj.box_size, 0.004, kT, gam_t, gam_r, 1.0, (1.0, 1.2, 0.9), seed=seed)
    xd, qd, pd, Ld = x.copy(), qq.copy(), pp.copy(), LL.copy()
    for k in range(6):
        x, qq, pp, LL, u = lo.step(x, qq, pp, LL)
        xd, qd, pd, Ld, _ = lo_d.step(xd, qd, pd, Ld)
        if save_every:
            np.testing.assert_allclose(tc[k].cpu().numpy(), x, rtol=0, atol=1e-10)
            np.testing.assert_allclose(tq[k].cpu().numpy(), qq, rtol=0, atol=1e-10)
            assert abs(et[k, :8].sum().item() - u) < 1e-8 * abs(u)
    np.testing.assert_allclose(c.cpu().numpy(), x, rtol=0, atol=1e-10)
    np.testing.assert_allclose(p.cpu().numpy(), pp, atol=1e-9)
    np.testing.assert_allclose(L.cpu().numpy(), LL, atol=1e-9)
    assert np.abs(pp - pd).max() > 1e-5  # the distribution matters: the discrete sequence goes elsewhere
    # the same six steps as load / advance(2) / advance(4) / store
    c2, q2, p2, L2 = (t.clone() for t in start)
    integ2 = LangevinIntegrator(s, dt=0.004, kT=kT, gamma_t=gam_t, gamma_r=gam_r, mass=1.0, inertia=(1.0, 1.2, 0.9), seed=seed)
    integ2.load(c2, q2, p2, L2)
    integ2.advance(2)
    integ2.advance(4)
    integ2.store(c2, q2, p2, L2)
    for a, b in zip((c2, q2, p2, L2), (c, q, p, L)):
        assert torch.equal(a, b)
    # back to the discrete sequence on the same system: the plain instantiation again
    s.set_pseq()
    c3, q3, p3, L3 = (t.clone() for t in start)
    LangevinIntegrator(s, dt=0.004, kT=kT, gamma_t=gam_t, gamma_r=gam_r, mass=1.0, inertia=(1.0, 1.2, 0.9), seed=seed).run(c3, q3, p3, L3, 6)
    np.testing.assert_allclose(p3.cpu().numpy(), pd, atol=1e-9)


def test_simulator_runs_with_a_sequence_distribution():
    """HipMDSimulator with a pseq energy function (the reference's plugin surface: the same energy function object goes
    to the simulator): a one-hot distribution steps exactly as the discrete sequence does, fp32, dynamic list, and two
    replicas of a soft distribution run in one launch per step (each replica its own copy of the base pairs)."""
    import dataclasses as dc

    from mythos_amd.simulators.hip_md import HipMDSimulator, StaticSimulatorParams, nvt_langevin
    from mythos_amd.simulators.neighbors import NoNeighborList

    top, traj, _, _ = H.load_golden(2, "simple-helix")
    disp, shift = space.free()
    n = top.n_nucleotides
    sc = scm.from_bps(n, np.array([[0, 15], [3, 12]]))
    ef = dna2.create_default_energy_fn(topology=top, displacement_fn=disp)
    hot = ef.with_params(pseq=scm.dseq_to_pseq(top.seq, sc), pseq_constraints=sc)
    kT = 296.15 * 0.1 / 300.0
    sp = StaticSimulatorParams(seq=top.seq, mass=(1.0, (1.0, 1.0, 1.0)), gamma=(kT / 2.5, kT / 7.5), bonded_neighbors=top.bonded_neighbors,
                               checkpoint_every=0, dt=0.005, kT=kT)
    base = HipMDSimulator(energy_fn=ef, simulator_params=sp, space=(disp, shift), simulator_init=nvt_langevin,
                          neighbors=NoNeighborList(unbonded_nbrs=top.unbonded_neighbors), save_every=10)
    st = _states(traj, 0, torch.float32)
    a = base.run({}, st, 40, key=3).observables[0]
    b = dc.replace(base, energy_fn=hot).run({}, st, 40, key=3).observables[0]
    np.testing.assert_allclose(b.center.cpu().numpy(), a.center.cpu().numpy(), rtol=0, atol=2e-5)
    rng = np.random.default_rng(2)
    up = rng.random((sc.n_unpaired, 4)) + 0.1
    bp = rng.random((sc.n_bp, 4)) + 0.1
    soft = ef.with_params(pseq=(up / up.sum(1, keepdims=True), bp / bp.sum(1, keepdims=True)), pseq_constraints=sc)
    two = dc.replace(base, energy_fn=soft, n_replicas=2).run({}, st, 40, key=3).observables[0]
    assert two.center.shape[0] == 2 * 4 and torch.isfinite(two.center).all()
    one = dc.replace(base, energy_fn=soft).run({}, st, 40, key=3).observables[0]
    # replica 0 of the pair is the single run (same Philox stream per nucleotide index, same distribution) as far as fp32
    # on a grid offset allows over 40 thermostatted steps; a wrong distribution for a replica (base pairs not re-indexed
    # per replica) moves nucleotides by tenths
    np.testing.assert_allclose(two.center[:4].cpu().numpy(), one.center.cpu().numpy(), rtol=0, atol=0.05)
