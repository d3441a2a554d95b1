"""GPU tests through the reference-shaped Python surface.  They read like the reference's
mythos/energy/dna{1,2}/tests/test_integration.py: build each term with default parameters and a
periodic box of 20, ``energy_fn.map(states)``, divide by N, round to 6 decimals, compare with the
``split_energy.dat`` column."""

import numpy as np
import pytest
import torch

from mythos_amd.energy import dna1, dna2
from mythos_amd.energy.base import Quaternion, RigidBody, space
from mythos_amd.optimization import objective
from tests import helpers as H

pytestmark = pytest.mark.gpu

KT = 296.15 * 0.1 / 300.0


def _states(traj, dtype=torch.float64):
    dev = torch.device("cuda", 0)
    return RigidBody(
        center=torch.as_tensor(traj.center, dtype=dtype, device=dev),
        orientation=Quaternion(vec=torch.as_tensor(traj.quaternions, dtype=dtype, device=dev)),
    )


def _setup(model, name):
    top, traj, split, energy = H.load_golden(model, name)
    mod = dna1 if model == 1 else dna2
    displacement_fn, _ = space.periodic(20.0)
    return top, traj, split, energy, mod, mod.default_configs()[1], mod.default_transform_fn(), displacement_fn


TERM_CASES = [
    (2, "simple-helix", "Fene", "FeneConfiguration", "fene", {}),
    (2, "simple-helix", "BondedExcludedVolume", "BondedExcludedVolumeConfiguration", "bonded_excluded_volume", {}),
    (2, "simple-helix", "Stacking", "StackingConfiguration", "stacking", {"kt": KT}),
    (2, "simple-helix", "UnbondedExcludedVolume", "UnbondedExcludedVolumeConfiguration", "unbonded_excluded_volume", {}),
    (2, "simple-helix", "HydrogenBonding", "HydrogenBondingConfiguration", "hydrogen_bonding", {}),
    (2, "simple-helix", "CrossStacking", "CrossStackingConfiguration", "cross_stacking", {}),
    (2, "simple-coax", "CoaxialStacking", "CoaxialStackingConfiguration", "coaxial_stacking", {}),
    (2, "simple-helix", "Debye", "DebyeConfiguration", "debye", {"kt": KT, "salt_conc": 0.5, "half_charged_ends": False}),
    (1, "simple-helix", "Stacking", "StackingConfiguration", "stacking", {"kt": KT}),
    (1, "simple-coax", "CoaxialStacking", "CoaxialStackingConfiguration", "coaxial_stacking", {}),
    (1, "simple-helix", "HydrogenBonding", "HydrogenBondingConfiguration", "hydrogen_bonding", {}),
]


@pytest.mark.parametrize(("model", "name", "cls", "cfg_cls", "section", "extra"), TERM_CASES)
def test_single_term_matches_split_energy(model, name, cls, cfg_cls, section, extra):
    top, traj, split, _, mod, default_params, transform_fn, displacement_fn = _setup(model, name)
    energy_config = getattr(mod, cfg_cls)(**{**default_params[section], **extra})
    energy_fn = getattr(mod, cls)(
        displacement_fn=displacement_fn, transform_fn=transform_fn, topology=top, params=energy_config.init_params()
    )
    energy = energy_fn.map(_states(traj)).cpu().numpy()
    energy = np.around(energy / top.n_nucleotides, 6)
    np.testing.assert_allclose(energy, split[:, H.SPLIT_COLUMNS.index(section)], atol=H.TERM_ATOL[section])


@pytest.mark.parametrize(("model", "name", "hce"), [(1, "simple-helix", False), (2, "simple-helix", False), (2, "simple-helix-half-charged-ends", True)])
def test_total_energy_default_fn(model, name, hce):
    top, traj, _, energy, mod, _, _, displacement_fn = _setup(model, name)
    energy_fn = mod.create_default_energy_fn(top, displacement_fn)
    if model == 2:
        energy_fn = energy_fn.with_params(half_charged_ends=hce)
    states = _states(traj)
    e = energy_fn.map(states).cpu().numpy() / top.n_nucleotides
    np.testing.assert_allclose(e, energy, atol=1e-3 if model == 2 else 1e-4)
    terms = energy_fn.compute_terms(states)
    assert terms.shape == (100, len(energy_fn.energy_fns))
    np.testing.assert_allclose(terms.sum(1).cpu().numpy() / top.n_nucleotides, e, atol=1e-12)
    single = energy_fn(states[3])
    assert single.dim() == 0 and abs(single.item() / top.n_nucleotides - e[3]) < 1e-12


def test_kt_is_a_shared_parameter():
    """with_params(kt=...) reaches stacking and Debye (dna2/tests/test_integration.py:366-369)."""
    top, traj, _, _, mod, _, _, displacement_fn = _setup(2, "simple-helix")
    ef = mod.create_default_energy_fn(top, displacement_fn)
    st = _states(traj)[:2]
    a = ef.compute_terms(st)
    b = ef.with_params(kt=0.12).compute_terms(st)
    changed = (a - b).abs().max(0).values.cpu().numpy() > 1e-9
    assert changed.tolist() == [False, False, True, False, False, False, False, True]


def test_weights_and_term_selection():
    top, traj, _, _, mod, _, _, displacement_fn = _setup(2, "simple-helix")
    ef = mod.create_default_energy_fn(top, displacement_fn)
    st = _states(traj)[:3]
    terms = ef.compute_terms(st)
    w = torch.tensor([1.0, 0.5, 2.0, 1.0, 0.0, 1.0, 1.0, 3.0], dtype=torch.float64)
    efw = ef.replace(weights=w)
    torch.testing.assert_close(efw(st).cpu(), (terms.cpu() * w).sum(1), rtol=1e-12, atol=1e-12)
    no_dh = ef.without_terms("Debye")
    torch.testing.assert_close(no_dh(st).cpu(), terms[:, :7].sum(1).cpu(), rtol=1e-12, atol=1e-12)


def test_autograd_positions_orientations_and_parameters():
    """torch.autograd through the HIP op == oracle autograd (stand-in for jax.grad), incl. term weights."""
    from oracle import oxdna_oracle as orc

    top, traj, _, _, mod, _, _, displacement_fn = _setup(2, "simple-helix")
    w = torch.tensor([1.0, 0.5, 2.0, 1.0, 0.7, 1.0, 1.0, 3.0], dtype=torch.float64)
    ef = mod.create_default_energy_fn(top, displacement_fn).replace(weights=w)
    st = _states(traj)[7]
    c = st.center.clone().requires_grad_(True)
    q = st.orientation.vec.clone().requires_grad_(True)
    a_stack = torch.tensor(6.0, dtype=torch.float64, requires_grad=True)
    eps_hb = torch.tensor(1.0678, dtype=torch.float64, requires_grad=True)
    u = ef.with_params(a_stack=a_stack, eps_hb=eps_hb)(RigidBody(c, Quaternion(q)))
    gc, gq, ga, ge = torch.autograd.grad(u, (c, q, a_stack, eps_hb))
    # oracle with the same weights
    a2 = torch.tensor(6.0, dtype=torch.float64, requires_grad=True)
    e2 = torch.tensor(1.0678, dtype=torch.float64, requires_grad=True)
    P = H.oracle_params(2, overrides={"stacking": {"a_stack": a2}, "hydrogen_bonding": {"eps_hb": e2}}, half_charged_ends=True)
    seq, is_end, b, un = H.topo_tensors(top)
    c2 = torch.as_tensor(traj.center[7]).requires_grad_(True)
    q2 = torch.as_tensor(traj.quaternions[7]).requires_grad_(True)
    u2 = (orc.energy_terms(2, P, c2, q2, seq, is_end, b, un, box=traj.box_size) * w).sum()
    rc, rq, ra, re = torch.autograd.grad(u2, (c2, q2, a2, e2))
    assert abs(u.item() - u2.item()) < 1e-9 * abs(u2.item())
    assert (gc.cpu() - rc).abs().max() < 1e-7 * rc.abs().max()
    assert (gq.cpu() - rq).abs().max() < 1e-7 * rq.abs().max()
    assert abs(ga.item() - ra.item()) < 1e-7 * abs(ra.item())
    assert abs(ge.item() - re.item()) < 1e-7 * abs(re.item())


def test_difftre_loss_and_grad_matches_oracle():
    """compute_loss_and_grad (objective.py:235) on 20 golden frames vs the same math on oracle energies."""
    from oracle import oxdna_oracle as orc

    top, traj, _, _, mod, _, _, displacement_fn = _setup(2, "simple-helix")
    ef = mod.create_default_energy_fn(top, displacement_fn)
    frames = list(range(0, 100, 5))
    st = _states(traj)[frames]
    ref_e = ef.map(st).detach()
    target = torch.tensor(0.2, dtype=torch.float64)

    def observable(states):  # mean COM distance of the first base pair, per frame
        return (states.center[:, 0] - states.center[:, 15]).norm(dim=1).to(torch.float64)

    def loss_fn(states, weights, energy_fn, opt_params, observables):  # noqa: ARG001
        measured = (weights * observable(states)).sum()
        return (measured - target.to(measured.device)) ** 2, (measured, None)

    opt = {"eps_hb": 1.10, "a_stack": 6.2, "k_cross": 47.0}
    (loss, (neff, measured, new_e)), grads = objective.compute_loss_and_grad(opt, ef, 1.0 / KT, loss_fn, st, ref_e, [])
    assert 0.0 < float(neff) <= 1.0 and new_e.shape == (len(frames),)
    # oracle
    leaves = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in opt.items()}
    P = H.oracle_params(2, half_charged_ends=True, overrides={
        "hydrogen_bonding": {"eps_hb": leaves["eps_hb"]}, "stacking": {"a_stack": leaves["a_stack"]}, "cross_stacking": {"k_cross": leaves["k_cross"]}})
    seq, is_end, b, un = H.topo_tensors(top)
    es = torch.stack([
        orc.energy(2, P, torch.as_tensor(traj.center[f]), torch.as_tensor(traj.quaternions[f]), seq, is_end, b, un, box=traj.box_size)
        for f in frames
    ])
    w_o, _ = objective.compute_weights_and_neff(1.0 / KT, es, ref_e.cpu())
    obs = torch.as_tensor(np.linalg.norm(traj.center[frames, 0] - traj.center[frames, 15], axis=1))
    loss_o = ((w_o * obs).sum() - target) ** 2
    g_o = torch.autograd.grad(loss_o, list(leaves.values()))
    assert abs(loss.item() - loss_o.item()) < 1e-8 * max(1.0, abs(loss_o.item()))
    for k, g in zip(leaves, g_o):
        assert abs(grads[k].item() - g.item()) <= 1e-5 * max(abs(g.item()), 1e-8), (k, grads[k].item(), g.item())


def test_simulator_run_surface():
    from mythos_amd.simulators.hip_md import HipMDSimulator, StaticSimulatorParams, nvt_langevin
    from mythos_amd.simulators.neighbors import NoNeighborList, VerletNeighborList

    top, traj, _, _, mod, _, _, _ = _setup(2, "simple-helix")
    disp, shift = space.free()
    ef = mod.create_default_energy_fn(top, disp)
    params = StaticSimulatorParams(
        seq=top.seq, mass=(1.0, (1.0, 1.0, 1.0)),
        gamma=(KT / 2.5, KT / 7.5), bonded_neighbors=top.bonded_neighbors, checkpoint_every=0, dt=5e-3, kT=KT,
    )
    init = _states(traj, torch.float32)[0]
    for nb in (NoNeighborList(unbonded_nbrs=top.unbonded_neighbors), VerletNeighborList(3.25, 0.6, 20)):
        sim = HipMDSimulator(energy_fn=ef, simulator_params=params, space=(disp, shift), simulator_init=nvt_langevin, neighbors=nb,
                             trace_energy=True)
        assert sim.exposes()[0].startswith("trajectory.HipMDSimulator.")
        out = sim.run({"eps_hb": 1.05}, init, 50, key=3)
        tr = out.observables[0]
        assert tr.center.shape == (50, 16, 3) and tr.orientation.vec.shape == (50, 16, 4)
        assert tr.temperature.shape == (50,) and torch.allclose(tr.temperature, torch.full((50,), KT, dtype=torch.float64, device=tr.temperature.device))
        assert torch.isfinite(tr.center).all()
        assert tr.slice(slice(10, 20)).length() == 10
        # the stored frames are consistent with the energy function (same parameters)
        e_traj = ef.with_params(eps_hb=1.05).map(RigidBody(tr.center, tr.orientation))
        torch.testing.assert_close(e_traj.cpu(), tr.metadata["energy_terms"].sum(1).cpu(), rtol=2e-4, atol=1e-3)


def test_integrator_plug_has_the_reference_shape_and_equals_the_one_call_loop():
    """`init_fn, step_fn = simulator_init(energy_fn, shift_fn, dt=, kT=, gamma=)`; `state = init_fn(key, R, mass=)`;
    `state = step_fn(state)`; the loop reads `state.position` (mythos/simulators/jax_md/jaxmd.py:73-92, utils.py:19-28).
    Stepped that way, the states equal the rows of HipMDSimulator.run - the same loop in one call - bit for bit; a state is a
    value: stepping an older one again reproduces its successor."""
    from mythos_amd.simulators.hip_md import HipMDSimulator, NVTLangevinState, StaticSimulatorParams, nvt_langevin
    from mythos_amd.simulators.neighbors import NoNeighborList

    top, traj, _, _, mod, _, _, _ = _setup(2, "simple-helix")
    disp, shift = space.free()
    ef = mod.create_default_energy_fn(top, disp)
    gamma = RigidBody(center=KT / 2.5, orientation=KT / 7.5)
    mass = RigidBody(center=1.0, orientation=(1.0, 1.0, 1.0))
    init = _states(traj, torch.float64)[0]
    for dtype in (torch.float64, torch.float32):
        plug = nvt_langevin(ef, shift, dt=5e-3, kT=KT, gamma=gamma, dtype=dtype)
        init_fn, step_fn = plug
        state = init_fn(11, init, mass=mass)
        assert isinstance(state, NVTLangevinState) and state.step == 0 and state.mass is mass
        torch.testing.assert_close(state.position.center.double().cpu(), init.center.cpu().double(), rtol=0, atol=1e-6 if dtype == torch.float32 else 0)
        states = [state]
        for _ in range(6):
            states.append(step_fn(states[-1]))
        assert [s.step for s in states] == list(range(7))
        sim = HipMDSimulator(energy_fn=ef, simulator_params=StaticSimulatorParams(seq=top.seq, mass=mass, gamma=gamma, bonded_neighbors=top.bonded_neighbors,
                                                                                  checkpoint_every=0, dt=5e-3, kT=KT),
                             space=(disp, shift), simulator_init=nvt_langevin, neighbors=NoNeighborList(unbonded_nbrs=top.unbonded_neighbors),
                             save_every=1, dtype=dtype)
        rows = sim.run({}, init, 6, key=11).observables[0]
        for k in range(6):
            assert torch.equal(states[k + 1].position.center, rows.center[k]), (dtype, k)
            assert torch.equal(states[k + 1].position.orientation.vec, rows.orientation.vec[k]), (dtype, k)
        # a state is a value: an older one can be stepped again (reloaded, its closing half kick and the next step's opening one
        # are then two roundings where the resident frames have one: equal to rounding, not to the bit)
        tol = dict(rtol=0, atol=1e-12) if dtype == torch.float64 else dict(rtol=0, atol=2e-5)
        again = step_fn(states[2])
        assert again.step == 3
        torch.testing.assert_close(again.position.center, states[3].position.center, **tol)
        torch.testing.assert_close(again.position.orientation.vec, states[3].position.orientation.vec, **tol)
        torch.testing.assert_close(again.momentum.center, states[3].momentum.center, **tol)
        torch.testing.assert_close(again.momentum.orientation, states[3].momentum.orientation, **tol)
        onward = step_fn(again)
        torch.testing.assert_close(onward.position.center, states[4].position.center, **tol)
        plug.close()
    with pytest.raises(NotImplementedError, match="energy function of this package"):
        nvt_langevin(lambda R, **kw: 0.0, shift, dt=5e-3, kT=KT, gamma=gamma)


def test_simulator_batches_independent_replicas_in_one_launch():
    """n_replicas copies of the system advance as one launch per step, far apart in free space.  Cold and frictionless
    (kT -> 0, gamma = 0) the copies follow the same deterministic trajectory as a single run; thermal, they stay where
    they were put relative to their own origin, decorrelate from each other, and hand back (R, n, .) states."""
    import dataclasses as dc

    from mythos_amd.simulators.hip_md import HipMDSimulator, StaticSimulatorParams, nvt_langevin
    from mythos_amd.simulators.neighbors import NoNeighborList, VerletNeighborList

    top, traj, _, _, mod, _, _, _ = _setup(2, "simple-helix")
    disp, shift = space.free()
    ef = mod.create_default_energy_fn(top, disp)
    init = _states(traj, torch.float32)[0]
    cold = StaticSimulatorParams(seq=top.seq, mass=(1.0, (1.0, 1.0, 1.0)), gamma=(0.0, 0.0),
                                 bonded_neighbors=top.bonded_neighbors, checkpoint_every=0, dt=5e-3, kT=1e-14)
    for nb in (NoNeighborList(unbonded_nbrs=top.unbonded_neighbors), VerletNeighborList(3.25, 0.6, 20)):
        one = HipMDSimulator(energy_fn=ef, simulator_params=cold, space=(disp, shift), simulator_init=nvt_langevin, neighbors=nb,
                             save_every=10)
        ref = one.run({}, init, 60, key=5).observables[0].center  # (6, 16, 3)
        many = dc.replace(one, n_replicas=5)
        out = many.run({}, init, 60, key=5)
        got = out.observables[0].center
        assert got.shape == (5 * 6, 16, 3) and out.observables[0].orientation.vec.shape == (5 * 6, 16, 4)
        for r in range(5):
            torch.testing.assert_close(got[6 * r:6 * (r + 1)], ref, rtol=0, atol=2e-4)
        assert out.state["final_state"].center.shape == (5, 16, 3)
    warm = dc.replace(cold, kT=KT, gamma=(KT / 2.5, KT / 7.5))
    sim = HipMDSimulator(energy_fn=ef, simulator_params=warm, space=(disp, shift), simulator_init=nvt_langevin,
                         neighbors=VerletNeighborList(3.25, 0.6, 20), save_every=50, n_replicas=8)
    out = sim.run({}, init, 400, key=9)
    c = out.observables[0].center.reshape(8, 8, 16, 3)
    assert torch.isfinite(c).all()
    com = c.mean(dim=2)                                      # replicas sit at their own origin again
    assert (com - init.center.mean(0).to(com)).norm(dim=-1).max() < 3.0
    last = c[:, -1]
    assert (last[0] - last[1]).abs().max() > 1e-3            # different noise per replica
    e = ef.map(RigidBody(out.observables[0].center, out.observables[0].orientation))
    assert torch.isfinite(e).all() and e.shape == (64,)
    # the next run continues all replicas from where they are
    again = sim.run({}, **{k: out.state[k] for k in ("init_state", "key")}, n_steps=50)
    assert again.observables[0].center.shape == (8, 16, 3)
    # replicas need free space
    box_disp, box_shift = space.periodic(traj.box_size)
    with pytest.raises(ValueError, match="free space"):
        dc.replace(sim, energy_fn=mod.create_default_energy_fn(top, box_disp), space=(box_disp, box_shift)).run({}, init, 5, key=1)


def test_sequence_dependent_energy_function_from_the_references_file():
    """The drop-in route to sequence-dependent weights, as a user of the reference takes it: `read_ss_weights(file)` into
    `with_params` of the default oxDNA2 energy function (mythos/input/sequence_dependence.py:12-51), evaluated with `map` over
    oxDNA's own sequence-dependent run (tests/golden/regr/simple-helix-oxdna2-ss): oxDNA's total per nucleotide, 25 frames."""
    from mythos_amd.energy import dna2
    from mythos_amd.input.sequence_dependence import read_ss_weights
    from tests import helpers as H

    top, traj, split, _ = H.load_regr("simple-helix-oxdna2-ss")
    disp, _ = space.periodic(traj.box_size)
    w = read_ss_weights(H.GOLDEN / "seq-specific" / "seq_oxdna2.txt")
    ef = dna2.create_default_energy_fn(top, disp).with_params(half_charged_ends=False, **w)
    body = RigidBody(center=torch.as_tensor(traj.center, device="cuda"), orientation=Quaternion(vec=torch.as_tensor(traj.quaternions, device="cuda")))
    e = ef.map(body).cpu().numpy() / top.n_nucleotides
    assert e.shape == (25,) and np.abs(e - split.sum(1)).max() <= 3e-5, np.abs(e - split.sum(1)).max()
    plain = dna2.create_default_energy_fn(top, disp).with_params(half_charged_ends=False).map(body).cpu().numpy() / top.n_nucleotides
    assert np.abs(plain - split.sum(1)).max() > 1e-3  # (the average-sequence model is a different function)
