"""Structural observables on the GPU (mythos_amd/csrc/observables.h) against the torch oracle, stand-alone and fused
into the energy launch.

 * propeller twist, rise, pitch angle, <l0> and the axis autocorrelation of thermal frames == oracle (1e-10 in fp64;
   fp32 inputs are read as they are and the arithmetic is fp64, so the same tolerance holds on the same inputs);
 * free and periodic displacement, oxDNA1 and oxDNA2 site geometry, skip_ends on / off, empty lists;
 * ``energy_fn.with_observables(...)``: the rows that come back with the energies (mythos_oxdna_energy_obs) are the rows of the
   stand-alone launch, bit for bit, and the observable calls that follow do not launch again;
 * DiffTRe end to end with the fused path gives the same loss and gradient as without it.
"""

import numpy as np
import pytest
import torch

from mythos_amd.energy import dna1, dna2
from mythos_amd.energy.base import Quaternion, RigidBody, space
from mythos_amd.input import defaults
from mythos_amd.observables import PersistenceLength, PitchAngle, PropellerTwist, Rise, get_duplex_quartets
from mythos_amd.observables import base as PB
from mythos_amd.optimization import objective as O
from mythos_amd.simulators.io import SimulatorTrajectory
from mythos_amd.utils import generators
from oracle import observables_oracle as OO
from tests import helpers as H

pytestmark = pytest.mark.gpu

KT = 296.15 * 0.1 / 300.0


def _thermal_duplex(bp, frames, model=2, seed=0, dtype=torch.float64, shift=None):
    """Ideal duplex + Gaussian noise per frame: every observable is away from its symmetric value."""
    top, c, q = generators.ideal_duplex(bp, model=model, seed=seed)
    rng = np.random.default_rng(seed)
    c = c[None] + 0.08 * rng.standard_normal((frames, *c.shape))
    q = q[None] + 0.06 * rng.standard_normal((frames, *q.shape))
    q /= np.linalg.norm(q, axis=-1, keepdims=True)
    if shift is not None:
        c = c + np.asarray(shift)
    dev = torch.device("cuda", 0)
    traj = SimulatorTrajectory(center=torch.as_tensor(c, dtype=dtype, device=dev), orientation=Quaternion(vec=torch.as_tensor(q, dtype=dtype, device=dev)),
                               temperature=torch.full((frames,), KT, dtype=torch.float64, device=dev))
    return top, traj


def _cpu64(traj):
    return SimulatorTrajectory(center=traj.center.double().cpu(), orientation=Quaternion(vec=traj.orientation.vec.double().cpu()))


@pytest.mark.parametrize(("model", "periodic", "dtype"), [(2, False, torch.float64), (2, True, torch.float64), (1, False, torch.float64),
                                                         (2, False, torch.float32), (3, False, torch.float64), (3, True, torch.float64)])
def test_observables_match_the_oracle(model, periodic, dtype):
    bp = 23
    shift = [19.0, 18.5, 17.0] if periodic else None  # the helix crosses the faces of a 20-unit box (unwrapped coordinates)
    top, traj = _thermal_duplex(bp, 7, model=model, seed=3, dtype=dtype, shift=shift)
    _, cfg = defaults.default_configs_for(H.model_dir(model))
    disp = space.periodic(20.0)[0] if periodic else space.free()[0]
    quartets = get_duplex_quartets(bp)
    pairs = np.stack([np.arange(bp), 2 * bp - 1 - np.arange(bp)], axis=1)[2:-2]
    ref = _cpu64(traj)
    checks = [
        (PropellerTwist(pairs)(traj), OO.PropellerTwist(pairs)(ref)),
        (Rise(quartets, disp, cfg["geometry"], model)(traj), OO.Rise(quartets, disp, cfg["geometry"], model)(ref)),
        (PitchAngle(quartets, disp, cfg["geometry"], model)(traj), OO.PitchAngle(quartets, disp, cfg["geometry"], model)(ref)),
    ]
    for skip in (True, False):
        got = PersistenceLength(quartets, disp, cfg["geometry"], model, skip_ends=skip).get_all_corrs_and_l0s(traj)
        want = OO.PersistenceLength(quartets, disp, cfg["geometry"], model, skip_ends=skip).get_all_corrs_and_l0s(ref)
        assert got[0].shape == (7, bp - 1 - (4 if skip else 0))
        checks += [(got[0], want[0]), (got[1], want[1])]
    for got, want in checks:
        assert got.dtype == torch.float64 and got.shape == want.shape
        assert want.abs().max() > 1e-3
        torch.testing.assert_close(got.cpu(), want, rtol=0, atol=1e-10)
    # fitted persistence length through the same host-side fit, with and without weights
    w = torch.rand(7, dtype=torch.float64)
    w /= w.sum()
    pl, ol = PersistenceLength(quartets, disp, cfg["geometry"], model, truncate=10), OO.PersistenceLength(quartets, disp, cfg["geometry"], model, truncate=10)
    assert abs(float(pl(traj)) - float(ol(ref))) <= 1e-8 * abs(float(ol(ref)))
    assert abs(float(pl(traj, weights=w.cuda())) - float(ol(ref, weights=w))) <= 1e-8 * abs(float(ol(ref, weights=w)))
    with pytest.raises(TypeError):
        pl(traj, weights=np.array([1.0, 2.0]))


def test_single_frame_empty_lists_and_bad_indices():
    top, traj = _thermal_duplex(8, 3)
    _, cfg = defaults.default_configs_for("dna2")
    disp = space.free()[0]
    one = RigidBody(center=traj.center[1], orientation=Quaternion(vec=traj.orientation.vec[1]))
    pairs = np.array([[2, 13], [3, 12]])
    assert torch.equal(PropellerTwist(pairs)(one), PropellerTwist(pairs)(traj)[1:2])
    s = PB.ObservableSet(16, 2, cfg["geometry"], None, None, None, True, torch.float64, traj.center.device)
    rows = s.eval(traj.center, traj.orientation.vec)
    assert s.width == 4 and rows.shape == (3, 4) and torch.equal(rows, torch.zeros_like(rows))
    # three quartets with skip_ends: nothing is left for the persistence-length partials, rise and pitch still are
    q3 = get_duplex_quartets(4)
    pl = PersistenceLength(q3, disp, cfg["geometry"], 2, skip_ends=True)
    top4, traj4 = _thermal_duplex(4, 2)
    c, l0 = pl.get_all_corrs_and_l0s(traj4)
    assert c.shape == (2, 0) and torch.equal(l0, torch.zeros_like(l0))
    assert Rise(q3, disp, cfg["geometry"], 2)(traj4).abs().min() > 1.0
    with pytest.raises(Exception, match="out of range"):
        PropellerTwist(np.array([[0, 99]]))(traj)
    zero = SimulatorTrajectory(center=traj.center[:0], orientation=Quaternion(vec=traj.orientation.vec[:0]))
    assert PropellerTwist(pairs)(zero).shape == (0,)


def test_rows_from_the_energy_launch_equal_the_stand_alone_rows():
    bp = 16
    top, traj = _thermal_duplex(bp, 12, seed=5)
    _, cfg = defaults.default_configs_for("dna2")
    disp = space.free()[0]
    quartets = get_duplex_quartets(bp)
    pairs = np.stack([np.arange(bp), 2 * bp - 1 - np.arange(bp)], axis=1)[1:-1]
    obs = [PropellerTwist(pairs), Rise(quartets, disp, cfg["geometry"]), PitchAngle(quartets, disp, cfg["geometry"]),
           PersistenceLength(quartets, disp, cfg["geometry"], truncate=6)]
    alone = [obs[0](traj), obs[1](traj), obs[2](traj), *obs[3].get_all_corrs_and_l0s(traj)]
    ef = dna2.create_default_energy_fn(topology=top, displacement_fn=disp)
    e_plain = ef.map(traj)
    PB._FUSED.clear()
    launches = {"n": 0}
    orig = PB.ObservableSet.eval

    def counting(self, c, q):
        launches["n"] += 1
        return orig(self, c, q)

    PB.ObservableSet.eval = counting
    try:
        e_fused = ef.with_observables(*obs).map(traj)
        assert torch.equal(e_fused, e_plain)
        fused = [obs[0](traj), obs[1](traj), obs[2](traj), *obs[3].get_all_corrs_and_l0s(traj)]
        assert launches["n"] == 0  # every row came out of the energy launch
        for a, b in zip(fused, alone):
            assert torch.equal(a, b)
        # an in-place change of the frames invalidates the remembered rows
        traj.center.add_(0.01)
        obs[0](traj)
        assert launches["n"] == 1
    finally:
        PB.ObservableSet.eval = orig
    # gradient modes carry the observables too (the DiffTRe evaluation asks for dU/dtheta)
    PB._FUSED.clear()
    par = {"eps_hb": torch.tensor(1.0678, dtype=torch.float64, requires_grad=True)}
    e = ef.with_observables(*obs).with_params(par).map(traj)
    (g,) = torch.autograd.grad(e.sum(), [par["eps_hb"]])
    assert torch.isfinite(g) and len(PB._FUSED) == 3  # rise and pitch describe the same set: one entry for both
    torch.testing.assert_close(obs[1](traj), OO.Rise(quartets, disp, cfg["geometry"])(_cpu64(traj)).cuda(), rtol=0, atol=1e-10)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_hip_observables_reproduce_the_references_own_known_answers(dtype):
    """mythos/observables/tests/test_rise.py:14-33 (14.753608), :42-83 (11.065206), test_propeller.py:39-80 (120) through
    the HIP kernels (stand-alone launch): site offsets 0, so a base site is a centre as in the reference's mocks."""
    dev = torch.device("cuda", 0)
    disp = space.free()[0]

    def traj(centers, quats, frames):
        c = torch.as_tensor(np.repeat(np.asarray(centers, dtype=np.float64)[None], frames, 0), dtype=dtype, device=dev)
        q = torch.as_tensor(np.repeat(np.asarray(quats, dtype=np.float64)[None], frames, 0), dtype=dtype, device=dev)
        return SimulatorTrajectory(center=c, orientation=Quaternion(vec=q))

    ident = [[1.0, 0, 0, 0]] * 4
    tol = 1e-7 if dtype == torch.float64 else 1e-6
    r1 = Rise(np.asarray(H.RISE_SINGLE["quartets"]), disp, H.ZERO_GEOMETRY)(traj(H.RISE_SINGLE["centers"], ident, 1))
    np.testing.assert_allclose(r1.cpu().numpy(), [H.RISE_SINGLE["expected"]], rtol=tol)
    r2 = Rise(np.asarray(H.RISE_CALL["quartets"]), disp, H.ZERO_GEOMETRY)(traj(H.RISE_CALL["centers"], ident, H.RISE_CALL["frames"]))
    np.testing.assert_allclose(r2.cpu().numpy(), [H.RISE_CALL["expected"]] * H.RISE_CALL["frames"], rtol=tol)
    pt = PropellerTwist(np.asarray(H.PROPELLER_CALL["pairs"]))(traj([[0.0, 0, 0]] * 4, H.PROPELLER_CALL["quats"], H.PROPELLER_CALL["frames"]))
    # (fp32 input: the quaternion components sqrt(1/2) are rounded, |a3|^2 = 1 - 1e-7, and the pair of EQUAL normals sits
    #  where acos has no derivative: its angle comes out as 0.02 degrees instead of 0)
    np.testing.assert_allclose(pt.cpu().numpy(), [H.PROPELLER_CALL["expected"]] * H.PROPELLER_CALL["frames"],
                               rtol=tol if dtype == torch.float64 else 2e-4)


def test_remembered_rows_never_serve_another_trajectory():
    """The rows remembered from an energy launch belong to ONE pair of tensors: a second trajectory of the same
    shape - allocated after the first was dropped, where the caching allocator would hand out the same address - and a
    buffer the integrator refills through raw pointers must both get rows of their own."""
    from mythos_amd.hip_system import LangevinIntegrator

    bp = 16
    top, traj = _thermal_duplex(bp, 6, seed=9)
    _, cfg = defaults.default_configs_for("dna2")
    disp = space.free()[0]
    pairs = np.stack([np.arange(bp), 2 * bp - 1 - np.arange(bp)], axis=1)[1:-1]
    ptw = PropellerTwist(pairs)
    ef = dna2.create_default_energy_fn(topology=top, displacement_fn=disp).with_observables(ptw)
    PB.clear_fused()
    ef.map(traj)
    first = ptw(traj).clone()
    shape_c, shape_q = traj.center.shape, traj.orientation.vec.shape
    c2 = traj.center.flip(0).contiguous() + 0.05
    q2 = traj.orientation.vec.flip(0).contiguous()
    del traj
    torch.cuda.empty_cache()
    other = SimulatorTrajectory(center=torch.empty(shape_c, dtype=c2.dtype, device=c2.device).copy_(c2),
                                orientation=Quaternion(vec=torch.empty(shape_q, dtype=q2.dtype, device=q2.device).copy_(q2)))
    got = ptw(other)
    PB.clear_fused()
    want = ptw(other)  # nothing remembered: the stand-alone launch
    assert torch.equal(got, want) and not torch.equal(got, first)
    # state tensors the integrator updates in place carry a new version afterwards
    sim, _ = defaults.default_configs_for("dna2")
    from mythos_amd import _lib as L
    from mythos_amd.energy import flat_params as fp
    from mythos_amd.hip_system import OxdnaSystem
    s = OxdnaSystem(2, top.seq, top.is_end, top.bonded_neighbors, dtype=torch.float32, device=other.center.device)
    s.set_params(fp.pack_flat(fp.derive_flat(2, cfg, kt=sim["kT"], half_charged_ends=True), L.param_names()))
    integ = LangevinIntegrator(s, dt=sim["dt"], kT=sim["kT"], gamma_t=sim["kT"] / 2.5, gamma_r=sim["kT"] / 7.5, seed=3)
    integ.set_neighbor_policy(3.25, 0.6, 25)
    c = other.center[0].to(torch.float32).contiguous()
    q = other.orientation.vec[0].to(torch.float32).contiguous()
    p, ang = integ.init_momenta()
    v0 = (c._version, q._version)
    integ.run(c, q, p, ang, 5)
    assert c._version > v0[0] and q._version > v0[1]


def test_difftre_loss_and_gradient_are_unchanged_by_fusing_the_observable():
    top, traj, _, _ = H.load_golden(1, "simple-helix")
    dev = torch.device("cuda", 0)
    disp, _ = space.periodic(traj.box_size)
    st = SimulatorTrajectory(center=torch.as_tensor(traj.center[::4], device=dev), orientation=Quaternion(vec=torch.as_tensor(traj.quaternions[::4], device=dev)),
                             temperature=torch.full((25,), KT, dtype=torch.float64, device=dev))
    n = top.n_nucleotides
    ptwist = PropellerTwist(np.stack([np.arange(n // 2), n - 1 - np.arange(n // 2)], axis=1)[1:-1])
    ef = dna1.create_default_energy_fn(topology=top, displacement_fn=disp)
    opt = {"eps_stack_base": 1.3448, "theta0_hb_4": float(np.pi)}
    ref_e = ef.with_params({"eps_stack_base": 1.30}).map(st).detach()

    def loss_fn(ref_states, weights, energy_fn, opt_params, observables):  # noqa: ARG001
        m = (weights * ptwist(ref_states).to(weights.dtype)).sum()
        return (m - 21.7) ** 2, (("propeller_twist", m.detach()), {})

    PB._FUSED.clear()
    (l0, (n0, _, _)), g0 = O.compute_loss_and_grad(opt, ef, 1.0 / KT, loss_fn, st, ref_e, [st])
    assert len(PB._FUSED) == 0
    (l1, (n1, _, _)), g1 = O.compute_loss_and_grad(opt, ef.with_observables(ptwist), 1.0 / KT, loss_fn, st, ref_e, [st])
    assert len(PB._FUSED) == 1
    assert float(l0) == float(l1) and float(n0) == float(n1)
    for k in opt:
        assert float(g0[k]) == float(g1[k])


def test_observable_loss_wrapper_on_a_hip_observable():
    """mythos_amd.losses.observable_wrappers.ObservableLossFn (the reference's class) around the HIP propeller twist:
    the weighted mean of the kernel's per-state values, the squared error to a target, and its gradient to the weights."""
    from mythos_amd.losses import observable_wrappers as ow

    top, traj = _thermal_duplex(12, 7)
    half = top.n_nucleotides // 2
    pairs = np.stack([np.arange(half), top.n_nucleotides - 1 - np.arange(half)], axis=1)[1:-1]
    pt = PropellerTwist(pairs)
    per_state = pt(traj).double()
    w = torch.full((7,), 1.0 / 7, dtype=torch.float64, device=per_state.device, requires_grad=True)
    loss, mean = ow.ObservableLossFn(observable=pt, loss_fn=ow.SquaredError(), return_observable=True)(traj, torch.tensor(21.7, dtype=torch.float64, device=per_state.device), w)
    assert float(mean.detach()) == pytest.approx(float(per_state.mean()), rel=1e-12) and float(loss.detach()) == pytest.approx((21.7 - float(per_state.mean())) ** 2, rel=1e-12)
    loss.backward()
    assert torch.allclose(w.grad, -2.0 * (21.7 - per_state.mean()) * per_state, rtol=1e-12)
