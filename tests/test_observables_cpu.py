"""The oracle of the structural observables (oracle/observables_oracle.py, torch) on geometries whose answers are known
in closed form: an ideal B-duplex from the generator has a fixed twist and rise per base pair and parallel base
normals; a discrete worm-like chain has a known persistence length.  The product path is HIP
(mythos_amd/csrc/observables.h) and is held to this oracle in tests/test_gpu_observables.py."""

import dataclasses as dc
import math

import numpy as np
import pytest
import torch

from mythos_amd.energy.base import Quaternion, space
from mythos_amd.input import defaults
from oracle import observables_oracle as OB
from oracle.observables_oracle import PitchAngle, PropellerTwist, Rise, compute_pitch, get_duplex_quartets
from mythos_amd.utils import generators


@dc.dataclass
class Traj:
    center: torch.Tensor
    orientation: Quaternion


def _duplex(bp, frames=3):
    top, c, q = generators.ideal_duplex(bp, model=2, seed=1)
    c = torch.as_tensor(np.repeat(c[None], frames, 0))
    q = torch.as_tensor(np.repeat(q[None], frames, 0))
    return top, Traj(center=c, orientation=Quaternion(vec=q))


def test_quartets_enumerate_adjacent_base_pairs():
    q = get_duplex_quartets(4)
    assert q.shape == (3, 2, 2)
    assert q[0].tolist() == [[0, 7], [1, 6]] and q[2].tolist() == [[2, 5], [3, 4]]


def test_ideal_duplex_has_the_generators_twist_rise_and_flat_base_pairs():
    bp = 21
    top, traj = _duplex(bp)
    _, cfg = defaults.default_configs_for("dna2")
    disp, _ = space.free()
    quartets = get_duplex_quartets(bp)
    angle = PitchAngle(quartets, disp, cfg["geometry"])(traj)
    rise = Rise(quartets, disp, cfg["geometry"])(traj)
    assert angle.shape == (3,) and rise.shape == (3,)
    # the generator builds 10.5 bp per turn: twist 2 pi / 10.5 per step; the reference's convention pitch = pi / <angle>
    assert torch.allclose(angle, torch.full((3,), 2 * math.pi / 10.5, dtype=angle.dtype), atol=1e-9)
    assert abs(float(compute_pitch(angle[0])) - 10.5 / 2) < 1e-8
    # the generator stacks base pairs generators.BASE_BASE length units apart along the axis (3.32 Angstrom)
    assert torch.allclose(rise, torch.full((3,), generators.BASE_BASE * OB.ANGSTROMS_PER_OXDNA_LENGTH, dtype=rise.dtype), atol=1e-9)
    # paired bases of the ideal helix are coplanar with antiparallel normals: 180 - acos(-1) = 0 degrees
    pairs = np.stack([np.arange(bp), 2 * bp - 1 - np.arange(bp)], axis=1)
    assert torch.allclose(PropellerTwist(pairs)(traj), torch.zeros(3, dtype=traj.center.dtype), atol=1e-5)


def test_sites_follow_the_geometry_section_and_periodic_displacement():
    top, traj = _duplex(4, frames=1)
    _, cfg2 = defaults.default_configs_for("dna2")
    _, cfg1 = defaults.default_configs_for("dna1")
    base2, back2, stack2 = OB.nucleotide_sites(traj, cfg2["geometry"], model=2)
    base1, back1, _ = OB.nucleotide_sites(traj, cfg1["geometry"], model=1)
    a1, a2, a3 = OB.axes_from_quaternion(traj.orientation.vec)
    assert torch.allclose(base2 - traj.center, 0.4 * a1) and torch.allclose(stack2 - traj.center, 0.34 * a1)
    assert torch.allclose(back2 - traj.center, -0.34 * a1 + 0.3408 * a2) and torch.allclose(back1 - traj.center, -0.4 * a1)
    assert torch.allclose(torch.cross(a3, a1, dim=-1), a2, atol=1e-12) and torch.allclose(base1, base2)
    # a helix translated across a periodic face (unwrapped coordinates, as the MD kernel keeps them) keeps its rise
    # with the periodic displacement; midpoints are plain averages of site coordinates, as in the reference
    disp, _ = space.periodic(50.0)
    shifted = Traj(center=traj.center + torch.tensor([49.0, 49.5, 48.0]), orientation=traj.orientation)
    r0 = Rise(get_duplex_quartets(4), space.free()[0], cfg2["geometry"])(traj)
    r1 = Rise(get_duplex_quartets(4), disp, cfg2["geometry"])(shifted)
    assert torch.allclose(r0, r1, atol=1e-9)


def test_vector_autocorrelation_and_fit_follow_their_definitions():
    from oracle.observables_oracle import persistence_length_fit, vector_autocorrelate

    rng = np.random.default_rng(3)
    v = rng.normal(size=(2, 9, 3))
    got = vector_autocorrelate(torch.as_tensor(v)).numpy()
    for f in range(2):
        for d in range(9):
            want = np.mean([v[f, i] @ v[f, i + d] for i in range(9 - d)])  # persistence_length.py:47-75
            assert abs(got[f, d] - want) < 1e-12
    lp, off = persistence_length_fit(torch.exp(-torch.arange(30, dtype=torch.float64) * 0.4 / 50.0) * 0.9, 0.4)
    assert abs(float(lp) - 50.0) < 1e-8 and abs(float(off) - math.log(0.9)) < 1e-10


def test_persistence_length_of_a_discrete_wormlike_chain():
    """Base-pair midpoints on a chain whose successive tangents bend by a random small angle:
    <t_k . t_(k+d)> = <cos theta>^d, so Lp = -l0 / ln <cos theta>."""
    from oracle.observables_oracle import PersistenceLength

    n, frames, l0, sigma = 40, 1500, 0.4, 0.12
    rng = np.random.default_rng(11)
    t = np.zeros((frames, n, 3))
    t[:, 0] = [0.0, 0.0, 1.0]
    for k in range(1, n):
        kick = rng.normal(scale=sigma, size=(frames, 3))
        kick -= (kick * t[:, k - 1]).sum(-1, keepdims=True) * t[:, k - 1]
        new = t[:, k - 1] + kick
        t[:, k] = new / np.linalg.norm(new, axis=-1, keepdims=True)
    pts = np.cumsum(l0 * t, axis=1)  # midpoint k
    centre = np.concatenate([pts, pts[:, ::-1]], axis=1)  # base pair k = (k, 2n-1-k), both bases on the midpoint
    quat = np.zeros((frames, 2 * n, 4))
    quat[..., 0] = 1.0
    traj = Traj(center=torch.as_tensor(centre), orientation=Quaternion(vec=torch.as_tensor(quat)))
    _, cfg = defaults.default_configs_for("dna2")
    disp, _ = space.free()
    obs = PersistenceLength(get_duplex_quartets(n), disp, cfg["geometry"], model=2, truncate=20, skip_ends=True)
    corrs, l0s = obs.get_all_corrs_and_l0s(traj)
    assert corrs.shape == (frames, n - 1 - 4) and torch.allclose(l0s, torch.full_like(l0s, l0), atol=1e-12)
    tangents = t[:, 1:]  # quartet k joins midpoints k, k+1
    cos_mean = np.mean((tangents[:, :-1] * tangents[:, 1:]).sum(-1))
    expect = -l0 / math.log(cos_mean)
    lp, offset = obs.lp_fit(traj)
    assert abs(float(lp) - expect) / expect < 0.08 and abs(float(offset)) < 0.02
    # uniform weights reproduce the unweighted fit; a one-hot weight reproduces the single-frame fit
    w = torch.full((frames,), 1.0 / frames, dtype=torch.float64)
    assert abs(float(obs(traj, weights=w)) - float(lp)) < 1e-9 * float(lp)
    one = torch.zeros(frames, dtype=torch.float64)
    one[0] = 1.0
    first = Traj(center=traj.center[:1], orientation=Quaternion(vec=traj.orientation.vec[:1]))
    assert abs(float(obs(traj, weights=one)) - float(obs(first))) < 1e-9 * abs(float(obs(first)))
    import pytest

    with pytest.raises(TypeError):
        obs(traj, weights=np.array([1.0, 2.0]))
    # skip_ends=False keeps all n - 1 quartets
    full = PersistenceLength(get_duplex_quartets(n), disp, cfg["geometry"], skip_ends=False)
    assert full.get_all_corrs_and_l0s(traj)[0].shape == (frames, n - 1)


def test_product_observables_need_a_gpu_and_share_the_host_side_pieces():
    """The product classes describe what to measure; the numbers come from the HIP library, so a CPU trajectory is
    refused loudly.  The quartet enumeration and the least-squares fit are host logic shared with the oracle."""
    import pytest

    from mythos_amd import _lib
    from mythos_amd import observables as PO

    assert torch.equal(PO.get_duplex_quartets(5), get_duplex_quartets(5))
    corr = torch.exp(-torch.arange(12, dtype=torch.float64) * 0.39 / 120.0)
    a, b = PO.persistence_length_fit(corr, 0.39), OB.persistence_length_fit(corr, 0.39)
    assert abs(float(a[0]) - float(b[0])) < 1e-9 and abs(float(a[0]) - 120.0) < 1e-6
    top, traj = _duplex(6, frames=2)
    with pytest.raises(_lib.MythosHipError, match="GPU"):
        PO.PropellerTwist(np.array([[0, 11]]))(traj)


def _known_answer_traj(centers, quats, frames):
    c = torch.as_tensor(np.repeat(np.asarray(centers, dtype=np.float64)[None], frames, 0))
    q = torch.as_tensor(np.repeat(np.asarray(quats, dtype=np.float64)[None], frames, 0))
    return Traj(center=c, orientation=Quaternion(vec=q))


def test_oracle_reproduces_the_references_own_known_answers():
    """mythos/observables/tests/test_rise.py:14-33 (14.753608), :42-83 (11.065206), test_propeller.py:39-80 (120): the
    values the reference's tests pin, through the oracle with site offsets 0 (tests/helpers.py says how the mocks map)."""
    from tests import helpers as H

    disp, _ = space.free()
    ident = [[1.0, 0, 0, 0]] * 4
    r1 = Rise(torch.as_tensor(H.RISE_SINGLE["quartets"]), disp, H.ZERO_GEOMETRY)(_known_answer_traj(H.RISE_SINGLE["centers"], ident, 1))
    np.testing.assert_allclose(r1.numpy(), [H.RISE_SINGLE["expected"]], rtol=1e-7)
    r2 = Rise(torch.as_tensor(H.RISE_CALL["quartets"]), disp, H.ZERO_GEOMETRY)(_known_answer_traj(H.RISE_CALL["centers"], ident, H.RISE_CALL["frames"]))
    np.testing.assert_allclose(r2.numpy(), [H.RISE_CALL["expected"]] * H.RISE_CALL["frames"], rtol=1e-7)
    pt = PropellerTwist(torch.as_tensor(H.PROPELLER_CALL["pairs"]))(
        _known_answer_traj([[0.0, 0, 0]] * 4, H.PROPELLER_CALL["quats"], H.PROPELLER_CALL["frames"]))
    np.testing.assert_allclose(pt.numpy(), [H.PROPELLER_CALL["expected"]] * H.PROPELLER_CALL["frames"], rtol=1e-7)


def test_persistence_length_fit_needs_two_lags():
    """(the closed-form line fit divides by n sum d^2 - (sum d)^2, zero for fewer than two lags)"""
    import torch

    from mythos_amd.observables.persistence_length import persistence_length_fit

    with pytest.raises(ValueError, match="at least 2 lags"):
        persistence_length_fit(torch.tensor([0.9]), 0.4)
    lp, off = persistence_length_fit(torch.exp(-0.4 * torch.arange(6, dtype=torch.float64) / 120.0), 0.4)
    assert abs(float(lp) - 120.0) < 1e-9 and abs(float(off)) < 1e-12


def test_loss_wrappers_as_the_reference_tests_them():
    """mythos_amd.losses.observable_wrappers against the known answers of mythos/losses/tests/test_observable_wrapper.py:
    the base class refuses, SquaredError element-wise and on scalars, ObservableLossFn with and without the observable."""
    import numpy as np
    import torch

    from mythos_amd.losses import observable_wrappers as ow

    with pytest.raises(NotImplementedError):
        ow.LossFn()(None, None, None)
    for x, y, want in ((torch.arange(5), torch.ones(5), (torch.arange(5) - torch.ones(5)) ** 2), (torch.tensor(2), torch.tensor(1), torch.tensor(1)),
                       (torch.tensor(0), torch.tensor(0), torch.tensor(0))):
        assert torch.equal(ow.SquaredError()(x, y).double(), want.double())
    assert float(ow.RootMeanSquaredError()(torch.tensor([1.0, 3.0]), torch.tensor([0.0, 0.0]))) == pytest.approx(np.sqrt(5.0))
    assert float(ow.l2_loss(torch.tensor([1.0, 3.0]), torch.tensor([0.0, 1.0]))) == 5.0
    traj = torch.tensor([[1, 1, 1], [2, 2, 2], [3, 3, 3]], dtype=torch.float64)
    weights = torch.ones(3, dtype=torch.float64) / 3
    want_obs = (traj.sum(1) * weights).sum()
    for ret in (True, False):
        out = ow.ObservableLossFn(observable=lambda t: t.sum(1), loss_fn=lambda actual, target: actual - target, return_observable=ret)(
            trajectory=traj, target=torch.tensor(1.0), weights=weights)
        assert isinstance(out, tuple) and len(out) == (2 if ret else 1) and float(out[0]) == pytest.approx(float(want_obs) - 1.0)
        if ret:
            assert float(out[1]) == pytest.approx(float(want_obs))
    # the loss carries a gradient to the weights (what DiffTRe differentiates)
    w = weights.clone().requires_grad_(True)
    (loss,) = ow.ObservableLossFn(observable=lambda t: t.sum(1), loss_fn=ow.SquaredError())(traj, torch.tensor(5.0), w)
    loss.backward()
    assert torch.allclose(w.grad, -2.0 * (5.0 - want_obs) * traj.sum(1))
