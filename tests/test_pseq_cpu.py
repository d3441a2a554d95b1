"""Probabilistic sequences on the host: the restated constraint containers (mythos/input/sequence_constraints.py), the
oracle's expected pair weight (mythos/energy/utils.py:45-132) against the reference's known answers, its golden
energies with a one-hot distribution, and the brute-force enumeration the reference uses as its own check
(mythos/energy/dna1/tests/test_expected_energies.py:162-328)."""

import itertools
import warnings

import numpy as np
import pytest
import torch

from mythos_amd.input import sequence_constraints as scm
from mythos_amd.input import topology, trajectory
from oracle import oxdna_oracle as orc
from tests import helpers as H

# the case of mythos/energy/tests/test_utils.py:10-150: 4 nucleotides, one base pair (0, 3), nucleotides 1, 2 unpaired
UP = np.array([[0.27, 0.03, 0.68, 0.02], [0.04, 0.56, 0.22, 0.18]])
BP = np.array([[0.66, 0.14, 0.01, 0.19]])
TABLE = np.array([[0.2, 0.1, 0.3, 0.4], [0.05, 0.25, 0.1, 0.6], [0.55, 0.15, 0.2, 0.1], [0.1, 0.15, 0.6, 0.15]])


def _sc4():
    return scm.SequenceConstraints(n_nucleotides=4, n_unpaired=2, n_bp=1, is_unpaired=np.array([0, 1, 1, 0]), unpaired=np.array([1, 2]),
                                   bps=np.array([[0, 3]]), idx_to_unpaired_idx=np.array([-1, 0, 1, -1]),
                                   idx_to_bp_idx=np.array([[0, 0], [-1, -1], [-1, -1], [0, 1]]))


def test_from_bps_and_dseq_to_pseq_known_answers():
    """mythos/input/tests/test_sequence_constraints.py:296-353."""
    sc, want = scm.from_bps(4, np.array([[0, 3]])), _sc4()
    for f in ("n_nucleotides", "n_unpaired", "n_bp"):
        assert getattr(sc, f) == getattr(want, f)
    for f in ("is_unpaired", "unpaired", "bps", "idx_to_unpaired_idx", "idx_to_bp_idx"):
        np.testing.assert_array_equal(getattr(sc, f), getattr(want, f))
    up, bp = scm.dseq_to_pseq(np.array([0, 1, 2, 3]), want)
    np.testing.assert_allclose(up, [[0, 1, 0, 0], [0, 0, 1, 0]])
    np.testing.assert_allclose(bp, [[1, 0, 0, 0]])
    none = scm.from_bps(3, np.zeros((0, 2), dtype=np.int32))
    assert none.n_bp == 0 and scm.dseq_to_pseq(np.array([3, 1, 0]), none)[1].shape == (1, 4)  # the dummy row of :210-214


def test_constraint_errors():
    """:277-294, :356-377 and a sample of the class validation (:13-275)."""
    with pytest.raises(ValueError, match=scm.ERR_BP_ARR_CONTAINS_DUPLICATES):
        scm.from_bps(4, np.array([[0, 3], [0, 1]]))
    with pytest.raises(ValueError, match=scm.ERR_INVALID_BP_INDICES):
        scm.from_bps(4, np.array([[0, 5]]))
    with pytest.raises(ValueError, match=scm.ERR_INVALID_BP_SHAPE):
        scm.from_bps(4, np.array([0, 3]))
    with pytest.raises(ValueError, match=scm.ERR_DSEQ_TO_PSEQ_INVALID_BP):
        scm.dseq_to_pseq(np.array([0, 1, 2, 0]), _sc4())
    good = dict(n_nucleotides=6, n_unpaired=2, n_bp=2, is_unpaired=np.array([0, 0, 1, 0, 1, 0]), unpaired=np.array([2, 4]),
                bps=np.array([[0, 5], [1, 3]]), idx_to_unpaired_idx=np.array([-1, -1, 0, -1, 1, -1]),
                idx_to_bp_idx=np.array([[0, 0], [1, 0], [-1, -1], [1, 1], [-1, -1], [0, 1]]))
    scm.SequenceConstraints(**good)
    for change, err in (
        ({"n_nucleotides": 0}, scm.ERR_SEQ_CONSTRAINTS_INVALID_NUMBER_NUCLEOTIDES),
        ({"unpaired": np.array([2])}, scm.ERR_SEQ_CONSTRAINTS_INVALID_UNPAIRED_SHAPE),
        ({"bps": np.array([[0, 5]])}, scm.ERR_INVALID_BP_SHAPE),
        ({"is_unpaired": np.array([0, 0, 1, 0, 1])}, scm.ERR_SEQ_CONSTRAINTS_INVALID_IS_UNPAIRED_SHAPE),
        ({"idx_to_unpaired_idx": np.array([-1, -1, 0, -1, 1])}, scm.ERR_SEQ_CONSTRAINTS_INVALID_UNPAIRED_MAPPER_SHAPE),
        ({"idx_to_bp_idx": np.array([[0, 0], [1, 0], [-1, -1], [1, 1], [-1, -1]])}, scm.ERR_SEQ_CONSTRAINTS_INVALID_BP_MAPPER_SHAPE),
        ({"is_unpaired": np.array([0, 0, 2, 0, 1, 0])}, scm.ERR_SEQ_CONSTRAINTS_IS_UNPAIRED_INVALID_VALUES),
        ({"is_unpaired": np.array([0, 1, 0, 0, 1, 0])}, scm.ERR_SEQ_CONSTRAINTS_INVALID_IS_UNPAIRED),
        ({"bps": np.array([[0, 5], [1, 2]])}, scm.ERR_SEQ_CONSTRAINTS_INVALID_COVER),
        ({"idx_to_unpaired_idx": np.array([0, -1, 0, -1, 1, -1])}, scm.ERR_SEQ_CONSTRAINTS_PAIRED_NT_MAPPED_TO_UNPAIRED),
        ({"idx_to_unpaired_idx": np.array([-1, -1, 0, -1, 0, -1])}, scm.ERR_SEQ_CONSTRAINTS_INCOMPLETE_UNPAIRED_MAPPED_IDXS),
        ({"idx_to_bp_idx": np.array([[0, 0], [1, 0], [0, 0], [1, 1], [-1, -1], [0, 1]])}, scm.ERR_SEQ_CONSTRAINTS_UNPAIRED_NT_MAPPED_TO_PAIRED),
        ({"idx_to_bp_idx": np.array([[0, 0], [1, 0], [-1, -1], [1, 0], [-1, -1], [0, 1]])}, scm.ERR_SEQ_CONSTRAINTS_INCOMPLETE_BP_MAPPED_IDXS),
    ):
        with pytest.raises(ValueError, match=err):
            scm.SequenceConstraints(**{**good, **change})


@pytest.mark.parametrize(("nt1", "nt2", "expected"), [
    # both members of the one base pair (test_utils.py:13-38): AT, TA, GC, CG read table[0][3], [3][0], [2][1], [1][2]
    (0, 3, 0.66 * 0.4 + 0.14 * 0.1 + 0.01 * 0.15 + 0.19 * 0.1),
    # nt1 in the base pair (member 0), nt2 unpaired (:39-76)
    (0, 1, sum(b * u * TABLE[a, k] for b, a in zip(BP[0], (0, 3, 2, 1)) for k, u in enumerate(UP[0]))),
    # both unpaired (:77-112)
    (1, 2, float(UP[0] @ TABLE @ UP[1])),
    # nt1 unpaired, nt2 member 1 of the base pair (:113-150): AT, TA, GC, CG give nt2 = T, A, C, G
    (2, 3, sum(u * b * TABLE[k, a] for k, u in enumerate(UP[1]) for b, a in zip(BP[0], (3, 0, 1, 2)))),
])
def test_expected_weight_known_answers(nt1, nt2, expected):
    sc = _sc4()
    w = orc.compute_seq_dep_weight((UP, BP), torch.tensor([nt1]), torch.tensor([nt2]), TABLE, sc)
    assert np.isclose(float(w[0]), expected, rtol=1e-12)
    # the kernel's formulation - per-nucleotide marginals, base pairs tied through their type - is the same number
    marg, unit, bp = scm.kernel_tables((UP, BP), sc)
    if unit[nt1] >= 0 and unit[nt2] >= 0 and unit[nt1] // 2 == unit[nt2] // 2:
        wk = sum(bp[unit[nt1] // 2, t] * TABLE[scm.BP_IDXS[t][unit[nt1] & 1], scm.BP_IDXS[t][unit[nt2] & 1]] for t in range(4))
    else:
        wk = marg[nt1] @ TABLE @ marg[nt2]
    assert np.isclose(wk, expected, rtol=1e-12)


def test_the_fourth_known_answer_as_the_reference_spells_it():
    """test_utils.py:113-150 lists this sum term by term."""
    want = (0.04 * 0.14 * 0.2 + 0.04 * 0.01 * 0.1 + 0.04 * 0.19 * 0.3 + 0.04 * 0.66 * 0.4 + 0.56 * 0.14 * 0.05 + 0.56 * 0.01 * 0.25
            + 0.56 * 0.19 * 0.1 + 0.56 * 0.66 * 0.6 + 0.22 * 0.14 * 0.55 + 0.22 * 0.01 * 0.15 + 0.22 * 0.19 * 0.2 + 0.22 * 0.66 * 0.1
            + 0.18 * 0.14 * 0.1 + 0.18 * 0.01 * 0.15 + 0.18 * 0.19 * 0.6 + 0.18 * 0.66 * 0.15)
    w = orc.compute_seq_dep_weight((UP, BP), torch.tensor([2]), torch.tensor([3]), TABLE, _sc4())
    assert np.isclose(float(w[0]), want, rtol=1e-12)


@pytest.mark.parametrize(("case", "weights", "top_file"), [("simple-helix", False, "generated.top"), ("simple-helix-seq-dep", True, "generated.top"),
                                                           ("simple-helix-seq-dep", True, "generated-new.top")])
def test_one_hot_pseq_reproduces_the_golden_energies(case, weights, top_file):
    """dna1/tests/test_integration.py:192-293 with use_pseq=True: a one-hot distribution over the discrete sequence
    gives oxDNA's own hydrogen-bonding (1e-3) and stacking (1e-6) energies."""
    top, traj, split, _ = H.load_golden(1, case, top_file)
    over = {}
    if weights:
        ss = H.read_ss_weights(H.GOLDEN / "dna1" / case / "seq_dep.dat")
        over = {"stacking": {"ss_stack_weights": ss["ss_stack_weights"], "eps_stack_kt_coeff": ss["eps_stack_kt_coeff"]},
                "hydrogen_bonding": {"ss_hb_weights": ss["ss_hb_weights"]}}
    sc = scm.from_bps(top.n_nucleotides, np.array([[0, 15]]) if not weights else np.zeros((0, 2), dtype=np.int32))
    pseq = scm.dseq_to_pseq(top.seq, sc)
    for sec in ("stacking", "hydrogen_bonding"):
        over.setdefault(sec, {}).update({"pseq": pseq, "pseq_constraints": sc})
    P = H.oracle_params(1, overrides=over)
    e = H.oracle_terms_traj(1, P, top, traj, frames=range(0, 100, 7))
    np.testing.assert_allclose(np.around(e[:, 4], 6), split[0:100:7, 5], atol=1e-3)
    if top_file == "generated.top":
        np.testing.assert_allclose(np.around(e[:, 2], 6), split[0:100:7, 3], atol=1e-6)


def _helix4():
    base = H.GOLDEN / "dna1" / "helix-4bp"
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        top = topology.from_oxdna_file(base / "sys.top")
    traj = trajectory.from_file(base / "output.dat", top.strand_counts, is_5p_3p=False)
    return top, traj


def enumerate_sequences(sc, up, bp):
    """Every discrete sequence the constraints allow, with its probability (test_expected_energies.py:140-158, 208-228)."""
    for idxs in itertools.product(range(4), repeat=sc.n_unpaired + sc.n_bp):
        seq, prob = np.zeros(sc.n_nucleotides, dtype=np.int64), 1.0
        for k, u in enumerate(sc.unpaired):
            seq[u] = idxs[k]
            prob *= up[k, idxs[k]]
        for k, (a, b) in enumerate(sc.bps):
            t = idxs[sc.n_unpaired + k]
            seq[a], seq[b] = scm.BP_IDXS[t]
            prob *= bp[k, t]
        yield seq, prob


@pytest.mark.parametrize("term", ["hydrogen_bonding", "stacking"])
def test_expected_energy_equals_brute_force_enumeration(term):
    """test_expected_energies.py:162-235 (hydrogen bonding), 254-328 (stacking): random weight table, random
    distributions, constraints bps = (0,7), (1,6), (2,5) on the 8-nt duplex; atol 1e-4 there, 1e-10 here."""
    top, traj = _helix4()
    sc = scm.from_bps(8, np.array([[0, 7], [1, 6], [2, 5]]))
    rng = np.random.default_rng(7)
    table = rng.random((4, 4))
    table /= table.sum(1, keepdims=True)
    up = rng.random((sc.n_unpaired, 4))
    up /= up.sum(1, keepdims=True)
    bp = rng.random((sc.n_bp, 4))
    bp /= bp.sum(1, keepdims=True)
    key = "ss_hb_weights" if term == "hydrogen_bonding" else "ss_stack_weights"
    P_d = H.oracle_params(1, overrides={term: {key: table}})
    P_p = H.oracle_params(1, overrides={term: {key: table, "pseq": (up, bp), "pseq_constraints": sc}})
    col = 4 if term == "hydrogen_bonding" else 2
    frames = [0, 55]
    seq_t, is_end, b, u = H.topo_tensors(top)

    def term_energy(P, seq):
        return np.array([float(orc.energy_terms(1, P, torch.as_tensor(traj.center[f]), torch.as_tensor(traj.quaternions[f]), seq, is_end, b, u,
                                                box=traj.box_size)[col]) for f in frames])

    got = term_energy(P_p, seq_t)
    want = sum(prob * term_energy(P_d, torch.as_tensor(seq)) for seq, prob in enumerate_sequences(sc, up, bp))
    assert np.abs(want).max() > 1e-3
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-10)


def test_differentiable_kernel_tables_equal_the_plain_ones():
    """kernel_tables_torch: the same marginals and base-pair rows as kernel_tables, and their Jacobian is the 0 / 1
    incidence the chain rule of dU/d(pseq) needs (a paired nucleotide's marginal sums its pair's type probabilities)."""
    import torch

    from mythos_amd.input import sequence_constraints as scm

    sc = scm.from_bps(8, np.array([[0, 7], [2, 5]]))
    rng = np.random.default_rng(0)
    up = rng.random((sc.n_unpaired, 4))
    bp = rng.random((2, 4))
    m, unit, b = scm.kernel_tables((up, bp), sc)
    up_t, bp_t = torch.tensor(up, requires_grad=True), torch.tensor(bp, requires_grad=True)
    mt, bt = scm.kernel_tables_torch((up_t, bp_t), sc)
    np.testing.assert_array_equal(mt.detach().numpy(), m)
    np.testing.assert_array_equal(bt.detach().numpy(), b)
    w = torch.tensor(rng.random(m.shape))
    g_up, g_bp = torch.autograd.grad((mt * w).sum(), [up_t, bp_t])
    for k, idx in enumerate(sc.unpaired):
        np.testing.assert_allclose(g_up[k].numpy(), w[idx].numpy())
    # type t of pair k puts probability on base BP_IDXS[t][0] of its first member and BP_IDXS[t][1] of its second
    for k, (i, j) in enumerate(sc.bps):
        for t in range(4):
            assert abs(float(g_bp[k, t]) - float(w[i, scm.BP_IDXS[t, 0]] + w[j, scm.BP_IDXS[t, 1]])) < 1e-15
    sc0 = scm.from_bps(4, np.zeros((0, 2), dtype=np.int32))
    m0, _, b0 = scm.kernel_tables((np.eye(4), np.zeros((1, 4))), sc0)
    mt0, bt0 = scm.kernel_tables_torch((torch.eye(4, dtype=torch.float64), torch.zeros(1, 4)), sc0)
    assert np.array_equal(mt0.numpy(), m0) and bt0.shape == b0.shape
