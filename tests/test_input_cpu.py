"""CPU tests of the host-side input layer (the reference's counterparts: mythos/input/tests/): topology formats
and pair bookkeeping, trajectory parsing / quaternions / round trips, default parameters and the TOML subset."""

import io
import math
import re

import numpy as np
import pytest

from mythos_amd.input import defaults, topology, trajectory
from tests.helpers import GOLDEN


def test_bonded_and_unbonded_pairs_of_linear_and_circular_strands():
    b = topology.bonded_pairs([3, 2], [False, False])
    assert b.tolist() == [[0, 1], [1, 2], [3, 4]]
    bc = topology.bonded_pairs([3, 2], [True, False])
    assert bc.tolist() == [[0, 1], [1, 2], [0, 2], [3, 4]]  # a ring is closed by (first, last), topology.py:178-180
    u = topology.unbonded_pairs(5, b)
    assert len(u) == 5 * 4 // 2 - 3 and all(i < j for i, j in u)
    assert not ({tuple(p) for p in u} & {tuple(p) for p in b})
    top = topology.from_arrays(np.array([0, 1, 2, 3, 0]), [3, 2], is_circular=[True, False])
    assert top.is_end.tolist() == [0, 0, 0, 1, 1] and top.n_nucleotides == 5
    # ring (0, 1, 2): bonds (0,1), (1,2) and the closing (0,2) - nucleotide 0 is nn_i of two bonds, nucleotide 2 nn_j of two
    part = top.bonded_partners
    assert part.tolist() == [[-1, 1, -1, 2], [0, 2, -1, -1], [1, -1, 0, -1], [-1, 4, -1, -1], [3, -1, -1, -1]]
    lin = topology.from_arrays(np.array([0, 1, 2]), [3]).bonded_partners
    assert lin.tolist() == [[-1, 1, -1, -1], [0, 2, -1, -1], [1, -1, -1, -1]]


def test_classic_and_new_topology_formats_describe_the_same_system():
    base = GOLDEN / "dna1" / "simple-helix-seq-dep"
    with _nowarn():
        classic, f_classic = topology.from_oxdna_file(base / "generated.top", return_format=True)
        new, f_new = topology.from_oxdna_file(base / "generated-new.top", return_format=True)
    assert f_classic == topology.oxDNAFormat.CLASSIC and f_new == topology.oxDNAFormat.NEW
    assert classic.n_nucleotides == new.n_nucleotides and np.array_equal(classic.strand_counts, new.strand_counts)
    assert np.array_equal(classic.bonded_neighbors, new.bonded_neighbors)
    # the new format lists 5'->3' and is reversed per strand on load (topology.py:291): same memory order
    assert np.array_equal(classic.seq, new.seq)


class _nowarn:
    def __enter__(self):
        import warnings

        self._c = warnings.catch_warnings()
        self._c.__enter__()
        warnings.simplefilter("ignore")

    def __exit__(self, *a):
        return self._c.__exit__(*a)


def test_quaternion_axes_round_trip_and_handedness():
    rng = np.random.default_rng(0)
    q = rng.standard_normal((200, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    a1, a2, a3 = trajectory.quaternion_to_axes(q)
    np.testing.assert_allclose(np.cross(a3, a1), a2, atol=1e-14)  # a2 = a3 x a1 (trajectory.py:164-170)
    np.testing.assert_allclose((a1 * a1).sum(1), 1.0, atol=1e-14)
    q2 = trajectory.axes_to_quaternion(a1, a3)
    sign = np.sign((q * q2).sum(1, keepdims=True))
    np.testing.assert_allclose(q2 * sign, q, atol=1e-12)
    # near-180-degree rotations (w ~ 0) take the other Shepperd branches
    q_pi = np.array([[1e-9, 1.0, 0.0, 0.0], [0.0, 0.0, 1.0, 0.0], [1e-12, 0.0, 0.0, 1.0]])
    q_pi /= np.linalg.norm(q_pi, axis=1, keepdims=True)
    b1, _, b3 = trajectory.quaternion_to_axes(q_pi)
    q3 = trajectory.axes_to_quaternion(b1, b3)
    c1, _, c3 = trajectory.quaternion_to_axes(q3)
    np.testing.assert_allclose(c1, b1, atol=1e-8)
    np.testing.assert_allclose(c3, b3, atol=1e-8)


def test_trajectory_parse_slice_and_write_round_trip(tmp_path):
    base = GOLDEN / "dna2" / "simple-helix"
    with _nowarn():
        top = topology.from_oxdna_file(base / "generated.top")
    traj = trajectory.from_file(base / "output.dat", top.strand_counts, is_5p_3p=False)
    assert traj.center.shape == (100, 16, 3) and traj.quaternions.shape == (100, 16, 4)
    assert traj.box_size.shape == (3,) and np.all(traj.box_size > 0)
    np.testing.assert_allclose(np.linalg.norm(traj.a1, axis=-1), 1.0, atol=1e-6)
    part = traj.slice(slice(10, 13))
    assert part.center.shape[0] == 3 and np.array_equal(part.center[0], traj.center[10])
    out = tmp_path / "copy.dat"
    part.to_file(out)
    back = trajectory.from_file(out, top.strand_counts, is_5p_3p=False)
    np.testing.assert_allclose(back.center, part.center, atol=1e-12)
    np.testing.assert_allclose(back.a1, part.a1, atol=1e-12)
    np.testing.assert_allclose(back.a3, part.a3, atol=1e-12)
    # 5'->3' trajectories are reversed per strand on load (trajectory.py:309-313)
    rev = trajectory.from_file(out, top.strand_counts, is_5p_3p=True)
    n0 = int(top.strand_counts[0])
    np.testing.assert_allclose(rev.center[0, :n0], part.center[0, :n0][::-1], atol=1e-12)


def test_native_trajectory_reader_matches_the_numpy_parse(tmp_path):
    """mythos_oxdna_read_trajectory (host C++ in libmythos_hip.so) against the line-by-line parse, bit for bit."""
    from mythos_amd import _lib

    if not _lib.lib_path().exists():
        pytest.skip("libmythos_hip.so not built")
    for model, case in (("dna1", "simple-helix"), ("dna2", "simple-helix"), ("dna1", "simple-coax")):
        base = GOLDEN / model / case
        if not (base / "output.dat").exists():
            continue
        with _nowarn():
            top = topology.from_oxdna_file(base / "generated.top")
        for rev in (False, True):
            a = trajectory.from_file(base / "output.dat", top.strand_counts, is_5p_3p=rev, native=True)
            b = trajectory.from_file(base / "output.dat", top.strand_counts, is_5p_3p=rev, native=False)
            assert np.array_equal(a.frames, b.frames) and np.array_equal(a.times, b.times)
            assert np.array_equal(a.energies, b.energies) and np.array_equal(a.box_size, b.box_size)
    # exponents, missing trailing newline, blank lines between frames, CRLF
    rows = "\n".join(" ".join(f"{(i * 15 + k) * 1e-3:.6e}" for k in range(15)) for i in range(2))
    text = f"t = 0\nb = 10 10 10\nE = -1.5 2e-1 0\n{rows}\n\nt = 1e3\r\nb = 10 10 10\r\nE = 0 0 0\r\n{rows}"
    f = tmp_path / "odd.dat"
    f.write_text(text)
    a = trajectory.from_file(f, [2], is_5p_3p=False, native=True)
    b = trajectory.from_file(f, [2], is_5p_3p=False, native=False)
    assert a.frames.shape == (2, 2, 15) and np.array_equal(a.frames, b.frames) and a.times.tolist() == [0.0, 1000.0]
    # wrong strand lengths fail the same way on both paths
    for native in (True, False):
        with pytest.raises(ValueError, match=re.escape(trajectory.ERR_N_NUCLEOTIDE_STRAND_LENGTHS)):
            trajectory.from_file(f, [3], is_5p_3p=False, native=native)
        with pytest.raises(ValueError, match=re.escape(trajectory.ERR_N_NUCLEOTIDE_STRAND_LENGTHS)):
            trajectory.from_file(f, [1], is_5p_3p=False, native=native)
    # empty file: zero frames
    e = tmp_path / "empty.dat"
    e.write_text("")
    assert trajectory.from_file(e, [2], native=True).frames.shape == (0, 2, 15)


def test_native_trajectory_writer_round_trips_and_matches_the_numpy_writer(tmp_path):
    """mythos_oxdna_write_trajectory and the numpy writer: what they write parses back to EXACTLY the arrays that
    were written (shortest round-trip text, as the reference's str(float) / 17 significant digits), headers included;
    written by several threads, frames stay in order."""
    from mythos_amd import _lib

    if not _lib.lib_path().exists():
        pytest.skip("libmythos_hip.so not built")
    rng = np.random.default_rng(5)
    n, s = 37, 23
    frames = rng.normal(size=(s, n, 15)) * 10.0 ** rng.integers(-8, 8, size=(s, n, 15))
    frames[0, 0, :3] = [0.0, -0.0, 1e-300]
    traj = trajectory.Trajectory(n_nucleotides=n, strand_lengths=[30, 7], times=np.arange(s) * 1e3 + 0.5,
                                 energies=rng.normal(size=(s, 3)), frames=frames, box_size=np.array([12.5, 13.0, 1e2]))
    a, b = tmp_path / "native.dat", tmp_path / "numpy.dat"
    traj.to_file(a, native=True)
    traj.to_file(b, native=False)
    ra = trajectory.from_file(a, [30, 7], is_5p_3p=False, native=False)
    rb = trajectory.from_file(b, [30, 7], is_5p_3p=False, native=False)
    assert np.array_equal(ra.frames, rb.frames) and np.array_equal(ra.times, rb.times)
    assert np.array_equal(ra.box_size, rb.box_size)
    assert np.array_equal(ra.energies, rb.energies) and np.array_equal(ra.energies, traj.energies)
    assert np.array_equal(ra.frames, frames) and np.array_equal(rb.frames, frames)  # bit for bit: restarts lose nothing
    assert np.array_equal(ra.times, traj.times)
    # the native text is the shortest that round-trips, e.g. 0.1 stays "0.1"
    traj1 = trajectory.Trajectory(n_nucleotides=1, strand_lengths=[1], times=np.array([0.0]), energies=np.zeros((1, 3)),
                                  frames=np.full((1, 1, 15), 0.1), box_size=np.array([1.0, 1.0, 1.0]))
    traj1.to_file(tmp_path / "short.dat", native=True)
    assert (tmp_path / "short.dat").read_text().splitlines()[3].split()[0] == "0.1"
    # SimulatorTrajectory.to_file goes through the same writer
    import torch

    from mythos_amd.energy.base import Quaternion
    from mythos_amd.simulators.io import SimulatorTrajectory

    q = rng.normal(size=(3, n, 4))
    q /= np.linalg.norm(q, axis=-1, keepdims=True)
    st = SimulatorTrajectory(center=torch.as_tensor(frames[:3, :, :3]), orientation=Quaternion(vec=torch.as_tensor(q)))
    st.to_file(tmp_path / "sim_native.dat", native=True)
    st.to_file(tmp_path / "sim_numpy.dat", native=False)
    sa = trajectory.from_file(tmp_path / "sim_native.dat", [30, 7], is_5p_3p=False)
    sb = trajectory.from_file(tmp_path / "sim_numpy.dat", [30, 7], is_5p_3p=False)
    assert np.array_equal(sa.frames, sb.frames) and sa.times.tolist() == [0.0, 1.0, 2.0]


def test_default_parameters_and_toml_subset(tmp_path):
    sim1, e1 = defaults.default_configs_for("dna1")
    sim2, e2 = defaults.default_configs_for("dna2")
    assert abs(sim2["kT"] - 296.15 * 0.1 / 300.0) < 1e-12 and sim2["salt_conc"] == 0.5 and sim2["dt"] == 5e-3
    assert e1["fene"]["r0_backbone"] == 0.7525 and e2["fene"]["r0_backbone"] == 0.7564
    assert e1["stacking"]["eps_stack_base"] == 1.3448 and e2["stacking"]["eps_stack_base"] == 1.3523
    assert abs(e2["coaxial_stacking"]["theta0_coax_1"] - (math.pi - 0.25)) < 1e-12
    assert "debye" in e2 and "debye" not in e1
    sim3, e3 = defaults.default_configs_for("rna2")  # mythos/input/rna2/default_energy.toml
    assert e3["fene"]["r0_backbone"] == 0.761070781051 and e3["geometry"]["pos_back_a3"] == 0.2 and "theta0_stack_9" in e3["stacking"]
    assert "theta0_cross_4" not in e3["cross_stacking"] and sim3["salt_conc"] == 1.0
    sim4, e4 = defaults.default_configs_for("na1")  # three sets: oxDNA2, oxRNA2, mythos/input/na1/default_energy.toml
    assert set(e4) == {"dna", "rna", "drh"} and e4["drh"]["hydrogen_bonding"]["eps_hb"] == 1.5 and "fene" not in e4["drh"]
    with pytest.raises((KeyError, ValueError)):
        defaults.default_configs_for("dna3")
    toml = tmp_path / "p.toml"
    toml.write_text("# comment\n[a]\nx = 1.5\ny = \"pi - 0.25\"\nflag = true\n[b]\nz = [1, 2.0, \"3 * 2\"]\n")
    parsed = defaults.parse_toml(toml)
    assert parsed["a"]["x"] == 1.5 and abs(parsed["a"]["y"] - (math.pi - 0.25)) < 1e-12 and parsed["a"]["flag"] is True
    assert parsed["b"]["z"][2] == 6 and defaults.parse_toml(toml, key="a") == parsed["a"]
    # strings that are not arithmetic stay strings: nothing is ever evaluated as Python
    evil = "__import__('os').system('true')"
    assert defaults.parse_str(evil) == evil and defaults.parse_str("sqrt(4) + pi") == 2.0 + math.pi


SYSDEFS = (("persistence-length-500bp", "init.top", "relaxed.dat", 1000, 350.0), ("wlc-fit", "generated.top", "generated.dat", 220, 200.0))


@pytest.mark.parametrize("name, top_file, conf_file, n, box", SYSDEFS)
def test_system_definitions_the_reference_ships_read_as_relaxed_duplexes(name, top_file, conf_file, n, box):
    """data/sys-defs of the reference (inputs of its example notebooks): a relaxed 500 bp duplex and the 110 bp WLC system.
    Both readers agree on them bit for bit, and the oracle sees what a thermalised duplex is: every base pair bonded,
    about -1.5 units per nucleotide, no coaxial stacking, in both models."""
    import torch

    from mythos_amd import _lib
    from oracle import oxdna_oracle as orc
    from tests import helpers as H

    base = GOLDEN / "sys-defs" / name
    with _nowarn():
        top = topology.from_oxdna_file(base / top_file)
    assert top.n_nucleotides == n and list(top.strand_counts) == [n // 2, n // 2]
    assert len(top.bonded_neighbors) == n - 2 and len(top.unbonded_neighbors) == n * (n - 1) // 2 - (n - 2)
    tr = trajectory.from_file(base / conf_file, top.strand_counts, is_5p_3p=False, native=False)
    assert tr.center.shape == (1, n, 3) and np.allclose(tr.box_size, box)
    if _lib.lib_path().exists():
        nat = trajectory.from_file(base / conf_file, top.strand_counts, is_5p_3p=False, native=True)
        assert np.array_equal(nat.frames, tr.frames) and np.array_equal(nat.box_size, tr.box_size)
    assert np.allclose(np.linalg.norm(tr.quaternions[0], axis=1), 1.0, atol=1e-12)
    seq, is_end, b, u = H.topo_tensors(top)
    for model in (1, 2):
        P = H.oracle_params(model, half_charged_ends=(model == 2))
        e = orc.energy_terms(model, P, torch.as_tensor(tr.center[0]), torch.as_tensor(tr.quaternions[0]), seq, is_end, b, u,
                             box=tr.box_size).numpy()
        assert np.isfinite(e).all() and -1.6 < e.sum() / n < -1.4, (model, e)
        # terms: fene, bonded excl., stacking, unbonded excl., H-bond, cross-stacking, coaxial (, Debye)
        assert e[0] > 0 and -1.25 < e[2] / n < -1.05 and -0.40 < e[4] / n < -0.28 and -0.14 < e[5] / n < -0.09 and abs(e[6]) < 1e-6


def test_sequence_dependence_files_read_like_the_reference(tmp_path):
    """mythos_amd.input.sequence_dependence.read_ss_weights (mythos/input/sequence_dependence.py:12-51): the two files the
    reference ships for its users, the goldens' own, the `f` suffix and white space, one member of a Watson-Crick pair
    standing for both, a missing entry as a KeyError."""
    from mythos_amd.input.sequence_dependence import read_ss_weights
    from tests import helpers as H

    for path in (GOLDEN / "seq-specific" / "seq_oxdna1.txt", GOLDEN / "seq-specific" / "seq_oxdna2.txt",
                 GOLDEN / "dna1" / "simple-helix-seq-dep" / "seq_dep.dat"):
        got, want = read_ss_weights(path), H.read_ss_weights(path)
        assert set(got) == {"eps_stack_kt_coeff", "ss_stack_weights", "ss_hb_weights"}
        assert got["ss_stack_weights"].dtype.is_floating_point and got["ss_stack_weights"].shape == (4, 4)
        for k in got:
            assert np.array_equal(np.asarray(got[k]), np.asarray(want[k])), (path.name, k)
        hb = np.asarray(got["ss_hb_weights"])
        assert np.count_nonzero(hb) == 4 and np.array_equal(hb, hb.T) and hb[0, 3] > 0 and hb[1, 2] > 0  # A-T, C-G (A, C, G, T)
    w2 = read_ss_weights(GOLDEN / "seq-specific" / "seq_oxdna2.txt")
    assert float(w2["eps_stack_kt_coeff"]) == 0.18 and float(w2["ss_stack_weights"][0, 0]) == 1.84642  # STCK_A_A
    lines = [ln for ln in (GOLDEN / "seq-specific" / "seq_oxdna2.txt").read_text().splitlines() if ln.strip()]
    odd = tmp_path / "odd.txt"
    odd.write_text("\n".join(("  " + ln.replace(" = ", "=") + "f" if k % 2 else ln) for k, ln in enumerate(lines)
                             if not ln.startswith(("HYDR_T_A", "HYDR_G_C"))) + "\n\n")
    w3 = read_ss_weights(odd)
    for k in w2:
        assert np.array_equal(np.asarray(w3[k]), np.asarray(w2[k])), k
    odd.write_text("\n".join(ln for ln in lines if not ln.startswith("STCK_G_A")))
    with pytest.raises(KeyError, match="STCK_G_A"):
        read_ss_weights(odd)
    odd.write_text("\n".join(ln for ln in lines if not ln.startswith(("HYDR_A_T", "HYDR_T_A"))))
    with pytest.raises(KeyError, match="HYDR_T_A"):
        read_ss_weights(odd)
