"""The C++/OpenMP CPU port (oracle/cpu_port) - bench.py's second CPU baseline and the sanitizer target - held to the
golden-pinned torch oracle, and run under AddressSanitizer + UBSan.

It compiles the same pair-physics templates as the HIP kernels for the host, so a clean sanitizer run and parity with
the oracle here say something about those templates too (the GPU pool offers no sanitizer)."""

import subprocess

import numpy as np
import pytest
import torch

from mythos_amd.energy import flat_params as fp
from mythos_amd.input import defaults
from mythos_amd.simulators.neighbors import verlet_pairs_numpy
from mythos_amd.utils import generators
from oracle import cpu_port
from oracle import oxdna_oracle as orc
from oracle.langevin_oracle import LangevinOracle
from tests import helpers as H

KT = 296.15 * 0.1 / 300.0


SALT = {1: 0.5, 2: 0.5, 3: 1.0}  # rna2 goldens: salt 1.0 (rna2/tests/test_integration.py)


def _flat(model, hce):
    sim, cfg = defaults.default_configs_for(H.model_dir(model))
    d = fp.derive_flat(model, cfg, kt=sim["kT"], salt_conc=SALT[model], half_charged_ends=hce)
    from mythos_amd import _lib

    return fp.pack_flat(d, _lib.param_names()).numpy()


@pytest.fixture(scope="module", autouse=True)
def _built():
    if not cpu_port.LIB.exists():
        subprocess.run(["make", "-C", str(cpu_port.BUILD.parent)], check=True, capture_output=True)


@pytest.mark.parametrize(("model", "name", "hce"), [(1, "simple-helix", False), (2, "simple-helix", False), (2, "simple-coax", False),
                                                    (2, "simple-helix-half-charged-ends", True),
                                                    (3, "simple-helix-12bp", False), (3, "simple-coax", False)])
def test_energies_and_gradients_match_oracle(model, name, hce):
    top, traj, split, _ = H.load_golden(model, name)
    port = cpu_port.CpuPort(model, top.seq, top.is_end, top.bonded_neighbors, _flat(model, hce), box=traj.box_size)
    port.set_pairs(top.unbonded_neighbors)
    P = H.oracle_params(model, half_charged_ends=hce, salt=SALT[model])
    tt = H.topo_tensors(top)
    for f in (0, 41, 99):
        c, q = traj.center[f], traj.quaternions[f]
        e, gc, gq, tb = port.energy(c, q)
        e_ref = orc.energy_terms(model, P, torch.as_tensor(c), torch.as_tensor(q), *tt, box=traj.box_size).numpy()
        np.testing.assert_allclose(e[: len(e_ref)], e_ref, rtol=0, atol=1e-10)
        # and the golden file itself, at the reference's tolerances
        for k, term in enumerate(H.SPLIT_COLUMNS[1 : 1 + len(e_ref)]):
            if model == 1 and term == "stacking" and name != "simple-helix":
                continue
            assert abs(e[k] / top.n_nucleotides - split[f, 1 + k]) <= H.TERM_ATOL[term] + 5e-7, term
        _, gc_ref, gq_ref = orc.energy_and_grads(model, P, torch.as_tensor(c), torch.as_tensor(q), *tt, box=traj.box_size)
        np.testing.assert_allclose(gc, gc_ref.numpy(), rtol=0, atol=1e-9 * max(1.0, gc_ref.abs().max().item()))
        np.testing.assert_allclose(gq, gq_ref.numpy(), rtol=0, atol=1e-9 * max(1.0, gq_ref.abs().max().item()))
        tb_ref = orc.quat_grad_to_body_torque(torch.as_tensor(q), gq_ref).numpy()
        np.testing.assert_allclose(tb, tb_ref, rtol=0, atol=1e-9 * max(1.0, np.abs(tb_ref).max()))


@pytest.mark.parametrize("model", [1, 2, 3, 4])
@pytest.mark.parametrize("bonded", [False, True])
def test_random_dimers_match_oracle(model, bonded):
    """Every term of the pair templates on random relative poses (tests/helpers.py random_dimers): the golden
    trajectories hold 0 for the coaxial term in every stored frame, and visit little of the other angular windows.
    Model 4 (oxNA): random DNA / RNA types, so all four kinds of pair - the hybrid coaxial term has no golden."""
    top, c, q, live = H.random_dimers(model, bonded)
    assert all(v >= 5 for k, v in live.items() if not k.startswith("pairs")), live
    pairs = np.arange(top.n_nucleotides).reshape(-1, 2)
    seq, is_end, b, _ = H.topo_tensors(top)
    u = torch.zeros((0, 2), dtype=torch.long) if bonded else torch.as_tensor(pairs)
    ct, qt = torch.as_tensor(c), torch.as_tensor(q)
    if model == 4:
        is_rna = np.asarray(top.nt_type) == 2
        assert min(live["pairs dna-dna / rna-rna / hybrid"]) >= 30
        port = cpu_port.CpuPort(4, top.seq, top.is_end, top.bonded_neighbors, _flat_na1(), box=None, is_rna=is_rna)
        P = H.oracle_params_na1()
        e_ref = orc.energy_terms_na1(P, ct, qt, seq, torch.as_tensor(is_rna), is_end, b, u).numpy()
        _, gc_ref, gq_ref = orc.energy_and_grads_na1(P, ct, qt, seq, torch.as_tensor(is_rna), is_end, b, u)
    else:
        port = cpu_port.CpuPort(model, top.seq, top.is_end, top.bonded_neighbors, _flat(model, False), box=None)
        P = H.oracle_params(model, salt=SALT[model])
        e_ref = orc.energy_terms(model, P, ct, qt, seq, is_end, b, u).numpy()
        _, gc_ref, gq_ref = orc.energy_and_grads(model, P, ct, qt, seq, is_end, b, u)
    port.set_pairs(np.zeros((0, 2), np.int64) if bonded else pairs)
    e, gc, gq, tb = port.energy(c, q)
    np.testing.assert_allclose(e[: len(e_ref)], e_ref, rtol=1e-12, atol=1e-10)
    assert (np.abs(e_ref[:3] if bonded else e_ref[3:]) > 0.2).all()  # every term contributes
    np.testing.assert_allclose(gc, gc_ref.numpy(), rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(gq, gq_ref.numpy(), rtol=1e-9, atol=1e-9)


def _flat_na1():
    from mythos_amd import _lib

    sim, cfg = defaults.default_configs_for("na1")
    named = fp.derive_flat_na1(cfg["dna"], cfg["rna"], cfg["drh"], kt=sim["kT"], salt_conc=0.5, half_charged_ends=False)
    return fp.pack_flat_na1(named, _lib.param_names()).numpy()


@pytest.mark.parametrize("name", H.NA1_CASES)
def test_oxna_energies_and_gradients_match_oracle(name):
    """The oxNA instantiation of the pair templates (parameter vector and form by the types of the pair, sites by the type
    of each nucleotide) against the oracle on the reference's seven hybrid goldens: energies per term, forces, dU/dq."""
    top, traj, split, is_rna = H.load_golden_na1(name)
    port = cpu_port.CpuPort(4, top.seq, top.is_end, top.bonded_neighbors, _flat_na1(), box=traj.box_size, is_rna=is_rna)
    port.set_pairs(top.unbonded_neighbors)
    P = H.oracle_params_na1()
    seq, is_end, b, u = H.topo_tensors(top)
    rna = torch.as_tensor(is_rna)
    for f in (0, 41, 99):
        c, q = traj.center[f], traj.quaternions[f]
        e, gc, gq, tb = port.energy(c, q)
        e_ref = orc.energy_terms_na1(P, torch.as_tensor(c), torch.as_tensor(q), seq, rna, is_end, b, u, box=traj.box_size).numpy()
        np.testing.assert_allclose(e, e_ref, rtol=0, atol=1e-10)
        for k, term in enumerate(H.SPLIT_COLUMNS[1:9]):
            if term == "coaxial_stacking" and name == "simple-coax-dna-dna-rna":
                continue  # oxNA's standalone code has the hybrid spring constant at 0 (na1/tests/test_integration.py:404-406)
            assert abs(e[k] / top.n_nucleotides - split[f, 1 + k]) <= H.NA1_TERM_ATOL[term] + 5e-7, term
        _, gc_ref, gq_ref = orc.energy_and_grads_na1(P, torch.as_tensor(c), torch.as_tensor(q), seq, rna, is_end, b, u, box=traj.box_size)
        np.testing.assert_allclose(gc, gc_ref.numpy(), rtol=0, atol=1e-9 * max(1.0, gc_ref.abs().max().item()))
        np.testing.assert_allclose(gq, gq_ref.numpy(), rtol=0, atol=1e-9 * max(1.0, gq_ref.abs().max().item()))


def test_langevin_steps_match_oracle_on_the_same_random_stream():
    top, traj, _, _ = H.load_golden(2, "simple-helix")
    port = cpu_port.CpuPort(2, top.seq, top.is_end, top.bonded_neighbors, _flat(2, False), box=traj.box_size)
    port.set_pairs(top.unbonded_neighbors)
    gam_t, gam_r, seed, inertia = KT / 2.5, KT / 7.5, 0x9E3779B97F4A7C15, (1.0, 1.3, 0.8)
    o = LangevinOracle(2, H.oracle_params(2), H.topo_tensors(top), traj.box_size, 0.005, KT, gam_t, gam_r, 1.0, inertia, seed=seed)
    rng = np.random.default_rng(0)
    x, q = traj.center[5].copy(), traj.quaternions[5].copy()
    p, L = 0.3 * rng.standard_normal(x.shape), 0.3 * rng.standard_normal(x.shape)
    x2, q2, p2, L2 = x.copy(), q.copy(), p.copy(), L.copy()
    us = o.run(x, q, p, L, 7)
    builds, e = port.run(x2, q2, p2, L2, 7, dt=0.005, kT=KT, gamma_t=gam_t, gamma_r=gam_r, inertia=inertia, seed=seed)
    assert builds == 0
    for a, b in ((x, x2), (q, q2), (p, p2), (L, L2)):
        np.testing.assert_allclose(b, a, rtol=0, atol=1e-11)
    assert abs(e[:8].sum() - us[-1]) < 1e-10 * abs(us[-1])
    ke_t, ke_r = o.kinetic(p, L)
    assert abs(e[8] - ke_t) < 1e-10 * ke_t and abs(e[9] - ke_r) < 1e-10 * ke_r


def test_cell_list_equals_kd_tree_and_dynamic_list_equals_static():
    top, c0, q0 = generators.ideal_duplex(300, model=2, seed=4)
    flat = _flat(2, True)
    port = cpu_port.CpuPort(2, top.seq, top.is_end, top.bonded_neighbors, flat)
    nbar = port.build_pairs(c0, 3.25)
    ref = verlet_pairs_numpy(c0, top.bonded_neighbors, 3.25)
    assert abs(nbar - 2.0 * len(ref) / top.n_nucleotides) < 1e-12
    e_cells = port.energy(c0, q0)[0]
    port.set_pairs(ref)
    np.testing.assert_allclose(port.energy(c0, q0)[0], e_cells, rtol=0, atol=1e-9)
    # periodic box, random cloud folded into it: the direct grid finds every minimum-image pair
    rng = np.random.default_rng(2)
    box = np.array([14.0, 15.0, 16.5])
    cloud = rng.uniform(-20, 40, size=(top.n_nucleotides, 3))
    pb = cpu_port.CpuPort(2, top.seq, top.is_end, top.bonded_neighbors, flat, box=box)
    nb = pb.build_pairs(cloud, 3.25)
    ref_p = verlet_pairs_numpy(cloud, top.bonded_neighbors, 3.25, box=box)
    assert abs(nb - 2.0 * len(ref_p) / top.n_nucleotides) < 1e-12
    # 40 steps on a rebuilt list == 40 steps on a static list that contains every pair that ever interacts
    out = []
    for dynamic in (False, True):
        port = cpu_port.CpuPort(2, top.seq, top.is_end, top.bonded_neighbors, flat)
        x, q = c0.copy(), q0.copy()
        rs = np.random.default_rng(1)
        p, L = 0.3 * rs.standard_normal(x.shape), 0.3 * rs.standard_normal(x.shape)
        if not dynamic:
            port.set_pairs(verlet_pairs_numpy(c0, top.bonded_neighbors, 4.5))
        builds, _ = port.run(x, q, p, L, 40, dt=0.005, kT=KT, gamma_t=KT / 2.5, gamma_r=KT / 7.5, seed=3,
                             r_cut=3.25 if dynamic else 0.0, skin=0.2 if dynamic else 0.0, rebuild_every=7 if dynamic else 0)
        assert builds >= (6 if dynamic else 0)
        out.append((x, q, p))
    for a, b in zip(*out):
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-9)


def test_thread_count_does_not_change_results():
    top, c0, q0 = generators.ideal_duplex(120, model=2, seed=4)
    port = cpu_port.CpuPort(2, top.seq, top.is_end, top.bonded_neighbors, _flat(2, True))
    port.build_pairs(c0, 3.25)
    n0 = cpu_port.threads()
    try:
        res = []
        for t in (1, max(2, n0)):
            cpu_port.set_threads(t)
            x, q = c0.copy(), q0.copy()
            p, L = np.zeros_like(x), np.zeros_like(x)
            port.run(x, q, p, L, 10, dt=0.005, kT=KT, gamma_t=KT / 2.5, gamma_r=KT / 7.5, seed=1)
            res.append((x, p, L))
        for a, b in zip(*res):
            assert np.array_equal(a, b)  # gathers, no atomics: forces do not depend on the number of threads
    finally:
        cpu_port.set_threads(n0)


def test_sanitizers_are_clean_and_agree_with_the_plain_build(tmp_path):
    """AddressSanitizer + UndefinedBehaviorSanitizer over energy, the cell-list build and 20 Langevin steps with
    rebuilds (oxDNA1 and oxDNA2; free and periodic): no report, and the same numbers as the optimised build."""
    plain, san = cpu_port.BUILD / "md_cpu_selftest", cpu_port.BUILD / "md_cpu_selftest_san"
    if not san.exists():
        subprocess.run(["make", "-C", str(cpu_port.BUILD.parent)], check=True, capture_output=True)
    for model, box in ((2, None), (1, np.array([40.0, 40.0, 60.0])), (3, np.array([20.0, 20.0, 20.0])), (4, np.array([20.0, 20.0, 20.0]))):
        is_rna = None
        if model == 3:    # oxRNA2: the reference's RNA helix (an ideal B-duplex is outside the FENE well of the RNA geometry)
            top, traj, _, _ = H.load_golden(3, "simple-helix-12bp")
            c0, q0, flat = traj.center[0], traj.quaternions[0], _flat(3, False)
        elif model == 4:  # oxNA: the DNA-RNA hybrid helix, three vectors and the nucleotide types
            top, traj, _, is_rna = H.load_golden_na1("simple-helix-dna-rna")
            c0, q0, flat = traj.center[0], traj.quaternions[0], _flat_na1()
        else:
            top, c0, q0 = generators.ideal_duplex(80, model=model, seed=9)
            flat = _flat(model, model == 2)
        path = tmp_path / f"sys{model}.bin"
        with open(path, "wb") as f:
            np.array([model, top.n_nucleotides, len(top.bonded_neighbors), len(flat), 0 if box is None else 1, 20], np.int32).tofile(f)
            np.asarray(top.seq, np.int32).tofile(f)
            np.asarray(top.is_end, np.int32).tofile(f)
            np.asarray(top.bonded_neighbors, np.int32).tofile(f)
            (np.zeros(3) if box is None else box).astype(np.float64).tofile(f)
            flat.astype(np.float64).tofile(f)
            c0.astype(np.float64).tofile(f)
            q0.astype(np.float64).tofile(f)
            if is_rna is not None:
                np.asarray(is_rna, np.int32).tofile(f)
        env = {"OMP_NUM_THREADS": "4", "ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1"}
        a = subprocess.run([str(plain), str(path)], capture_output=True, text=True, env=env, timeout=120)
        b = subprocess.run([str(san), str(path)], capture_output=True, text=True, env=env, timeout=300)
        assert a.returncode == 0, a.stderr
        assert b.returncode == 0, b.stderr[-3000:]
        assert "ERROR: AddressSanitizer" not in b.stderr and "runtime error" not in b.stderr, b.stderr[-3000:]
        la, lb = a.stdout.split("\n"), b.stdout.split("\n")
        assert la[0].startswith("E ") and la[2].startswith("MD builds")
        for x, y in zip(la, lb):
            tx, ty = x.split(), y.split()
            assert len(tx) == len(ty)
            for u, v in zip(tx, ty):
                try:
                    fu, fv = float(u), float(v)
                except ValueError:
                    assert u == v
                    continue
                assert abs(fu - fv) <= 1e-7 * max(1.0, abs(fu)), (x, y)  # -O1 vs -O3 -mfma: contraction differs
        fs = [float(t) for t in la[1].split()[1:]]
        assert max(abs(t) for t in fs) < 1e-8  # Newton's third law


# ---- MARTINI (oracle/cpu_port/martini_cpu.cpp): bench.py's cpu_baseline of BASELINE configs[2] ------------------
def _martini_port(angle_kind=0, mass=None):
    from tests import martini_helpers as MH

    s = MH.system()
    port = cpu_port.MartiniCpuPort(s["types"], s["sigma"], s["eps"], s["top"].bonded_neighbors, s["bond_k"], s["bond_r0"], s["top"].angles,
                                   s["angle_k"], s["angle_t0"], angle_kind=angle_kind, mass=mass)
    return port, s


@pytest.mark.parametrize("angle_kind", [0, 1])
def test_martini_port_energies_and_forces_match_the_oracle_and_gromacs(angle_kind):
    """The port's LJ / bond / angle energies against the GROMACS energies the reference ships (its own tolerance, rtol
    1e-5) and against the torch oracle's energies and autograd forces on the same frames."""
    from oracle import martini_oracle as mo
    from tests import martini_helpers as MH

    port, s = _martini_port(angle_kind)
    x, box, e_lj = MH.frames("lj")
    _, _, e_b = MH.frames("bond")
    for f in (0, 5, 9):
        e, g = port.energy(x[f], box[f])
        if angle_kind == 0:
            np.testing.assert_allclose(e[:2], [e_lj[f], e_b[f]], rtol=1e-5)
        e_o, g_o = mo.energies_and_forces(torch.as_tensor(x[f]), torch.as_tensor(box[f]), s["types"], torch.as_tensor(s["sigma"]),
                                          torch.as_tensor(s["eps"]), s["top"].bonded_neighbors, torch.as_tensor(s["bond_k"]),
                                          torch.as_tensor(s["bond_r0"]), s["top"].angles, torch.as_tensor(s["angle_k"]),
                                          torch.as_tensor(s["angle_t0"]), angle_kind == 0)
        np.testing.assert_allclose(e, e_o.numpy(), rtol=1e-10)
        np.testing.assert_allclose(g, g_o.numpy(), rtol=0, atol=1e-9 * np.abs(g_o.numpy()).max())
    if angle_kind == 0:
        xa, ba, e_a = MH.frames("angle")
        np.testing.assert_allclose(port.energy(xa[3], ba[3])[0][2], e_a[3], rtol=1e-5)


def test_martini_port_steps_match_the_langevin_oracle():
    """Five BAOAB steps with unequal masses against oracle/martini_langevin_oracle.py on the same Philox stream, with a
    list rebuild in the middle; the result does not depend on the thread count."""
    from oracle.martini_langevin_oracle import MartiniLangevinOracle
    from tests import martini_helpers as MH

    rng = np.random.default_rng(5)
    mass = rng.uniform(40.0, 90.0, size=1280)
    port, s = _martini_port(0, mass)
    x, box, _ = MH.frames("lj")
    x0, b0 = x[3].copy(), box[3].copy()
    v0 = 0.3 * rng.standard_normal(x0.shape)
    kT = 0.0083144626 * 273.0
    outs = []
    all_threads = cpu_port.threads()
    for threads in (all_threads, 1):
        cpu_port.set_threads(threads)
        xp, vp = x0.copy(), v0.copy()
        builds, e4 = port.run(xp, vp, b0, 5, dt=0.01, kT=kT, gamma=2.0, seed=0xABCDEF012345, skin=0.25, rebuild_every=2)
        assert builds == 3
        outs.append((xp, vp, e4))
    cpu_port.set_threads(all_threads)
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    orc = MartiniLangevinOracle(s["types"], s["sigma"], s["eps"], s["top"].bonded_neighbors, s["bond_k"], s["bond_r0"], s["top"].angles,
                                s["angle_k"], s["angle_t0"], True, b0, 0.01, kT, 2.0, mass, seed=0xABCDEF012345)
    xo, vo = x0.copy(), v0.copy()
    e_ref = orc.run(xo, vo, 5)
    np.testing.assert_allclose(outs[0][0], xo, rtol=0, atol=1e-10)
    np.testing.assert_allclose(outs[0][1], vo, rtol=0, atol=1e-10)
    np.testing.assert_allclose(outs[0][2], e_ref[-1], rtol=1e-9, atol=1e-7)
