"""Positions-only trajectories: the reference's own semantics of ``run`` (mythos/simulators/jax_md/jaxmd.py:84-99 - the
scan emits ``state.position`` of EVERY step and nothing else; mythos/simulators/io.py:18-60 is what receives it).

With ``e_trace == NULL`` the step launch that produces a saved state writes it to the caller's row as well
(md_step_kernel, mythos_amd/csrc/langevin_core.inc): no energy-trace instantiation, no reduction launch, no closing launch.
Held here to three independent routes to the same states, bit for bit:

 * the frames ``store`` hands out after single-step ``advance`` calls,
 * the frames the energy-trace instantiation writes (``e_trace != NULL``) at the same steps,
 * one ``run`` of the summed length,

for oxDNA1, oxDNA2, oxRNA2 and oxNA, both precisions, the DENSE instantiation, a cadence that does not divide the call,
and through halt-and-resume (a skin that is too thin, and segments of a few launches).
"""

import numpy as np
import pytest
import torch

from mythos_amd import _lib
from mythos_amd.energy import flat_params as fp
from mythos_amd.input import defaults
from mythos_amd.utils import generators
from tests import helpers as H

pytestmark = pytest.mark.gpu

KT = 296.15 * 0.1 / 300.0
R_CUT = 3.25


def _case(model, dtype):
    """(system, centre, quaternion, dynamic list?) of a small system of the model."""
    from mythos_amd.hip_system import OxdnaSystem

    if model == 4:
        from tests.test_gpu_na1 import _system as na1_system

        top, traj, _, is_rna = H.load_golden_na1("simple-helix-dna-rna")
        s = na1_system(top, is_rna, traj.box_size, dtype)
        return s, traj.center[2], traj.quaternions[2], False
    if model == 3:
        top, traj, _, _ = H.load_golden(3, "simple-helix-12bp")
        sim, cfg = defaults.default_configs_for(H.model_dir(3))
        flat = fp.pack_flat(fp.derive_flat(3, cfg, kt=sim["kT"], salt_conc=1.0, half_charged_ends=False), _lib.param_names())
        s = OxdnaSystem(3, top.seq, top.is_end, top.bonded_neighbors, box=traj.box_size, dtype=dtype)
        s.set_params(flat)
        s.set_neighbors(top.unbonded_neighbors)
        return s, traj.center[0], traj.quaternions[0], False
    top, c0, q0 = generators.ideal_duplex(300, model=model, seed=17)
    sim, cfg = defaults.default_configs_for(f"dna{model}")
    flat = fp.pack_flat(fp.derive_flat(model, cfg, kt=sim["kT"], salt_conc=0.5, half_charged_ends=True), _lib.param_names())
    s = OxdnaSystem(model, top.seq, top.is_end, top.bonded_neighbors, box=None, dtype=dtype)
    s.set_params(flat)
    return s, c0, q0, True


def _integrator(s, dynamic, skin=0.6, every=25, seed=5):
    from mythos_amd.hip_system import LangevinIntegrator

    integ = LangevinIntegrator(s, dt=0.004, kT=KT, gamma_t=KT / 2.5, gamma_r=KT / 7.5, mass=1.0, inertia=(1.0, 1.2, 0.9), seed=seed)
    if dynamic:
        integ.set_neighbor_policy(R_CUT, skin, every)
    return integ


def _dev(a, dtype, s):
    return torch.as_tensor(a, dtype=dtype, device=s.device).contiguous()


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("model", [1, 2, 3, 4])
def test_frames_at_cadence_one_equal_single_step_stores_and_the_energy_trace_route(model, dtype, md_lanes):
    s, c0, q0, dynamic = _case(model, dtype)
    n_steps = 12

    # route A: positions only, one call, n launches, frame left open
    integ = _integrator(s, dynamic)
    integ.set_timing(1)
    c, q = _dev(c0, dtype, s), _dev(q0, dtype, s)
    p, L = integ.init_momenta()
    start = [t.clone() for t in (c, q, p, L)]
    integ.load(c, q, p, L)
    tc, tq, et = integ.advance(n_steps, save_every=1, want_energy=False)
    assert et is None and tc.shape == (n_steps, s.n, 3) and tq.shape == (n_steps, s.n, 4)
    assert integ.last_kernel_ms()["launches"] == n_steps  # no closing launch, no other instantiation
    end_a = [torch.empty_like(t) for t in start]
    integ.store(*end_a)

    # route B: single-step advances, store after each
    integ = _integrator(s, dynamic)
    c, q, p, L = (t.clone() for t in start)
    integ.load(c, q, p, L)
    for k in range(n_steps):
        integ.advance(1)
        integ.store(c, q, p, L)
        assert torch.equal(tc[k], c), (k, float((tc[k] - c).abs().max()))
        assert torch.equal(tq[k], q), k
    for a, b in zip(end_a, (c, q, p, L)):
        assert torch.equal(a, b)

    # route C: the energy-trace instantiation at the same steps
    integ = _integrator(s, dynamic)
    c, q, p, L = (t.clone() for t in start)
    integ.load(c, q, p, L)
    tc2, tq2, et2 = integ.advance(n_steps, save_every=1)
    assert et2.shape == (n_steps, 10) and torch.isfinite(et2).all()
    assert torch.equal(tc, tc2) and torch.equal(tq, tq2)

    # route D: run() of the same length (closes inside the call) with a cadence that does not divide it
    integ = _integrator(s, dynamic)
    c, q, p, L = (t.clone() for t in start)
    tc3, tq3, et3 = integ.run(c, q, p, L, n_steps, save_every=5, want_energy=False)
    assert et3 is None and tc3.shape[0] == 2
    assert torch.equal(tc3[0], tc[4]) and torch.equal(tc3[1], tc[9]) and torch.equal(tq3[1], tq[9])
    for a, b in zip(end_a, (c, q, p, L)):
        assert torch.equal(a, b)


@pytest.mark.parametrize("which", ["center", "quat"])
def test_either_row_array_alone(which):
    """traj_center or traj_quat may be NULL on their own (include/mythos_hip.h)."""
    import ctypes as C

    s, c0, q0, dynamic = _case(2, torch.float32)
    integ = _integrator(s, dynamic)
    c, q = _dev(c0, torch.float32, s), _dev(q0, torch.float32, s)
    p, L = integ.init_momenta()
    integ.load(c, q, p, L)
    start = [t.clone() for t in (c, q, p, L)]
    tc, tq, _ = integ.advance(4, save_every=2, want_energy=False)
    integ2 = _integrator(s, dynamic)
    integ2.load(*start)
    rows = torch.zeros_like(tc if which == "center" else tq)
    args = (_lib.ptr(rows), None) if which == "center" else (None, _lib.ptr(rows))
    rc = integ2._lib.mythos_langevin_advance(integ2._h, 4, 2, *args, None, C.c_void_p(torch.cuda.current_stream(s.device).cuda_stream))
    assert rc == 0, _lib.last_error()
    torch.cuda.synchronize()
    assert torch.equal(rows, tc if which == "center" else tq)
    if which == "quat":  # a quaternion row that is not aligned to four elements is refused, not written with a wide store
        odd = torch.zeros(rows.numel() + 1, dtype=rows.dtype, device=rows.device)[1:]
        rc = integ2._lib.mythos_langevin_advance(integ2._h, 4, 2, None, _lib.ptr(odd), None, C.c_void_p(torch.cuda.current_stream(s.device).cuda_stream))
        assert rc != 0 and "aligned" in _lib.last_error()


@pytest.mark.parametrize("model", [1, 2])
def test_dense_instantiation_writes_the_same_rows_as_its_single_step_stores(model):
    """(the DENSE fp32 instantiation is a second compilation of the arithmetic: compared with itself)"""
    s, c0, q0, dynamic = _case(model, torch.float32)
    try:
        _lib.debug_set("md_dense", 1)
        integ = _integrator(s, dynamic)
        c, q = _dev(c0, torch.float32, s), _dev(q0, torch.float32, s)
        p, L = integ.init_momenta()
        start = [t.clone() for t in (c, q, p, L)]
        integ.load(c, q, p, L)
        tc, tq, _ = integ.advance(8, save_every=1, want_energy=False)
        integ = _integrator(s, dynamic)
        c, q, p, L = (t.clone() for t in start)
        integ.load(c, q, p, L)
        for k in range(8):
            integ.advance(1)
            integ.store(c, q, p, L)
            assert torch.equal(tc[k], c) and torch.equal(tq[k], q), k
    finally:
        _lib.debug_set("md_dense", 0)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_rows_survive_halt_and_resume(dtype):
    """A skin far too thin for its rebuild interval, queued in segments of five launches: sites leave their skins, the
    launches behind the halt return at once, the host rebuilds and resumes.  Every row must be the state the
    uninterrupted route produces at that step: single-step advances with a store after each (which halt and resume on
    their own schedule: the list differs in what it lists beyond the cut-off, never in what acts)."""
    s, c0, q0, _ = _case(2, dtype)
    n_steps = 60
    integ = _integrator(s, True, skin=0.06, every=1000)
    c, q = _dev(c0, dtype, s), _dev(q0, dtype, s)
    p, L = integ.init_momenta()
    start = [t.clone() for t in (c, q, p, L)]
    integ.load(c, q, p, L)
    _lib.debug_set("md_segment", 5)
    try:
        tc, tq, _ = integ.advance(n_steps, save_every=1, want_energy=False)
    finally:
        _lib.debug_set("md_segment", 0)
    assert integ.last_recoveries() >= 2, integ.last_recoveries()
    integ.store(c, q, p, L)
    assert torch.equal(tc[-1], c) and torch.equal(tq[-1], q)
    # rows of the interrupted run are bitwise the states of the same run taken step by step (a halt is raised by the launch
    # that moves a site out of its skin and the list is rebuilt at the state that launch left, however many launches
    # were queued behind it: both routes rebuild at the same states)
    s3, _, _, _ = _case(2, dtype)
    integ3 = _integrator(s3, True, skin=0.06, every=1000)
    c3, q3, p3, L3 = (t.clone() for t in start)
    integ3.load(c3, q3, p3, L3)
    recoveries = 0
    for k in range(n_steps):
        integ3.advance(1)
        recoveries += integ3.last_recoveries()
        integ3.store(c3, q3, p3, L3)
        assert torch.equal(tc[k], c3) and torch.equal(tq[k], q3), k
    assert recoveries >= 2
    # ... and, to summation order, the trajectory on a static all-pairs list
    top, _, _ = generators.ideal_duplex(300, model=2, seed=17)
    s2, _, _, _ = _case(2, dtype)
    s2.set_neighbors(top.unbonded_neighbors)
    integ2 = _integrator(s2, False)
    c2, q2, p2, L2 = (t.clone() for t in start)
    integ2.load(c2, q2, p2, L2)
    tc2, tq2, _ = integ2.advance(n_steps, save_every=1, want_energy=False)
    tol = 1e-9 if dtype == torch.float64 else 2e-3
    assert float((tc - tc2).abs().max()) <= tol and float((tq - tq2).abs().max()) <= tol


def test_simulator_default_is_positions_only_and_energies_come_from_map():
    """HipMDSimulator.run stores every step's state like the reference's run and carries no energies; the energies of the
    stored states come from energy_fn.map, as in the reference (DiffTRe: objective.py:224-235)."""
    from mythos_amd.energy import dna2
    from mythos_amd.energy.base import Quaternion, RigidBody, space
    from mythos_amd.simulators.hip_md import HipMDSimulator, StaticSimulatorParams
    from mythos_amd.simulators.neighbors import NoNeighborList

    top, traj, _, _ = H.load_golden(2, "simple-helix")
    ef = dna2.create_default_energy_fn(top, space.periodic(traj.box_size)[0])
    sp = StaticSimulatorParams(seq=top.seq, mass=(1.0, (1.0, 1.0, 1.0)), gamma=(KT / 2.5, KT / 7.5), bonded_neighbors=top.bonded_neighbors,
                               checkpoint_every=0, dt=0.005, kT=KT)
    init = RigidBody(center=torch.as_tensor(traj.center[0], device="cuda"), orientation=Quaternion(vec=torch.as_tensor(traj.quaternions[0], device="cuda")))
    outs = {}
    for trace in (False, True):
        sim = HipMDSimulator(energy_fn=ef, simulator_params=sp, neighbors=NoNeighborList(top.unbonded_neighbors), dtype=torch.float64,
                             trace_energy=trace)
        out = sim.run({}, init, 25, 3)
        tr = out.observables[0]
        assert tr.center.shape == (25, top.n_nucleotides, 3)
        assert (tr.metadata is None) == (not trace)
        outs[trace] = tr
    assert torch.equal(outs[False].center, outs[True].center) and torch.equal(outs[False].orientation.vec, outs[True].orientation.vec)
    u = ef.map(RigidBody(center=outs[False].center, orientation=outs[False].orientation))
    torch.testing.assert_close(u.cpu(), outs[True].metadata["energy_terms"].sum(1).cpu(), rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("neighbors", ["static", "verlet"])
def test_consecutive_runs_on_one_simulator_equal_fresh_simulators(neighbors):
    """HipMDSimulator keeps its system handle and integrator between calls (the reference re-traces
    nothing when only opt_params change, jaxmd.py:60-68).  Two runs on one simulator - the parameters replaced in between,
    a new key, the second one starting where the first ended - are bit for bit the two runs on fresh simulators."""
    from mythos_amd.energy import dna2
    from mythos_amd.energy.base import Quaternion, RigidBody, space
    from mythos_amd.simulators.hip_md import HipMDSimulator, StaticSimulatorParams
    from mythos_amd.simulators.neighbors import NoNeighborList, VerletNeighborList

    top, traj, _, _ = H.load_golden(2, "simple-helix")
    ef = dna2.create_default_energy_fn(top, space.periodic(traj.box_size)[0])
    sp = StaticSimulatorParams(seq=top.seq, mass=(1.0, (1.0, 1.0, 1.0)), gamma=(KT / 2.5, KT / 7.5), bonded_neighbors=top.bonded_neighbors,
                               checkpoint_every=0, dt=0.005, kT=KT)
    nb = NoNeighborList(top.unbonded_neighbors) if neighbors == "static" else VerletNeighborList(3.25, 0.6, 10)
    init = RigidBody(center=torch.as_tensor(traj.center[0], device="cuda"), orientation=Quaternion(vec=torch.as_tensor(traj.quaternions[0], device="cuda")))
    pa, pb = {}, {"eps_stack_base": 1.30, "a_hb": 7.5, "eps_backbone": 2.1}

    def make():
        return HipMDSimulator(energy_fn=ef, simulator_params=sp, neighbors=nb, dtype=torch.float64)

    one = make()
    o1 = one.run(pa, init, 30, 3)
    o2 = one.run(pb, o1.state["final_state"], 30, 4)
    o3 = one.run(pa, init, 0, 9)  # nothing to step: no frames, the state comes back as it went in
    assert len(one._resident) == 1
    assert o3.observables[0].center.shape[0] == 0 and torch.equal(o3.state["final_state"].center, init.center.to(torch.float64))
    f1 = make().run(pa, init, 30, 3)
    f2 = make().run(pb, o1.state["final_state"], 30, 4)
    for a, b in ((o1, f1), (o2, f2)):
        assert torch.equal(a.observables[0].center, b.observables[0].center)
        assert torch.equal(a.observables[0].orientation.vec, b.observables[0].orientation.vec)
        assert torch.equal(a.state["final_state"].center, b.state["final_state"].center)
        assert torch.equal(a.state["momentum"][0], b.state["momentum"][0])
    assert not torch.equal(o1.observables[0].center[-1], f2.observables[0].center[-1])
    one.release()
    assert len(one._resident) == 0
    import pickle

    again = pickle.loads(pickle.dumps(dc_replace_name(one)))
    assert again._resident == {}


def dc_replace_name(sim):
    """(a simulator whose energy function is picklable: none of the device handles travels)"""
    import dataclasses as dc

    return dc.replace(sim, energy_fn=None, neighbors=None)
