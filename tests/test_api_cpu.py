"""CPU-side tests of the drop-in surface: configurations, composed-function bookkeeping, the C ABI
symbol table and the replica-sharding collectives (gloo, world_size 2)."""

import os
import re
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

from mythos_amd import _lib
from mythos_amd.energy import dna1, dna2
from mythos_amd.energy.configuration import BaseConfiguration
from mythos_amd.input import topology
from mythos_amd.optimization import objective
from tests import helpers as H

ROOT = Path(__file__).resolve().parent.parent


def test_library_exports_every_declared_symbol():
    header = (ROOT / "include" / "mythos_hip.h").read_text()
    declared = set(re.findall(r"\b(mythos_[a-z0-9_]+)\s*\(", header))
    assert declared, "no prototypes found in include/mythos_hip.h"
    lib = _lib.load()
    missing = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert not missing, missing
    assert declared == set(_lib.DECLARED_SYMBOLS)
    assert lib.mythos_version().decode().startswith("mythos_amd")
    assert lib.mythos_oxdna_param_name(-1) is None


def test_no_kernel_holds_vector_work_in_a_block_entered_only_with_exec_zero():
    """A compiler fault met in round 4 (oxdna_energy_core.inc, MYTHOS_EN_PARK_FROM): spill reloads placed where no lane
    is enabled.  The scan runs over the disassembly of every gfx950 code object of the library that ships."""
    sys.path.insert(0, str(ROOT / "scripts"))
    try:
        import check_exec0_reloads as scan
    finally:
        sys.path.pop(0)
    if not Path(scan.OBJDUMP).exists():
        pytest.skip("llvm-objdump of the ROCm toolchain is not here")
    objects = list(scan.code_objects(_lib.lib_path()))
    assert len(objects) >= 8, "the library's offload bundles were not found"
    hits = scan.scan_library(_lib.lib_path())
    assert not hits, hits[:6]
    # the rule itself, on a listing in the shape the fault had
    listing = """_Zkernel:
\ts_and_saveexec_b64 s[2:3], vcc
\ts_cbranch_execz .LBB0_3
.LBB0_1:
\tv_add_u32_e32 v1, 8, v1
\ts_andn2_b64 exec, exec, s[4:5]
\ts_cbranch_execz .LBB0_3
\ts_branch .LBB0_1
.LBB0_3:
\tscratch_load_dword v130, off, off offset:44
.LBB0_4:
\ts_or_b64 exec, exec, s[2:3]
\ts_endpgm
"""
    tmp = ROOT / "build" / "exec0_listing.s"
    tmp.parent.mkdir(exist_ok=True)
    tmp.write_text(listing)
    found = scan.scan(str(tmp))
    assert len(found) == 1 and "scratch_load_dword v130" in found[0][3]


def test_the_kernels_with_scratch_are_the_known_ones():
    """Both wrong results of round 4 went with spill code at a register bound: the fp64 forces mode of the energy kernel
    (reloads under EXEC = 0) and the 16-lane oxNA fp64 step kernel (forces off by 2.6e-5 after an unrelated change moved the
    allocation).  Both are one compiler fault - an end-of-region EXEC restore removed as redundant, spill code placed
    behind it - and the whole library is compiled with -mllvm -amdgpu-remove-redundant-endcf=0 since.  What remains is
    bookkeeping: the kernels that use scratch are the oxDNA1 fp32 step kernels (20 B), the oxNA and oxRNA2-pseq fp64 step
    kernels (<= 88 B) and energy-kernel instantiations the parity tests run (segmented rows, the fp64 forces mode at three
    workgroups per CU, oxRNA2 / oxNA gradients); no plain oxDNA / oxRNA2 fp64 step kernel, no MARTINI kernel, no list
    builder.  A kernel that newly appears here is a decision to take with the GPU suite in hand."""
    sys.path.insert(0, str(ROOT / "scripts"))
    try:
        import check_exec0_reloads as scan
    finally:
        sys.path.pop(0)
    if not Path(scan.READELF).exists():
        pytest.skip("llvm-readelf of the ROCm toolchain is not here")
    rows = scan.kernels_with_scratch(_lib.lib_path())
    assert not [k for k in rows if "martini" in k or "build_rows" in k or "cell_" in k]
    f64_steps = {k: v for k, v in rows.items() if "md_step_kernelId" in k}
    # <double, MODEL, SAVE, ITEMS, PSEQ, DENSE, GL>: model 4, or model 3 under a probabilistic sequence
    assert all(("IdLi4E" in k or ("IdLi3E" in k and "ELb1ELb0ELi" in k)) and v <= 88 for k, v in f64_steps.items()), f64_steps
    f32_steps = {k: v for k, v in rows.items() if "md_step_kernelIf" in k}
    assert len(f32_steps) <= 3 and all(v <= 20 and "IfLi1E" in k for k, v in f32_steps.items()), f32_steps
    energy = {k: v for k, v in rows.items() if "oxdna_energy_kernel" in k}
    assert len(rows) == len(f32_steps) + len(f64_steps) + len(energy) and len(energy) <= 29 and max(energy.values()) <= 116, (len(energy), max(energy.values()))
    flags = (ROOT / "mythos_amd" / "csrc" / "Makefile").read_text()
    assert "-amdgpu-remove-redundant-endcf=0" in flags.split("CXXFLAGS ?=")[1].split("\n\n")[0]


def test_no_gpu_means_loud_failure():
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from mythos_amd.hip_system import OxdnaSystem

    with pytest.raises(_lib.MythosHipError):
        OxdnaSystem(2, [0, 1, 2, 3], None, [[0, 1], [1, 2], [2, 3]])


def test_configuration_semantics():
    cfg = dna2.FeneConfiguration.from_dict({"eps_backbone": 2.0, "r0_backbone": 0.7564, "delta_backbone": 0.25, "fmax": 500.0, "finf": 4.0}, ("*",))
    assert set(cfg.opt_params) == {"eps_backbone", "r0_backbone", "delta_backbone", "fmax", "finf"}
    with pytest.raises(ValueError, match="not initialized"):
        dna2.FeneConfiguration(eps_backbone=2.0)
    with pytest.raises(ValueError, match="permitted for optimization"):
        dna2.BondedExcludedVolumeConfiguration.from_dict(
            {"eps_exc": 2.0, "dr_star_base": 0.32, "sigma_base": 0.33, "sigma_back_base": 0.515, "sigma_base_back": 0.515,
             "dr_star_back_base": 0.5, "dr_star_base_back": 0.5}, ("b_base",))
    merged = cfg | {"fmax": 400.0}
    assert merged.fmax == 400.0 and cfg.fmax == 500.0
    assert isinstance(merged, BaseConfiguration)
    d = cfg.to_dictionary(include_dependent=True, exclude_non_optimizable=False)
    assert list(d) == list(dna2.FeneConfiguration.required_params)


def test_composed_namespace_and_dependents():
    top = topology.from_oxdna_file(H.GOLDEN / "dna2" / "simple-helix" / "generated.top")
    ef = dna2.create_default_energy_fn(top)
    assert len(ef.opt_params()) == 103  # 110 entries, 103 unique names
    assert "kt" not in ef.opt_params()
    # shared names reach every term that has them (energy/base.py:278-299)
    ef2 = ef.with_params(kt=0.11, eps_exc=2.5)
    assert float(ef2.energy_fns[2].params.kt) == 0.11 and float(ef2.energy_fns[7].params.kt) == 0.11
    assert float(ef2.energy_fns[1].params.eps_exc) == 2.5 and float(ef2.energy_fns[3].params.eps_exc) == 2.5
    with pytest.raises(ValueError, match="not used"):
        ef.with_params(does_not_exist=1.0)
    pd = ef.params_dict()
    assert float(pd["b_low_stack"]) == pytest.approx(-68.1857, rel=1e-5)  # model.h STCK_BLOW
    assert "b_low_coax" not in pd  # dna2 coaxial dependents are not declared (dna2/coaxial_stacking.py:56-76)
    noopt = ef.with_noopt("a_hb")
    assert "a_hb" not in noopt.opt_params() and "a_hb" in ef.opt_params()
    assert len(ef.without_terms("Debye").energy_fns) == 7
    assert len(dna1.create_default_energy_fn(top).energy_fns) == 7


def test_difftre_weights_known_answer():
    """mythos/optimization/tests/test_objective.py:188-204."""
    w, neff = objective.compute_weights_and_neff(1, np.array([1.0, 2.0, 3.0]), np.array([1.0, 2.0, 3.0]))
    assert np.allclose(w.numpy(), [1 / 3, 1 / 3, 1 / 3]) and np.allclose(float(neff), 1.0)
    t = torch.tensor([0.1, 0.1, 0.2, 0.2])
    e = torch.tensor([1.0, 2.0, 3.0, 4.0], dtype=torch.float64)
    assert objective.compute_min_segment_neff(t, e, e) == pytest.approx(1.0)


_WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["MYTHOS_ROOT"])
import torch, torch.distributed as dist
from mythos_amd import distributed as md
from mythos_amd.optimization import objective
rank, world, _ = md.init("gloo")
assert world == 2
ids = md.shard_replicas(5, rank, world)
assert ids == ([0, 2, 4] if rank == 0 else [1, 3])
local = torch.tensor([[float(i), 10.0 * i] for i in ids], dtype=torch.float64)
allv = md.all_gather_observables(local)
assert allv.shape == (5, 2) and torch.equal(allv[:, 0], torch.arange(5, dtype=torch.float64)), allv
# the job's replica count known: one collective, no host read-back - ragged (5 over 2 ranks) and equal (4 over 2) shares
assert torch.equal(md.all_gather_observables(local, n_total=5), allv)
ids4 = md.shard_replicas(4, rank, world)
loc4 = torch.tensor([[float(i), 10.0 * i, -float(i)] for i in ids4], dtype=torch.float64)
all4 = md.all_gather_observables(loc4, n_total=4)
assert all4.shape == (4, 3) and torch.equal(all4[:, 0], torch.arange(4, dtype=torch.float64)) and torch.equal(all4[:, 2], -all4[:, 0]), all4
try:
    md.all_gather_observables(loc4, n_total=7)
    raise SystemExit("a share that does not match n_total must be refused")
except ValueError:
    pass
g = torch.Generator().manual_seed(7)
e_new = torch.randn(10, generator=g, dtype=torch.float64) * 3
e_ref = torch.randn(10, generator=g, dtype=torch.float64) * 3
w_all, neff_all = objective.compute_weights_and_neff(0.7, e_new, e_ref)
mine = slice(0, 6) if rank == 0 else slice(6, 10)
w, neff = objective.distributed_weights_and_neff(0.7, e_new[mine], e_ref[mine])
assert torch.allclose(w, w_all[mine], rtol=1e-12), (w, w_all[mine])
assert abs(float(neff) - float(neff_all)) < 1e-12
s = md.all_reduce_sum(torch.tensor([float(rank + 1)]))
assert float(s) == 3.0
assert not w.requires_grad  # the all-reduced normaliser is not in any graph: weights come back detached

# ---- DiffTRe across ranks: the gradient of a loss of a reweighted mean, frames sharded r mod world, equals the
#      single-process autograd value on all frames (a toy energy stands in for the HIP energy function: the host
#      logic under test only needs map() to be differentiable in the parameters)
class ToyEnergy:
    def __init__(self, f, h, p=None):
        self.f, self.h, self.p = f, h, p or {}
    def with_params(self, *ds, **kw):
        p = dict(self.p)
        for d in ds:
            p.update(d)
        p.update(kw)
        return ToyEnergy(self.f, self.h, p)
    def map(self, states):
        a, b, c = (torch.as_tensor(self.p[k], dtype=torch.float64) for k in ("a", "b", "c"))
        return a * self.f[states] + b * b * self.h[states] + torch.sin(c) * self.f[states] * self.h[states]

S = 64 * 3  # 64 replicas x 3 stored frames
g = torch.Generator().manual_seed(11)
f = torch.randn(S, generator=g, dtype=torch.float64)
h = torch.randn(S, generator=g, dtype=torch.float64)
obs_all = torch.randn(S, generator=g, dtype=torch.float64) + 20.0
temps = torch.where(torch.arange(S) % 2 == 0, 0.09, 0.11).to(torch.float64)  # two temperature segments
beta_all = 1.0 / temps
ref_p, new_p = {"a": 0.3, "b": -0.7, "c": 0.2}, {"a": 0.35, "b": -0.66, "c": 0.25}
ef = ToyEnergy(f, h)
every = torch.arange(S)
ref_e_all = ef.with_params(ref_p).map(every)
target = 20.3
def loss_fn(ref_states, weights, energy_fn, opt_params, observables):
    m = (weights * obs_all[ref_states]).sum()
    return (m - target) ** 2, (("obs", m.detach()), {})
(l_one, (neff_one, meas, _)), g_one = objective.compute_loss_and_grad(new_p, ef, beta_all, loss_fn, every, ref_e_all, [])
replicas = md.shard_replicas(64, rank, world)
mine = torch.cat([torch.arange(3 * r, 3 * r + 3) for r in replicas])
(l_d, (neff_d, mean_d, e_loc)), g_d = objective.distributed_compute_loss_and_grad(
    new_p, ef, beta_all[mine], lambda st: obs_all[st], lambda m: (m - target) ** 2, mine, ref_e_all[mine])
assert abs(float(l_d) - float(l_one)) <= 1e-12 * max(1.0, abs(float(l_one))), (float(l_d), float(l_one))
assert abs(float(neff_d) - float(neff_one)) <= 1e-12
assert abs(float(mean_d) - float(meas[1])) <= 1e-12 * abs(float(meas[1]))
for k in new_p:
    assert abs(float(g_d[k]) - float(g_one[k])) <= 1e-12 * max(1.0, abs(float(g_one[k]))), (k, float(g_d[k]), float(g_one[k]))
assert e_loc.shape == mine.shape
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_replica_sharding_collectives_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MYTHOS_ROOT=str(ROOT), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2")
    procs = [
        subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        for r in range(2)
    ]
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


def test_adam_matches_the_published_update_rule():
    """optax.adam semantics: bias-corrected moments, update = -lr m_hat / (sqrt(v_hat) + eps)."""
    from mythos_amd.optimization.optimization import Adam, apply_updates

    opt = Adam(learning_rate=0.1, b1=0.9, b2=0.999, eps=1e-8)
    params = {"a": 1.0, "b": torch.tensor(-2.0, dtype=torch.float64)}
    state = opt.init(params)
    g1 = {"a": torch.tensor(0.5, dtype=torch.float64), "b": torch.tensor(-4.0, dtype=torch.float64)}
    upd, state = opt.update(g1, state, params)
    # first step: m_hat = g, v_hat = g^2  ->  update = -lr * sign(g) (up to eps)
    assert abs(float(upd["a"]) + 0.1) < 1e-7 and abs(float(upd["b"]) - 0.1) < 1e-7
    params = apply_updates(params, upd)
    g2 = {"a": torch.tensor(0.25, dtype=torch.float64), "b": torch.tensor(0.0, dtype=torch.float64)}
    upd, state = opt.update(g2, state, params)
    m = 0.9 * 0.05 + 0.1 * 0.25
    v = 0.999 * (0.001 * 0.25) + 0.001 * 0.0625
    want = -0.1 * (m / (1 - 0.9**2)) / ((v / (1 - 0.999**2)) ** 0.5 + 1e-8)
    assert abs(float(upd["a"]) - want) < 1e-12 and state["count"] == 2


def test_objective_protocol_without_a_gpu():
    from mythos_amd.optimization.objective import DiffTReObjective, Objective

    obj = Objective(name="o", required_observables=("x", "y"), logging_observables=("x",),
                    grad_or_loss_fn=lambda x, y: ({"p": x + y}, [("sum", x + y)]))
    miss = obj.calculate({"x": 1.0})
    assert not miss.is_ready and miss.needs_update == ("y",)
    out = obj.calculate({"x": 1.0, "y": 2.0})
    assert out.is_ready and out.grads == {"p": 3.0} and out.observables == {"sum": 3.0, "x": 1.0, "y": 2.0}
    assert obj.get_logging_observables(out.observables) == [("x", 1.0)]
    with pytest.raises(ValueError, match="n_equilibration_steps"):
        DiffTReObjective(name="d", required_observables=("t",), grad_or_loss_fn=lambda *a: None, energy_fn=object(),
                         n_equilibration_steps=-1)
    with pytest.raises(ValueError, match="max_valid_opt_steps"):
        DiffTReObjective(name="d", required_observables=("t",), grad_or_loss_fn=lambda *a: None, energy_fn=object(),
                         max_valid_opt_steps=0)
    spent = DiffTReObjective(name="d", required_observables=("t",), grad_or_loss_fn=lambda *a: None, energy_fn=object(),
                             max_valid_opt_steps=2).calculate({}, opt_params={}, opt_steps=2)
    assert not spent.is_ready and spent.needs_update == ("t",) and spent.state == {"opt_steps": 0}


def test_bench_strings_fit_the_drivers_parsed_record():
    """The driver keeps 120 characters of a string: workload, timed_region and the CPU baseline's sample must say what
    was measured within that."""
    import argparse
    import importlib.util
    from pathlib import Path

    spec = importlib.util.spec_from_file_location("bench_module", Path(__file__).resolve().parent.parent / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for steps, save_every, n in ((20, 0, 21), (2000, 0, 5), (100000, 1000, 5)):
        args = argparse.Namespace(steps=steps, save_every=save_every, dtype="f32", trace_energy=save_every > 0)
        m = {"rebuilds_total": n * (steps // 50), "samples_ms": [0.3] * n}
        txt = bench._timed_region(args, m, {"steps_per_s": 40714.2})
        assert len(txt) <= 120 and f"{n} x {steps} steps" in txt and "f64 40.7k" in txt and "steps / total time" in txt
        launches = n * (steps + (1 if save_every and steps % save_every == 0 else 0))
        assert f"= {launches} launches" in txt and f"+ {n * (steps // 50)} list rebuilds" in txt


def test_units_as_the_reference_defines_them():
    """mythos_amd.utils.units mirrors mythos/utils/units.py:5-38; the defaults of the package are at its 296.15 K."""
    from mythos_amd.input import defaults
    from mythos_amd.utils import units

    assert units.NM_PER_OXDNA_LENGTH == pytest.approx(0.8518) and units.PN_PER_OXDNA_FORCE == 48.63 and units.JOULES_PER_OXDNA_ENERGY == 4.142e-20
    assert units.get_kt(300.0) == pytest.approx(0.1) and units.get_kt_from_c(23.0) == pytest.approx(units.get_kt(296.15))
    assert units.get_kt_from_string("296.15K") == pytest.approx(0.0987166666) and units.get_kt_from_string("23C") == pytest.approx(units.get_kt(296.15))
    assert units.from_kt(units.get_kt(310.0)) == pytest.approx(310.0)
    assert np.allclose(units.get_kt(np.array([300.0, 330.0])), [0.1, 0.11])
    with pytest.raises(ValueError, match="Invalid temperature string"):
        units.get_kt_from_string("300")
    sim, _ = defaults.default_configs_for("dna2")
    assert sim["kT"] == pytest.approx(units.get_kt(296.15))
