#!/usr/bin/env python3
"""Headline benchmark: MD steps/s of an oxDNA2 12 kbp duplex (24 000 nucleotides, Debye-Hueckel,
Langevin) per GPU, one independent replica per GPU (BASELINE.json metric / configs[3]).

    python bench.py --gpus 1 --steps 2000 --warmup 200
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...      (no launcher: starts the N ranks itself as child processes)

A "step" is one fused force + BAOAB kernel over the whole system (mythos_amd/csrc/langevin_core.inc).
The timed region is one ``mythos_langevin_advance`` call of exactly K steps with the state resident in HBM
(loaded into the integrator before the clock starts), bracketed by barrier + synchronize, MAX over ranks.
It is measured ``--repeats`` times back to back (the trajectory continues from sample to sample; default 5, or 21
when K < 500).  **value = N_gpus * repeats * K / the SUM of the samples**: every scheduled list rebuild that falls into
the repeats is paid for in it (round 3 reported the median sample, which for K = 20 never contains a rebuild: ADVICE r3);
the median rate is ``steps_per_s_median``, all samples are in ``config.samples_ms`` and the scheduled rebuilds inside
each in ``config.scheduled_rebuilds_per_sample``.  A GPU that sat idle until the clock starts runs its first
milliseconds at clocks still coming up; the untimed warm-up is therefore at least W steps AND at least
``--min-warmup-ms`` of stepping (``config.warmup_steps_done``).  Nothing else is inside a sample: the per-dispatch HIP
events behind ``roofline.kernel_ms`` are taken in a second, untimed pass.  For N > 1 every rank steps its own replica
(weak scaling; the MD data path has no collective): barrier + synchronize, the rank's clock around its K steps,
synchronize + barrier, MAX over ranks.  ``--save-every S`` stores positions + quaternions every S steps inside the
timed region (the reference's run stores every step, jaxmd.py:84-99); with ``--trace-energy`` also the energies, and
for N > 1 those are all-gathered over RCCL (one collective).  The headline precision is fp32 (north_star: fp32 forces
at 1e-3); the same measurement in the reference's fp64 is ``f64_steps_per_s`` / ``f64_frac`` / ``f64_kernel_ms`` and
``config.f64``.

Rank 0 prints ONE JSON line with the fields of the driver contract plus
  roofline      HBM roofline of the step kernel: algorithmic bytes per launch (2 x 14 state
                words + 13 B topology + 4 B per neighbour entry, per nucleotide) / the kernel's mean duration,
                measured with HIP events on the launch stream.  ``traffic`` is the PMC-measured HBM-side byte count per
                launch (2 x FETCH_SIZE + WRITE_SIZE, gfx950 correction) of the SAME command, read from
                profiles/traffic.json when that file matches the workload.
  cpu_baseline  the C++/OpenMP port of the same kernels (oracle/cpu_port, fp64, all cores the process may use; median of
                5 x <= 1 000 steps) and, under ``torch_restatement``, the vectorised torch-fp64 oracle that stands in for
                JAX-CPU, both on a bounded sample of the same system (N=1, rank 0 only)
  config.save_every_1   the same K steps with every step's positions + quaternions stored (the reference's semantics of
                run), both precisions: steps/s, its ratio to the no-output rate, bytes written per step
  secondary     (N = 1) the other BASELINE configs, each with its own kernel time, algorithmic bytes and HBM fraction:
                cfg1 oxDNA2 1 kbp (fp32 steps/s + the fp64 energy check), cfg2 MARTINI 20 480 beads (with its own
                cpu_baseline), cfg4 DiffTRe: U and dU/dtheta frames/s at 6 400 x 64 nt in both precisions and ONE
                end-to-end iteration (64 replicas x 2 000 steps -> map -> grad -> Adam) with its host-side share
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

from mythos_amd import _lib  # noqa: E402
from mythos_amd.energy import flat_params as fp  # noqa: E402
from mythos_amd.hip_system import LangevinIntegrator, OxdnaSystem  # noqa: E402
from mythos_amd.input import defaults  # noqa: E402
from mythos_amd.utils import generators  # noqa: E402
from mythos_amd.utils.units import PS_PER_OXDNA_TIME as OXDNA_TIME_UNIT_PS  # noqa: E402  (3.03: oxDNA simulation time unit)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
R_CUT = 3.25  # largest centre-centre interaction range: Debye r_cut 2.2867 + 2 * |backbone offset| 0.4814


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--bp", type=int, default=12000, help="base pairs of the duplex (12000 = headline config)")
    ap.add_argument("--dtype", choices=["f32", "f64"], default="f32")
    ap.add_argument("--skin", type=float, default=None, help="Verlet skin (default: 0.9 oxDNA length units / 0.5 nm MARTINI)")
    ap.add_argument("--rebuild-every", type=int, default=None, help="steps between list rebuilds (default: 50 oxDNA / 12 MARTINI)")
    ap.add_argument("--inner-list", type=str, default=None,
                    help="MARTINI workload: MARGIN,EVERY of the pruned rows (default 0.2,4; 0,0 switches them off)")
    ap.add_argument("--dt", type=float, default=0.005)
    ap.add_argument("--save-every", type=int, default=0, help="observable cadence inside the timed region (N>1: all-gathered)")
    ap.add_argument("--trace-energy", action="store_true",
                    help="with --save-every: also the 8 term + 2 kinetic energies of the saved steps (the energy-trace instantiation); "
                         "without it the saved steps are positions only, the reference's run (jaxmd.py:84-99)")
    ap.add_argument("--cpu-steps", type=int, default=-1, help="CPU-baseline sample size in steps (-1: auto, 0: skip)")
    ap.add_argument("--workload", choices=["oxdna2-12kbp", "martini-bilayer"], default="oxdna2-12kbp",
                    help="oxdna2-12kbp is the headline (BASELINE.json metric); martini-bilayer is BASELINE configs[2] "
                         "(20 480-bead DMPC bilayer, LJ + bonds + angles, Langevin) and prints its own line")
    ap.add_argument("--no-second-dtype", action="store_true", help="skip the measurement at the other precision (config.f64 / config.f32)")
    ap.add_argument("--instrument-steps", type=int, default=512,
                    help="least length of the untimed, event-instrumented pass behind the timed region (dev: short runs of wrong-physics bounds)")
    ap.add_argument("--repeats", type=int, default=None,
                    help="the timed region (exactly --steps steps) is run this many times back to back, each bracketed by barrier + "
                         "synchronize; value = steps / the MEDIAN sample, every sample is in config.samples_ms "
                         "(default 5; 21 for timed regions under 500 steps)")
    ap.add_argument("--min-warmup-ms", type=float, default=150.0,
                    help="the untimed warm-up lasts at least --warmup steps and at least this long (GPU clocks ramp for the "
                         "first milliseconds after idling)")
    ap.add_argument("--debug-set", action="append", default=[], metavar="KEY=VALUE",
                    help="dev A/B: mythos_debug_set switches (mythos_amd._lib.DEBUG_KEYS), e.g. mm_subcells=1")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary configs (cfg1 / cfg2 / cfg4) and save_every_1")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="development aid: all ranks share cuda:0 and meet over gloo (a 1-GPU box cannot run RCCL "
                         "between ranks); the printed line is then marked as a rehearsal, not a measurement")
    args = ap.parse_args()
    if args.repeats is None:
        args.repeats = 5 if args.steps >= 500 else 21
    return args


def algorithmic_bytes_per_step(n: int, nbar: float, word: int) -> float:
    """read+write {center 3, quat 4, p 3, L 4} words, 13 B topology,
    4 B per directed neighbour entry."""
    return n * (2 * 14 * word + 13 + 4.0 * nbar)


def measured_traffic(dtype_name: str, n: int):
    """HBM-side bytes per md_step_kernel launch from the committed PMC passes of this command
    (profiles/traffic.json for fp32, profiles/traffic_f64.json for fp64, written by scripts/collect_traffic.py), or None
    if there is none for this workload."""
    f = ROOT / "profiles" / ("traffic.json" if dtype_name == "f32" else "traffic_f64.json")
    if not f.exists():
        return None
    t = json.loads(f.read_text())
    if t.get("n_nucleotides") != n or t.get("dtype") != dtype_name:
        return None
    return t["hbm_bytes_per_launch"]


def host_cores() -> int:
    """Cores this process may really use: affinity mask, capped by the cgroup CPU quota (a GPU box hands
    a 1-GPU job a share of a much larger host; oversubscribing it makes the baseline meaningless).  ``nproc`` in the
    output line is os.cpu_count(), the host's total."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    cap = os.environ.get("MYTHOS_BENCH_CPU_CORES")  # (dev: a fixed thread count for A/B runs; default = all the process may use)
    return max(1, min(n, int(cap))) if cap else max(1, n)


def param_gradient_error(dev) -> dict:
    """Second half of BASELINE.json's metric: max-abs error of dU/dtheta against the CPU reference path.

    oxDNA2 16-nt golden duplex (tests/golden/dna2/simple-helix), three frames: the HIP kernel's dU/dflat carried to
    the reference's optimisable parameters (110 section entries, 103 unique names) by the chain rule, against torch autograd of the fp64 oracle through the
    same derivation of the dependent constants.  Reported relative to the largest gradient component."""
    from oracle import oxdna_oracle as orc
    from tests import helpers as H

    top, traj, _, _ = H.load_golden(2, "simple-helix")
    sim, cfg = defaults.default_configs_for("dna2")
    leaves, sections = {}, {}
    for sec, d in cfg.items():
        sections[sec] = {}
        for k, v in d.items():
            # the reference's optimisable set: every energy-section key (geometry is fixed, dna2/__init__.py:45-46)
            if sec != "geometry" and isinstance(v, (int, float)) and not isinstance(v, bool) and k not in ("kt", "salt_conc", "half_charged_ends"):
                leaf = torch.tensor(float(v), dtype=torch.float64, requires_grad=True)
                leaves[(sec, k)] = leaf
                sections[sec][k] = leaf
            else:
                sections[sec][k] = v
    fl = fp.pack_flat(fp.derive_flat(2, sections, kt=sim["kT"], salt_conc=0.5, half_charged_ends=True), _lib.param_names())
    P = orc.init_all(2, sections, kt=sim["kT"], salt_conc=0.5, half_charged_ends=True)
    seq, is_end, b, u = H.topo_tensors(top)
    frames = [0, 33, 77]
    u_ref = sum(orc.energy(2, P, torch.as_tensor(traj.center[f]), torch.as_tensor(traj.quaternions[f]), seq, is_end, b, u, traj.box_size)
                for f in frames)
    g_ref = torch.autograd.grad(u_ref, list(leaves.values()), allow_unused=True)
    g_ref = torch.stack([torch.zeros(()) .double() if g is None else g for g in g_ref])
    out = {"system": "oxDNA2 16 nt golden duplex, 3 frames", "n_parameters": len(leaves)}
    for name, dtype in (("f64", torch.float64), ("f32", torch.float32)):
        s = OxdnaSystem(2, top.seq, top.is_end, top.bonded_neighbors, box=traj.box_size, dtype=dtype, device=dev)
        s.set_params(fl.detach())
        s.set_neighbors(top.unbonded_neighbors)
        c = torch.as_tensor(traj.center[frames], dtype=dtype, device=dev)
        q = torch.as_tensor(traj.quaternions[frames], dtype=dtype, device=dev)
        _, _, _, gp = s.energy(c, q, grads=True, param_grads=True)
        g_flat = gp.sum(0).cpu()
        g = torch.autograd.grad(fl, list(leaves.values()), grad_outputs=g_flat, allow_unused=True, retain_graph=True)
        g = torch.stack([torch.zeros(()).double() if x is None else x for x in g])
        out[name] = {"max_abs_err": float((g - g_ref).abs().max()), "max_abs_grad": float(g_ref.abs().max()),
                     "rel": float((g - g_ref).abs().max() / g_ref.abs().max())}
    return out


def cpu_baseline(top, c0, q0, sim, n_steps: int, pairs: np.ndarray, budget_s: float = 20.0) -> dict:
    """Time the CPU oracle (torch fp64, vectorised over the same Verlet pair list) on the host: up to
    ``n_steps`` steps, stopping early once ``budget_s`` seconds are spent."""
    from oracle.langevin_oracle import LangevinOracle
    from tests import helpers as H

    torch.set_num_threads(host_cores())
    P = H.oracle_params(2, half_charged_ends=True)
    tt = (
        torch.as_tensor(top.seq, dtype=torch.long),
        torch.as_tensor(top.is_end, dtype=torch.long),
        torch.as_tensor(top.bonded_neighbors, dtype=torch.long),
        torch.as_tensor(pairs, dtype=torch.long),
    )
    kT = sim["kT"]
    o = LangevinOracle(2, P, tt, None, sim["dt"], kT, kT / sim["diff_coef"], kT / sim["rot_diff_coef"], seed=0)
    x, q = c0.copy(), q0.copy()
    p, L = np.zeros_like(x), np.zeros_like(x)
    t0 = time.perf_counter()
    o.run(x, q, p, L, 1)  # warm-up (thread pools, allocator); two force evaluations
    est = 0.5 * (time.perf_counter() - t0)
    n_steps = int(max(1, min(n_steps, budget_s / max(est, 1e-9))))
    print(f"[bench] cpu baseline: {n_steps} steps, about {est * n_steps:.0f} s on {torch.get_num_threads()} threads",
          file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    o.run(x, q, p, L, n_steps)
    dt = time.perf_counter() - t0
    return {
        "value": n_steps / dt,
        "unit": "steps/s",
        "cores": torch.get_num_threads(),
        "kind": "port",
        "sample": f"{n_steps} steps, same {top.n_nucleotides}-nt system, torch-fp64 oracle, {len(pairs)} listed pairs, {dt:.1f} s",
    }


def cpu_baseline_openmp(top, c0, q0, sim, flat, n_steps: int = 1000, repeats: int = 5, warmup: int = 100) -> dict:
    """The C++/OpenMP port (oracle/cpu_port: the same pair-physics templates as the HIP kernels compiled for the host,
    fp64, Verlet list rebuilt on the bench's schedule): ``warmup`` steps, then the median of ``repeats`` timings of
    ``n_steps`` steps each (at least 1 000 steps, median of 5), continuing one trajectory."""
    from oracle import cpu_port

    cores = host_cores()
    cpu_port.set_threads(cores)
    port = cpu_port.CpuPort(2, top.seq, top.is_end, top.bonded_neighbors, flat.detach().numpy())
    kT = sim["kT"]
    kw = dict(dt=sim["dt"], kT=kT, gamma_t=kT / sim["diff_coef"], gamma_r=kT / sim["rot_diff_coef"], mass=sim["nucleotide_mass"],
              inertia=sim["moment_of_inertia"], seed=0, r_cut=R_CUT, skin=0.6, rebuild_every=25)
    x, q = c0.copy(), q0.copy()
    p, L = np.zeros_like(x), np.zeros_like(x)
    t0 = time.perf_counter()
    port.run(x, q, p, L, warmup, step0=0, **kw)
    est = (time.perf_counter() - t0) / warmup
    n_steps = int(max(50, min(n_steps, 6.0 / max(est, 1e-9))))  # about 30 s for the five repeats at most
    print(f"[bench] cpu baseline (C++/OpenMP port): {repeats} x {n_steps} steps, about {est * n_steps * repeats:.0f} s on {cores} threads",
          file=sys.stderr, flush=True)
    rates, done = [], warmup
    for _ in range(repeats):
        t0 = time.perf_counter()
        port.run(x, q, p, L, n_steps, step0=done, **kw)
        rates.append(n_steps / (time.perf_counter() - t0))
        done += n_steps
    assert np.isfinite(x).all()
    return {"value": float(np.median(rates)), "unit": "steps/s", "cores": cores, "kind": "port",
            "cores_note": f"all {cores} the process may use (affinity mask and cgroup quota) of the host's {os.cpu_count()}",
            "sample": f"C++/OpenMP port (oracle/cpu_port) fp64, median of {repeats} x {n_steps} steps after {warmup}, same "
                      f"{top.n_nucleotides}-nt system",
            "rates": [float(r) for r in rates], "neighbor_list": {"r_cut": R_CUT, "skin": 0.6, "rebuild_every": 25}}


def martini_main(args):
    """BASELINE configs[2]: the reference's shipped DMPC bilayer (tests/golden/martini) tiled 4 x 4 = 20 480 beads,
    Langevin dt 0.02 ps, 273 K, friction 1/ps; 1 GPU.  Not the headline metric: a secondary line (the headline's
    ``secondary.cfg2_martini`` carries the same measurement)."""
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    # skin 0.5 nm, rebuild every 12 steps: re-scanned in round 3 (0.4 / 8: 65.9 k steps/s; 0.4 / 9 68.6 k; 0.45 / 11 69.9 k;
    # 0.5 / 12 68.8 k, all without an out-of-turn rebuild in 40 000 steps; 0.4 / 10 and 0.5 / 14 have them, 0.6 and above
    # collapse under them).  0.5 / 12 for its margin.
    skin, every = (0.5 if args.skin is None else args.skin), (12 if args.rebuild_every is None else args.rebuild_every)
    inner = None if args.inner_list is None else tuple(float(v) for v in args.inner_list.split(","))
    m = measure_martini(dev, args.dtype, args.steps, args.warmup, skin, every, repeats=max(1, args.repeats if args.repeats else 3), inner=inner)
    kms = m["timing"]["kernel_ms"]
    fr = _frac(m["alg"], kms)
    cpu = None if args.cpu_steps == 0 else martini_cpu_baseline(m["port_args"], m["xt"], m["bt"], m["kT"], 0.3, 10)
    print(json.dumps({
        "metric": "MD steps/sec per GPU, MARTINI-2 DMPC bilayer 20 480 beads", "value": m["steps_per_s"], "unit": "steps/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 / m["steps_per_s"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"MARTINI-2 DMPC bilayer (fixture tiled 4x4, {m['n']} beads), LJ r_c 1.1 nm + bonds + G96 angles, dt 0.02 ps, 273 K",
                   "thermostat": "Langevin, gamma 1/ps", "ns_per_day": m["steps_per_s"] * 0.02e-3 * 86400.0, "samples_ms": m["samples_ms"],
                   "neighbor_list": {"skin": skin, "rebuild_every": every, "mean_row": m["mean_row"], "max_row": m["max_row"], "pruned_rows": m["pruned_rows"],
                                     "out_of_turn_rebuilds": m["recoveries"]}},
        "roofline": {"bound": "hbm", "achieved": fr["GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": fr["frac"], "traffic": None,
                     "kernel": "martini_md_step_kernel", "kernel_ms": kms, "loop_ms_per_launch": m["timing"]["loop_ms_per_launch"],
                     "algorithmic_bytes_per_launch": m["alg"]},
        "cpu_baseline": cpu,
    }))


def spawn_ranks(args) -> int:
    """``python bench.py --gpus N`` without a launcher: start the N ranks as children of this process
    (torch.distributed.run, rendezvous on 127.0.0.1) and relay their output.  Runs before anything in this
    process has touched the GPU, and nothing is exec'ed: the parent only waits."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env, check=False).returncode


def measure(args, dtype_name: str, top, c0, q0, sim, flat, dev, seed: int, dist=None, md=None, positions_too: bool = False) -> dict:
    """Warm up, time ``args.steps`` steps of the resident state (barrier + synchronize on both sides, MAX over
    ranks) ``args.repeats`` times, then - outside the timed region - the same number of steps with 16 dispatches
    bracketed by HIP events on the launch stream (a bracketed dispatch costs ~8 us of queue time, so it is not part of
    the timed run).  ``positions_too``: afterwards the same samples once more with every step's positions + quaternions
    stored (config.save_every_1)."""
    dtype = torch.float32 if dtype_name == "f32" else torch.float64
    kT = sim["kT"]
    system = OxdnaSystem(2, top.seq, top.is_end, top.bonded_neighbors, box=None, dtype=dtype, device=dev)
    system.set_params(flat)
    integ = LangevinIntegrator(
        system, dt=sim["dt"], kT=kT, gamma_t=kT / sim["diff_coef"], gamma_r=kT / sim["rot_diff_coef"],
        mass=sim["nucleotide_mass"], inertia=sim["moment_of_inertia"], seed=seed,
    )
    integ.set_neighbor_policy(R_CUT, args.skin, args.rebuild_every)
    c = torch.as_tensor(c0, dtype=dtype, device=dev).contiguous()
    q = torch.as_tensor(q0, dtype=dtype, device=dev).contiguous()
    p, L = integ.init_momenta()
    integ.load(c, q, p, L)  # inputs resident in HBM, in the integrator's layout, before the clock starts

    # ---- warm-up (untimed): W steps, and on until --min-warmup-ms of stepping have passed (clock ramp); it also
    #      thermalises the ideal helix
    t_w = time.perf_counter()
    integ.advance(args.warmup)
    torch.cuda.synchronize(dev)
    warm_done = args.warmup
    while (time.perf_counter() - t_w) * 1e3 < args.min_warmup_ms:
        integ.advance(max(args.warmup, 100))
        torch.cuda.synchronize(dev)
        warm_done += max(args.warmup, 100)

    world = 1 if dist is None else dist.get_world_size()

    def timed(save_every: int, want_energy: bool):
        """args.repeats samples of exactly args.steps steps -> (samples [s], scheduled rebuilds per sample, recoveries)"""
        samples, rebuilds, recoveries = [], [], 0
        # rows of saved states go into tensors allocated once, outside the clock (what is timed is the integrator's path;
        # a caller that wants fresh tensors per call pays torch's allocator, ~20 us per call, on top)
        n_save = args.steps // save_every if save_every > 0 else 0
        rows = ((torch.empty((n_save, top.n_nucleotides, 3), dtype=dtype, device=dev), torch.empty((n_save, top.n_nucleotides, 4), dtype=dtype, device=dev))
                if n_save else None)
        for _ in range(max(1, args.repeats)):
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            _, _, et = integ.advance(args.steps, save_every=save_every, want_energy=want_energy, out=rows)
            if dist is not None and et is not None:
                # a run that saves energies (--save-every + --trace-energy) gathers them inside the timed region: per-replica
                # trace, replica id = rank, ONE all-gather over RCCL / xGMI and no host read-back.  Otherwise a replica
                # produces nothing to exchange: the MD data path has no collective, and none is invented for the clock.
                obs = et.reshape(1, -1)
                gathered = md.all_gather_observables(obs.cpu() if args.rehearse_on_one_gpu else obs, n_total=world)
                assert gathered.shape[0] == world
            torch.cuda.synchronize(dev)
            elapsed = time.perf_counter() - t0  # this rank's K steps; the MAX over ranks below is the job's
            if dist is not None:
                dist.barrier()  # (the closing bracket; its own latency is not part of anybody's K steps)
                t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.rehearse_on_one_gpu else dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                elapsed = float(t.item())
            samples.append(elapsed)
            rebuilds.append(integ.last_rebuilds())
            recoveries += integ.last_recoveries()
        return samples, rebuilds, recoveries

    # ---- timed region
    samples, rebuilds, recoveries = timed(args.save_every, args.trace_energy)
    total = sum(samples)
    order = sorted(range(len(samples)), key=lambda k: samples[k])
    mid = order[len(order) // 2]  # the median sample (the upper one of an even count)

    out_pos = None
    if positions_too and world == 1:
        ps, _, _ = timed(1, False)
        word = 4 if dtype_name == "f32" else 8
        out_pos = {"steps_per_s": len(ps) * args.steps / sum(ps), "vs_no_output": (len(ps) * args.steps / sum(ps)) / (len(samples) * args.steps / total),
                   "bytes_written_per_step": 7 * word * top.n_nucleotides, "samples_ms": [1e3 * t for t in ps]}

    # ---- instrumented pass (untimed): duration of the step kernel from HIP events attached to sampled dispatches on
    #      the launch stream (rocprofv3's kernel trace of the same command, profiles/, is the cross-check)
    integ.set_timing(16)
    integ.advance(max(args.steps, args.instrument_steps))  # sampled dispatches far apart: a bracketed one disturbs the few behind it
    timing = integ.last_kernel_ms()
    integ.set_timing(0)
    integ.store(c, q, p, L)
    torch.cuda.synchronize(dev)
    assert torch.isfinite(c).all() and torch.isfinite(q).all(), "state diverged"
    mx, nbar = system.neighbor_stats()
    n = top.n_nucleotides
    alg = algorithmic_bytes_per_step(n, nbar, 4 if dtype_name == "f32" else 8)
    if args.save_every > 0:
        alg += 7 * (4 if dtype_name == "f32" else 8) * n / args.save_every  # trajectory output adds 7 s N / save_every
    kms = timing["kernel_ms"]
    achieved = alg / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
    if out_pos is not None:
        out_pos["algorithmic_bytes_per_step"] = algorithmic_bytes_per_step(n, nbar, 4 if dtype_name == "f32" else 8) + out_pos["bytes_written_per_step"]
    return {"elapsed_mean": total / len(samples), "elapsed_median": samples[mid],
            "steps_per_s": world * len(samples) * args.steps / total, "steps_per_s_median": world * args.steps / samples[mid],
            "kernel_ms": kms, "loop_ms_per_launch": timing["loop_ms_per_launch"], "alg": alg, "achieved": achieved,
            "mean_row": nbar, "max_row": mx, "recoveries": recoveries, "samples_ms": [1e3 * t for t in samples],
            "rebuilds": rebuilds, "rebuilds_total": sum(rebuilds), "warmup_steps_done": warm_done, "save_every_1": out_pos}


def _timed_region(args, m, m2) -> str:
    """What the timed region held, in under 120 characters: steps, launches and list rebuilds over all repeats, and the
    rate at the other precision."""
    # (an advance call is one force evaluation per step: the closing half kick of its last step rides on the next call's
    #  first launch - or on store's - see advance_typed in mythos_amd/csrc/langevin_core.inc)
    closes = args.trace_energy and args.save_every > 0 and args.steps % args.save_every == 0
    r = len(m["samples_ms"])
    txt = (f"{r} x {args.steps} steps = {r * (args.steps + (1 if closes else 0))} launches + {m['rebuilds_total']} list rebuilds + {r} syncs; "
           f"value = steps / total time")
    if m2 is not None:
        other = "f64" if args.dtype == "f32" else "f32"
        txt += f"; {other} {m2['steps_per_s'] / 1e3:.1f}k"
    return txt


# ------------------------------------------------------------------------------------------------------------------
# secondary configs (N = 1): every BASELINE config gets a driver-measured number in the same line
# ------------------------------------------------------------------------------------------------------------------
def _frac(alg_bytes: float, kernel_ms: float) -> dict:
    gbs = alg_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    return {"kernel_ms": kernel_ms, "alg_bytes": alg_bytes, "GBs": gbs, "frac": gbs / HBM_PEAK_GBS}


def secondary_1kbp(dev, sim, flat, steps: int = 2000) -> dict:
    """BASELINE configs[1]: oxDNA2 1 kbp duplex with Debye-Hueckel, fp32 stepping + the fp64 energy check (energies and
    forces of the fp32 energy kernel against the fp64 one on the thermalised state the fp32 run produced)."""
    top, c0, q0 = generators.ideal_duplex(1000, model=2, seed=1234)
    kT = sim["kT"]
    n = top.n_nucleotides
    out = {"workload": f"oxDNA2 1 kbp duplex ({n} nt), Debye-Hueckel, Langevin dt {sim['dt']}, free space, {steps} steps after 500"}
    state = {}
    for name, dtype in (("f32", torch.float32), ("f64", torch.float64)):
        system = OxdnaSystem(2, top.seq, top.is_end, top.bonded_neighbors, box=None, dtype=dtype, device=dev)
        system.set_params(flat)
        integ = LangevinIntegrator(system, dt=sim["dt"], kT=kT, gamma_t=kT / sim["diff_coef"], gamma_r=kT / sim["rot_diff_coef"],
                                   mass=sim["nucleotide_mass"], inertia=sim["moment_of_inertia"], seed=7)
        integ.set_neighbor_policy(R_CUT, 0.9, 50)
        c = torch.as_tensor(c0, dtype=dtype, device=dev).contiguous()
        q = torch.as_tensor(q0, dtype=dtype, device=dev).contiguous()
        p, L = integ.init_momenta()
        integ.load(c, q, p, L)
        integ.advance(500)
        torch.cuda.synchronize(dev)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            integ.advance(steps)
            torch.cuda.synchronize(dev)
            ts.append(time.perf_counter() - t0)
        integ.set_timing(16)
        integ.advance(512)
        kms = integ.last_kernel_ms()["kernel_ms"]
        integ.set_timing(0)
        integ.store(c, q, p, L)
        torch.cuda.synchronize(dev)
        _, nbar = system.neighbor_stats()
        out[name] = {"steps_per_s": 3 * steps / sum(ts), **_frac(algorithmic_bytes_per_step(n, nbar, 4 if name == "f32" else 8), kms)}
        state[name] = (system, c, q)
    # fp64 energy check of the fp32 path: the same (thermalised) state through both energy kernels, on the fp32 run's list
    s32, c32, q32 = state["f32"]
    s64 = state["f64"][0]
    s64.build_neighbors(c32.double(), R_CUT, 0.0)
    s32.build_neighbors(c32, R_CUT, 0.0)
    e32, g32, _, _ = s32.energy(c32, q32, grads=True)
    e64, g64, _, _ = s64.energy(c32.double(), q32.double(), grads=True)
    out["f64_energy_check"] = {"rel_err_energy": float((e32.sum() - e64.sum()).abs() / e64.sum().abs()),
                               "rel_err_force": float((g32.double() - g64).abs().max() / g64.abs().max())}
    return out


def martini_system(dev, dtype):
    """The reference's shipped DMPC bilayer (tests/golden/martini) tiled 4 x 4 = 20 480 beads -> (MartiniSystem, the
    arguments a CPU port takes, positions, box)."""
    from mythos_amd.hip_system import MartiniSystem
    from tests import martini_helpers as MH

    s = MH.system()
    x, box, _ = MH.frames("lj")
    x0, b0 = x[3].copy(), box[3].copy()
    for i, j in s["top"].bonded_neighbors:  # whole lipids before tiling (GROMACS wraps bead by bead)
        d = x0[j] - x0[i]
        x0[j] = x0[i] + d - b0 * np.round(d / b0)
    reps = 4
    xt = np.concatenate([x0 + np.array([i * b0[0], j * b0[1], 0.0]) for i in range(reps) for j in range(reps)])
    bt = b0 * np.array([reps, reps, 1.0])
    top = s["top"].tile(reps * reps)
    tile = lambda a: np.tile(a, reps * reps)  # noqa: E731
    port_args = (tile(s["types"]), s["sigma"], s["eps"], top.bonded_neighbors, tile(s["bond_k"]), tile(s["bond_r0"]), top.angles,
                 tile(s["angle_k"]), tile(s["angle_t0"]))
    system = MartiniSystem(*port_args, dtype=dtype, device=dev)
    return system, port_args, xt, bt


def martini_cpu_baseline(port_args, xt, bt, kT: float, skin: float, every: int, budget_s: float = 6.0) -> dict:
    """The C++/OpenMP MARTINI port (oracle/cpu_port/martini_cpu.cpp, fp64) on the same 20 480 beads: a bounded sample."""
    from oracle import cpu_port

    cores = host_cores()
    cpu_port.set_threads(cores)
    port = cpu_port.MartiniCpuPort(*port_args)
    x, v = xt.copy(), np.zeros_like(xt)
    t0 = time.perf_counter()
    port.run(x, v, bt, 10, dt=0.02, kT=kT, gamma=1.0, seed=0, skin=skin, rebuild_every=every)
    est = (time.perf_counter() - t0) / 10
    n_steps = int(max(20, min(500, budget_s / max(est, 1e-9))))
    t0 = time.perf_counter()
    port.run(x, v, bt, n_steps, dt=0.02, kT=kT, gamma=1.0, seed=0, step0=10, skin=skin, rebuild_every=every)
    dt = time.perf_counter() - t0
    assert np.isfinite(x).all()
    return {"value": n_steps / dt, "unit": "steps/s", "cores": cores, "kind": "port",
            "sample": f"C++/OpenMP MARTINI port (oracle/cpu_port) fp64, {n_steps} steps after 10, same {xt.shape[0]} beads, {dt:.1f} s"}


def measure_martini(dev, dtype_name: str, steps: int, warmup: int, skin: float, every: int, repeats: int = 3, inner=None):
    """BASELINE configs[2] through the resident integrator: load, warm up, ``repeats`` x advance(steps), then an
    instrumented pass."""
    from mythos_amd.hip_system import MartiniLangevinIntegrator

    dtype = torch.float32 if dtype_name == "f32" else torch.float64
    word = 4 if dtype_name == "f32" else 8
    system, port_args, xt, bt = martini_system(dev, dtype)
    kT = 0.0083144626 * 273.0
    integ = MartiniLangevinIntegrator(system, dt=0.02, kT=kT, gamma=1.0, seed=0)
    integ.set_neighbor_policy(skin, every)
    inner = (0.2, 4) if inner is None else inner  # pruned rows: 75.5 k -> 77.6 k steps/s at 0.5 / 12 (profiles/r04_experiments.md)
    integ.set_inner_list(float(inner[0]), int(inner[1]))
    pos = torch.as_tensor(xt, dtype=dtype, device=dev).contiguous()
    vel = integ.init_velocities()
    integ.load(pos, vel, bt)
    integ.advance(warmup)
    torch.cuda.synchronize(dev)
    ts, rec = [], 0
    for _ in range(repeats):
        t0 = time.perf_counter()
        integ.advance(steps)
        torch.cuda.synchronize(dev)
        ts.append(time.perf_counter() - t0)
        rec += integ.last_recoveries()
    integ.set_timing(16)  # untimed pass: HIP event pairs on 16 dispatches
    integ.advance(max(steps, 512))
    timing = integ.last_kernel_ms()
    integ.set_timing(0)
    integ.store(pos, vel)
    torch.cuda.synchronize(dev)
    assert torch.isfinite(pos).all()
    mx, nbar = integ.neighbor_stats()
    pruned = {"margin": float(inner[0]), "every": int(inner[1]), "mean_row": float(integ.rows(True)[1].mean())} if inner[0] > 0 and inner[1] >= 2 else None
    n = system.n
    alg = n * (2 * 6 * word + 4 + 4.0 * nbar)  # 2 x (pos 3 + vel 3) words + type 4 B + 4 B per neighbour entry
    return {"steps_per_s": repeats * steps / sum(ts), "samples_ms": [1e3 * t for t in ts], "n": n, "mean_row": nbar, "max_row": mx, "pruned_rows": pruned,
            "recoveries": rec, "kT": kT, "timing": timing, "alg": alg, "port_args": port_args, "xt": xt, "bt": bt}


def secondary_martini(dev, skin: float = 0.5, every: int = 12, steps: int = 2000) -> dict:
    m = measure_martini(dev, "f32", steps, 300, skin, every)
    out = {"workload": f"MARTINI-2 DMPC bilayer (fixture tiled 4x4, {m['n']} beads), LJ + bonds + G96 angles, dt 0.02 ps, 273 K, fp32",
           "steps_per_s": m["steps_per_s"], "ns_per_day": m["steps_per_s"] * 0.02e-3 * 86400.0, "mean_row": m["mean_row"],
           "pruned_rows": m["pruned_rows"], "out_of_turn_rebuilds": m["recoveries"], **_frac(m["alg"], m["timing"]["kernel_ms"]),
           "loop_ms_per_launch": m["timing"]["loop_ms_per_launch"]}
    out["cpu_baseline"] = martini_cpu_baseline(m["port_args"], m["xt"], m["bt"], m["kT"], 0.3, 10)
    return out


def secondary_difftre(dev, sim_cfg, n_frames: int = 6400) -> dict:
    """BASELINE configs[4]: (i) U and dU/dtheta of 6 400 frames x 64 nt (32 bp, all-pairs list: 64 replicas x 100
    snapshots) per call of the energy kernel, both precisions, kernel time from events on the call's stream;
    (ii) ONE end-to-end DiffTRe iteration in the reference's fp64: 64 replicas x 2 000 steps in one launch per step, 100
    stored states each -> map -> loss + gradient of a reweighted propeller twist -> Adam, with the simulator's
    per-call host cost (run of 0 steps on the cached handles) and the host share of the whole iteration."""
    from mythos_amd.energy import dna2
    from mythos_amd.energy.base import Quaternion, RigidBody, space
    from mythos_amd.observables import PropellerTwist
    from mythos_amd.optimization import objective as O
    from mythos_amd.optimization.optimization import Adam, apply_updates
    from mythos_amd.simulators.hip_md import HipMDSimulator, StaticSimulatorParams, nvt_langevin
    kT = sim_cfg["kT"]
    top, c0, q0 = generators.ideal_duplex(32, model=2, seed=21)
    n = top.n_nucleotides
    sim, cfg = defaults.default_configs_for("dna2")
    flat = fp.pack_flat(fp.derive_flat(2, cfg, kt=kT, salt_conc=0.5, half_charged_ends=True), _lib.param_names())
    rng = np.random.default_rng(0)
    C = np.repeat(c0[None], n_frames, 0) + 0.02 * rng.standard_normal((n_frames, *c0.shape))
    Q = np.repeat(q0[None], n_frames, 0) + 0.01 * rng.standard_normal((n_frames, *q0.shape))
    Q /= np.linalg.norm(Q, axis=-1, keepdims=True)
    out = {"workload": f"oxDNA2 32 bp ({n} nt), all-pairs list; energy calls on {n_frames} frames; iteration: 64 replicas x 2000 steps (Verlet), fp64"}
    n_params = len(_lib.param_names())
    for name, dtype in (("f32", torch.float32), ("f64", torch.float64)):
        s = OxdnaSystem(2, top.seq, top.is_end, top.bonded_neighbors, box=None, dtype=dtype, device=dev)
        s.set_params(flat)
        s.set_neighbors(top.unbonded_neighbors)
        _, nbar = s.neighbor_stats()
        word = 4 if name == "f32" else 8
        cd, qd = torch.as_tensor(C, dtype=dtype, device=dev), torch.as_tensor(Q, dtype=dtype, device=dev)
        rec = {}
        for key, kw, written in (("U", {}, 64.0), ("dU_dtheta", {"grads": True, "param_grads": True}, 64.0 + 7 * word * n + 8.0 * n_params)):
            s.energy(cd, qd, **kw)
            torch.cuda.synchronize(dev)
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
            t0 = time.perf_counter()
            for a, b in ev:  # (the energy call runs on torch's current stream: these events bracket its launches)
                a.record()
                s.energy(cd, qd, **kw)
                b.record()
            torch.cuda.synchronize(dev)
            wall = (time.perf_counter() - t0) / len(ev)
            kms = float(np.median([a.elapsed_time(b) for a, b in ev]))
            alg = n_frames * ((7 * word + 13 + 4.0 * nbar) * n + written)  # per frame, read + written
            rec[key] = {"frames_per_s": n_frames / wall, "ms_per_call": 1e3 * wall, **_frac(alg, kms)}
        out[name] = rec

    # ---- one end-to-end iteration (fp64: the reference's precision on this path)
    disp, shift = space.free()
    ef = dna2.create_default_energy_fn(topology=top, displacement_fn=disp)
    init = RigidBody(center=torch.as_tensor(c0, device=dev), orientation=Quaternion(vec=torch.as_tensor(q0, device=dev)))
    sp = StaticSimulatorParams(seq=top.seq, mass=(1.0, (1.0, 1.0, 1.0)), gamma=(kT / 2.5, kT / 7.5), bonded_neighbors=top.bonded_neighbors,
                               checkpoint_every=0, dt=0.005, kT=kT)
    n_rep, steps, save_every = 64, 2000, 20
    # the sampling runs on the device-built Verlet list (the headline's policy), not on the reference's all-pairs set the energy
    # calls above and the reweighting below use: same forces (compact support), rows of 22 instead of 61 entries -
    # 2 000 steps 40.1 -> 35.6 ms in fp64, 25.0 -> 23.2 ms in fp32 (scripts/exp_difftre_list_r04.py)
    from mythos_amd.simulators.neighbors import VerletNeighborList

    simr = HipMDSimulator(energy_fn=ef, simulator_params=sp, space=(disp, shift), simulator_init=nvt_langevin,
                          neighbors=VerletNeighborList(R_CUT, 0.9, 50), save_every=save_every, dtype=torch.float64,
                          n_replicas=n_rep)
    half = n // 2
    ptwist = PropellerTwist(np.stack([np.arange(half), n - 1 - np.arange(half)], axis=1)[1:-1])
    target = 21.7
    opt = {"eps_stack_base": 1.3523, "eps_hb": 1.0678, "theta0_hb_4": float(np.pi)}
    adam = Adam(learning_rate=1e-3)
    adam_state = adam.init(opt)

    def loss_fn(ref_states, weights, energy_fn, opt_params, observables):  # noqa: ARG001
        mval = (weights * ptwist(ref_states).to(weights.dtype)).sum()
        return (mval - target) ** 2, (("propeller_twist", mval.detach()), {})

    def iteration(opt, adam_state, state, key, sim_obj=None):
        t = {}
        t0 = time.perf_counter()
        o = (sim_obj or simr).run(opt, state, steps, key=key)
        torch.cuda.synchronize(dev)
        t["md_ms"] = 1e3 * (time.perf_counter() - t0)
        traj = o.observables[0]
        if traj.center.dtype != torch.float64:  # fp32 sampling, fp64 reweighting (north_star: fp32 forces + fp64 energy check)
            from mythos_amd.simulators.io import SimulatorTrajectory

            traj = SimulatorTrajectory(center=traj.center.double(), orientation=Quaternion(vec=traj.orientation.vec.double()),
                                       temperature=traj.temperature, metadata=None)
        t0 = time.perf_counter()
        with torch.no_grad():
            ref_e = ef.with_params(opt).map(traj).detach()
        torch.cuda.synchronize(dev)
        t["map_ms"] = 1e3 * (time.perf_counter() - t0)
        t0 = time.perf_counter()
        (loss, (neff, _, _)), grads = O.compute_loss_and_grad(opt, ef, 1.0 / kT, loss_fn, traj, ref_e, [traj])
        torch.cuda.synchronize(dev)
        t["grad_ms"] = 1e3 * (time.perf_counter() - t0)
        t0 = time.perf_counter()
        upd, adam_state = adam.update({k: torch.as_tensor(v) for k, v in grads.items()}, adam_state, opt)
        opt = {k: float(v) for k, v in apply_updates(opt, upd).items()}
        t["adam_ms"] = 1e3 * (time.perf_counter() - t0)
        t["total_ms"] = sum(t.values())
        t["frames"] = int(traj.center.shape[0])
        t["loss"], t["neff"] = float(loss), float(neff)
        return opt, adam_state, o.state["init_state"], t

    opt, adam_state, state, _ = iteration(opt, adam_state, init, 1)  # first call: handles, lists, allocator
    opt, adam_state, state, t2 = iteration(opt, adam_state, state, 2)
    # what a run() costs on the host once the handles exist: 0 steps
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(5):
        simr.run(opt, state, 0, key=10 + k)
    torch.cuda.synchronize(dev)
    call_ms = 1e3 * (time.perf_counter() - t0) / 5
    # GPU time of the iteration's step launches: 2 000 launches at the kernel's own duration (events on sampled dispatches)
    system, integ, _ = next(iter(simr._resident.values()))
    integ.set_timing(16)
    simr.run(opt, state, 512, key=99)
    kms = integ.last_kernel_ms()["kernel_ms"]
    integ.set_timing(0)
    md_gpu_ms = steps * kms
    t2.update({"run_call_host_ms": call_ms, "md_kernel_ms": kms, "md_gpu_ms": md_gpu_ms,
               "host_share": max(0.0, 1.0 - (md_gpu_ms + out["f64"]["U"]["kernel_ms"] * t2["frames"] / n_frames +
                                            out["f64"]["dU_dtheta"]["kernel_ms"] * t2["frames"] / n_frames) / t2["total_ms"]),
               "replicas": n_rep, "steps": steps, "save_every": save_every, "n_params": len(opt)})
    out["iteration"] = t2
    simr.release()
    # the same iteration with the MD in fp32 (hi + lo centres) and the reweighting in fp64
    import dataclasses as _dc

    sim32 = _dc.replace(simr, dtype=torch.float32)
    o3, a3, st3, _ = iteration(opt, adam_state, init, 3, sim32)
    _, _, _, t3 = iteration(o3, a3, st3, 4, sim32)
    out["iteration_md_f32"] = {k: t3[k] for k in ("md_ms", "map_ms", "grad_ms", "adam_ms", "total_ms", "frames", "neff")}
    sim32.release()
    return out


def main():
    args = parse_args()
    for kv in args.debug_set:
        key, _, val = kv.partition("=")
        _lib.debug_set(key, int(val))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(spawn_ranks(args))  # no GPU call has been made in this process
    if args.workload == "martini-bilayer":
        return martini_main(args)
    # skin 0.9, rebuild every 50 steps: scanned on MI355X in round 3 (0.6 / 25, the round-2 choice: 67.1 k steps/s; 0.7 / 36
    # 68.7 k; 0.8 / 44 69.5 k; 0.9 / 52 70.1 k; 1.0 / 60 70.2 k; 1.2 / 90 66.0 k with out-of-turn rebuilds) - a site of the
    # thermalised 24 000-nt duplex first leaves half of a 0.9 skin after ~57 steps: 300 000 steps at 50 without one
    # out-of-turn rebuild, fp64 and 100 kbp likewise.  (The CPU port keeps its own optimum, 0.6 / 25.)
    args.skin = 0.9 if args.skin is None else args.skin
    args.rebuild_every = 50 if args.rebuild_every is None else args.rebuild_every
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    dev_index = 0 if args.rehearse_on_one_gpu else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = md = None
    if world > 1:
        import torch.distributed as dist

        from mythos_amd import distributed as md

        md.init("gloo" if args.rehearse_on_one_gpu else "nccl")  # "nccl" is RCCL on ROCm

    sim, cfg = defaults.default_configs_for("dna2")
    sim["dt"] = args.dt
    kT = sim["kT"]
    top, c0, q0 = generators.ideal_duplex(args.bp, model=2, seed=1234)
    n = top.n_nucleotides
    flat = fp.pack_flat(fp.derive_flat(2, cfg, kt=kT, salt_conc=sim["salt_conc"], half_charged_ends=True), _lib.param_names())

    extras = world == 1 and not args.no_secondary
    m = measure(args, args.dtype, top, c0, q0, sim, flat, dev, rank, dist, md, positions_too=extras and args.save_every == 0)
    # the reference computes in fp64 (jax_enable_x64): the same measurement at that precision goes into the same line
    other = "f64" if args.dtype == "f32" else "f32"
    m2 = (measure(args, other, top, c0, q0, sim, flat, dev, rank, dist, md, positions_too=extras and args.save_every == 0)
          if not args.no_second_dtype else None)

    if rank == 0:
        out = {
            "metric": "MD steps/sec (and ns/day) per GPU, oxDNA2 12 kbp duplex",
            "value": m["steps_per_s"],
            "unit": "steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * m["elapsed_mean"] / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            # (strings of the record are kept under 120 characters: the driver's parsed copy truncates there)
            "config": {
                "workload": f"oxDNA2 {args.bp} bp duplex ({n} nt), Debye-Hueckel salt 0.5, Langevin dt {sim['dt']}, free space, 1 replica/GPU",
                "kT": kT,
                "half_charged_ends": True,
                "replicas": world,
                **({"rehearsal": "all ranks on cuda:0 over gloo - not a measurement"} if args.rehearse_on_one_gpu else {}),
                "neighbor_list": {"r_cut": R_CUT, "skin": args.skin, "rebuild_every": args.rebuild_every, "mean_row": m["mean_row"],
                                  "max_row": m["max_row"], "out_of_turn_rebuilds": m["recoveries"]},
                "ns_per_day": m["steps_per_s"] / world * sim["dt"] * OXDNA_TIME_UNIT_PS * 86400.0 * 1e-3,
                "timed_region": _timed_region(args, m, m2),
                "repeats": len(m["samples_ms"]),
                "samples_ms": m["samples_ms"],
                "scheduled_rebuilds_per_sample": m["rebuilds"],
                "warmup_steps_done": m["warmup_steps_done"],
                "ms_per_step_median": 1e3 * m["elapsed_median"] / args.steps,
                "precision": f"headline {args.dtype} (north_star: fp32 at 1e-3); the reference's fp64 is config.{other} and f64_steps_per_s",
            },
            # value = steps_per_s_mean: all repeats over their total time, list rebuilds included; the median sample beside it
            "steps_per_s_mean": m["steps_per_s"],
            "steps_per_s_median": m["steps_per_s_median"],
            "nproc": os.cpu_count(),
            "roofline": {
                "bound": "hbm",
                "achieved": m["achieved"],
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": m["achieved"] / HBM_PEAK_GBS,
                "traffic": measured_traffic(args.dtype, n) if args.save_every == 0 else None,
                "traffic_source": "profiles/traffic.json: PMC passes of this command (rocprofv3), not measured inside this run",
                "kernel": "md_step_kernel",
                "kernel_ms": m["kernel_ms"],
                "loop_ms_per_launch": m["loop_ms_per_launch"],
                "algorithmic_bytes_per_launch": m["alg"],
            },
        }
        if m2 is not None:
            out[f"{other}_steps_per_s"] = m2["steps_per_s"]  # top level too: nested objects do not survive every parser
            out[f"{other}_frac"] = m2["achieved"] / HBM_PEAK_GBS
            out[f"{other}_kernel_ms"] = m2["kernel_ms"]
            out["config"][other] = {"steps_per_s": m2["steps_per_s"], "steps_per_s_median": m2["steps_per_s_median"],
                                    "ms_per_step": 1e3 * m2["elapsed_mean"] / args.steps,
                                    "samples_ms": m2["samples_ms"], "scheduled_rebuilds_per_sample": m2["rebuilds"],
                                    "kernel_ms": m2["kernel_ms"], "loop_ms_per_launch": m2["loop_ms_per_launch"],
                                    "achieved_GBs": m2["achieved"], "frac": m2["achieved"] / HBM_PEAK_GBS,
                                    "algorithmic_bytes_per_launch": m2["alg"],
                                    "traffic": measured_traffic(other, n) if args.save_every == 0 else None}
        if m["save_every_1"] is not None:
            out["config"]["save_every_1"] = {args.dtype: m["save_every_1"], **({other: m2["save_every_1"]} if m2 is not None else {}),
                                             "what": "the same K steps, every step's positions + quaternions stored (jaxmd.py:84-99)"}
        if extras:
            sec = {}
            for key, fn in (("cfg1_1kbp", lambda: secondary_1kbp(dev, sim, flat)), ("cfg2_martini", lambda: secondary_martini(dev)),
                            ("cfg4_difftre", lambda: secondary_difftre(dev, sim))):
                t0 = time.perf_counter()
                try:
                    sec[key] = fn()
                except Exception as exc:  # noqa: BLE001 - a secondary config must not take the headline line down
                    sec[key] = {"error": f"{type(exc).__name__}: {exc}"[:118]}
                sec[key]["wall_s"] = time.perf_counter() - t0
                print(f"[bench] secondary {key}: {sec[key].get('wall_s', 0):.1f} s", file=sys.stderr, flush=True)
            out["secondary"] = sec
        cpu_steps = args.cpu_steps
        if cpu_steps < 0:
            cpu_steps = 60 if n > 8000 else 400  # capped at about 20 s of host work
        if cpu_steps > 0 and world == 1:
            from mythos_amd.simulators.neighbors import verlet_pairs_numpy

            # two CPU numbers: the C++/OpenMP port (the value: the faster, fairer baseline) and, as in round 1, the
            # vectorised torch-fp64 restatement that stands in for JAX-CPU (JAX is not installable offline)
            out["cpu_baseline"] = cpu_baseline_openmp(top, c0, q0, sim, flat)
            pairs = verlet_pairs_numpy(c0, top.bonded_neighbors, R_CUT + 0.1)
            out["cpu_baseline"]["torch_restatement"] = cpu_baseline(top, c0, q0, sim, min(cpu_steps, 30), pairs, budget_s=8.0)
            out["config"]["dU_dtheta_vs_cpu"] = param_gradient_error(dev)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
