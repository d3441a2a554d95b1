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
when K < 500: a 20-step sample lasts a third of a millisecond, and on a GPU that sat idle until the clock started the
first few samples run at clocks still coming up - 0.345 ms against 0.325 ms a few milliseconds later, measured);
value = N_gpus * K / the MEDIAN sample, all samples are in ``config.samples_ms`` and the number of scheduled
list rebuilds that fell inside each in ``config.scheduled_rebuilds_per_sample``.  Nothing else is inside a
sample: the per-dispatch HIP events behind ``roofline.kernel_ms`` are taken in a second, untimed pass.  For
N > 1 every rank steps its own replica (weak scaling; the MD data path has no collective): barrier + synchronize, the
rank's clock around its K steps, synchronize + barrier, MAX over ranks.  With ``--save-every`` the replicas' energy
traces are all-gathered over RCCL inside the timed region (one collective).  The headline precision is fp32 (north_star: fp32 forces at 1e-3); the same measurement in the
reference's fp64 is ``config.f64``, ``f64_steps_per_s`` and the tail of ``config.timed_region``.

Rank 0 prints ONE JSON line with the fields of the driver contract plus
  roofline      HBM roofline of the step kernel: algorithmic bytes per launch (2 x 14 state
                words + 13 B topology + 4 B per neighbour entry, per nucleotide) / the kernel's mean duration,
                measured with HIP events on the launch stream inside the timed region.  ``traffic`` is the
                PMC-measured HBM-side byte count per launch (2 x FETCH_SIZE + WRITE_SIZE, gfx950 correction)
                of the SAME command, read from profiles/traffic.json when that file matches the workload.
  cpu_baseline  the C++/OpenMP port of the same kernels (oracle/cpu_port, fp64, all host cores; median of 5 x 1 000
                steps) and, under ``torch_restatement``, the vectorised torch-fp64 oracle that stands in for JAX-CPU,
                both on a bounded sample of the same system (N=1, rank 0 only)
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

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
OXDNA_TIME_UNIT_PS = 3.03  # oxDNA simulation time unit
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


def measured_traffic(args, n: int):
    """HBM-side bytes per md_step_kernel launch from the committed PMC passes of this command
    (profiles/traffic.json, written by scripts/collect_traffic.py), or None if it is for another workload."""
    f = ROOT / "profiles" / "traffic.json"
    if not f.exists():
        return None
    t = json.loads(f.read_text())
    if t.get("n_nucleotides") != n or t.get("dtype") != args.dtype:
        return None
    return t["hbm_bytes_per_launch"]


def host_cores() -> int:
    """Cores this process may really use: affinity mask, capped by the cgroup CPU quota (a GPU box hands
    a 1-GPU job a share of a much larger host; oversubscribing it makes the baseline meaningless)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get("MYTHOS_BENCH_CPU_CORES", "16"))))


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
            "sample": f"C++/OpenMP port (oracle/cpu_port) fp64, median of {repeats} x {n_steps} steps after {warmup}, same "
                      f"{top.n_nucleotides}-nt system",
            "rates": [float(r) for r in rates], "neighbor_list": {"r_cut": R_CUT, "skin": 0.6, "rebuild_every": 25}}


def martini_main(args):
    """BASELINE configs[2]: the reference's shipped DMPC bilayer (tests/golden/martini) tiled 4 x 4 = 20 480 beads,
    Langevin dt 0.02 ps, 273 K, friction 1/ps; 1 GPU.  Not the headline metric: a secondary line."""
    from mythos_amd.hip_system import MartiniLangevinIntegrator, MartiniSystem
    from tests import martini_helpers as MH

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dtype = torch.float32 if args.dtype == "f32" else torch.float64
    word = 4 if args.dtype == "f32" else 8
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
    system = MartiniSystem(tile(s["types"]), s["sigma"], s["eps"], top.bonded_neighbors, tile(s["bond_k"]), tile(s["bond_r0"]),
                           top.angles, tile(s["angle_k"]), tile(s["angle_t0"]), dtype=dtype, device=dev)
    kT = 0.0083144626 * 273.0
    integ = MartiniLangevinIntegrator(system, dt=0.02, kT=kT, gamma=1.0, seed=0)
    # skin 0.5 nm, rebuild every 12 steps: re-scanned in round 3 (0.4 / 8: 65.9 k steps/s; 0.4 / 9 68.6 k; 0.45 / 11 69.9 k;
    # 0.5 / 12 68.8 k, all without an out-of-turn rebuild in 40 000 steps; 0.4 / 10 and 0.5 / 14 have them, 0.6 and above
    # collapse under them).  0.5 / 12 for its margin.
    skin, every = (0.5 if args.skin is None else args.skin), (12 if args.rebuild_every is None else args.rebuild_every)
    integ.set_neighbor_policy(skin, every)
    pos = torch.as_tensor(xt, dtype=dtype, device=dev).contiguous()
    vel = integ.init_velocities()
    integ.run(pos, vel, bt, args.warmup)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    integ.run(pos, vel, bt, args.steps)
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    integ.set_timing(16)  # untimed second pass: HIP event pairs on 16 dispatches (roofline.kernel_ms)
    integ.run(pos, vel, bt, max(args.steps, 64))
    timing = integ.last_kernel_ms()
    mx, nbar = integ.neighbor_stats()
    n = system.n
    alg = n * (2 * 6 * word + 4 + 4.0 * nbar)
    kms = timing["kernel_ms"]
    assert torch.isfinite(pos).all()
    print(json.dumps({
        "metric": "MD steps/sec per GPU, MARTINI-2 DMPC bilayer 20 480 beads", "value": args.steps / elapsed, "unit": "steps/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"MARTINI-2 DMPC bilayer (fixture tiled 4x4, {n} beads), LJ r_c 1.1 nm + bonds + G96 angles, dt 0.02 ps, 273 K",
                   "thermostat": "Langevin, gamma 1/ps", "ns_per_day": args.steps / elapsed * 0.02e-3 * 86400.0,
                   "neighbor_list": {"skin": skin, "rebuild_every": every, "mean_row": nbar, "max_row": mx,
                                     "out_of_turn_rebuilds": integ.last_recoveries()}},
        "roofline": {"bound": "hbm", "achieved": alg / (kms * 1e-3) / 1e9 if kms > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": (alg / (kms * 1e-3) / 1e9 if kms > 0 else 0.0) / HBM_PEAK_GBS, "traffic": None,
                     "kernel": "martini_md_step_kernel", "kernel_ms": kms, "loop_ms_per_launch": timing["loop_ms_per_launch"],
                     "algorithmic_bytes_per_launch": alg},
        "cpu_baseline": None,
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


def measure(args, dtype_name: str, top, c0, q0, sim, flat, dev, seed: int, dist=None, md=None) -> dict:
    """Warm up, time ``args.steps`` steps of the resident state (barrier + synchronize on both sides, MAX over
    ranks), then - outside the timed region - the same number of steps with 16 dispatches bracketed by HIP events
    on the launch stream (a bracketed dispatch costs ~8 us of queue time, so it is not part of the timed run)."""
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

    # ---- warm-up (untimed): also thermalises the ideal helix
    integ.advance(args.warmup)
    torch.cuda.synchronize(dev)

    # ---- timed region: exactly args.steps steps, measured args.repeats times back to back (the trajectory simply
    #      continues); every sample has its own barrier + synchronize on both sides and its own MAX over ranks
    world = 1 if dist is None else dist.get_world_size()
    samples, rebuilds, recoveries = [], [], 0
    for _ in range(max(1, args.repeats)):
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        _, _, et = integ.advance(args.steps, save_every=args.save_every, want_energy=args.trace_energy)
        if dist is not None and et is not None:
            # a run that saves observables (--save-every) gathers them inside the timed region: per-replica energy trace,
            # replica id = rank, ONE all-gather over RCCL / xGMI and no host read-back.  Without --save-every a replica
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
    order = sorted(range(len(samples)), key=lambda k: samples[k])
    mid = order[len(order) // 2]  # the median sample (the upper one of an even count): value, ms_per_step and its rebuild count
    elapsed = samples[mid]

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
    kms = timing["kernel_ms"]
    achieved = alg / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
    return {"elapsed": elapsed, "steps_per_s": world * args.steps / elapsed, "kernel_ms": kms,
            "loop_ms_per_launch": timing["loop_ms_per_launch"], "alg": alg, "achieved": achieved,
            "mean_row": nbar, "max_row": mx, "recoveries": recoveries, "samples_ms": [1e3 * t for t in samples],
            "rebuilds": rebuilds, "rebuilds_in_median": rebuilds[mid]}


def _timed_region(args, m, m2) -> str:
    """What the timed region held, in under 120 characters: steps, launches, list rebuilds inside the median sample,
    and the rate at the other precision."""
    # (an advance call is one force evaluation per step: the closing half kick of its last step rides on the next call's
    #  first launch - or on store's - see advance_typed in mythos_amd/csrc/langevin_core.inc)
    closes = args.trace_energy and args.save_every > 0 and args.steps % args.save_every == 0
    txt = (f"{args.steps} steps = {args.steps + (1 if closes else 0)} launches + {m['rebuilds_in_median']} list rebuilds + 1 sync; "
           f"median of {len(m['samples_ms'])}")
    if m2 is not None:
        other = "f64" if args.dtype == "f32" else "f32"
        txt += f"; {other} {m2['steps_per_s'] / 1e3:.1f}k steps/s"
    return txt


def main():
    args = parse_args()
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

    m = measure(args, args.dtype, top, c0, q0, sim, flat, dev, rank, dist, md)
    # the reference computes in fp64 (jax_enable_x64): the same measurement at that precision goes into the same line
    other = "f64" if args.dtype == "f32" else "f32"
    m2 = measure(args, other, top, c0, q0, sim, flat, dev, rank, dist, md) if not args.no_second_dtype else None

    if rank == 0:
        out = {
            "metric": "MD steps/sec (and ns/day) per GPU, oxDNA2 12 kbp duplex",
            "value": m["steps_per_s"],
            "unit": "steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * m["elapsed"] / args.steps,
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
                "precision": f"headline {args.dtype} (north_star: fp32 at 1e-3); the reference's fp64 is config.{other} and f64_steps_per_s",
            },
            "roofline": {
                "bound": "hbm",
                "achieved": m["achieved"],
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": m["achieved"] / HBM_PEAK_GBS,
                "traffic": measured_traffic(args, n),
                "traffic_source": "profiles/traffic.json: PMC passes of this command (rocprofv3), not measured inside this run",
                "kernel": "md_step_kernel",
                "kernel_ms": m["kernel_ms"],
                "loop_ms_per_launch": m["loop_ms_per_launch"],
                "algorithmic_bytes_per_launch": m["alg"],
            },
        }
        if m2 is not None:
            out[f"{other}_steps_per_s"] = m2["steps_per_s"]  # top level too: nested objects do not survive every parser
            out["config"][other] = {"steps_per_s": m2["steps_per_s"], "ms_per_step": 1e3 * m2["elapsed"] / args.steps,
                                    "samples_ms": m2["samples_ms"], "scheduled_rebuilds_per_sample": m2["rebuilds"],
                                    "kernel_ms": m2["kernel_ms"], "loop_ms_per_launch": m2["loop_ms_per_launch"],
                                    "achieved_GBs": m2["achieved"], "frac": m2["achieved"] / HBM_PEAK_GBS,
                                    "algorithmic_bytes_per_launch": m2["alg"]}
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
