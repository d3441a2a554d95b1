#!/usr/bin/env python3
"""Secondary benchmark: energy / force / dU/dtheta evaluation rate of mythos_oxdna_energy (the DiffTRe hot path,
BASELINE configs[4]: oxDNA2 32 bp, frames from 64 replicas x 100 snapshots) and of a 1 kbp duplex (configs[1]).

    python scripts/bench_energy.py            # prints one JSON object per case
    python scripts/bench_energy.py --obs      # the DiffTRe shape only, with the observables in every call (mythos_oxdna_energy_obs)
                                              # (propeller twist, rise, pitch, persistence length)
"""
import json, sys, time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from mythos_amd import _lib  # noqa: E402
from mythos_amd.energy import flat_params as fp  # noqa: E402
from mythos_amd.hip_system import OxdnaSystem  # noqa: E402
from mythos_amd.input import defaults  # noqa: E402
from mythos_amd.simulators.neighbors import verlet_pairs_numpy  # noqa: E402
from mythos_amd.utils import generators  # noqa: E402


def case(bp, frames, dtype, reps=20, obs=False):
    top, c, q = generators.ideal_duplex(bp, model=2, seed=1234)
    sim, cfg = defaults.default_configs_for("dna2")
    flat = fp.pack_flat(fp.derive_flat(2, cfg, kt=sim["kT"], salt_conc=0.5, half_charged_ends=True), _lib.param_names())
    s = OxdnaSystem(2, top.seq, top.is_end, top.bonded_neighbors, box=None, dtype=dtype)
    s.set_params(flat)
    if top.n_nucleotides <= 512:
        s.set_neighbors(top.unbonded_neighbors)  # the reference's all-pairs list
    else:
        s.set_neighbors(verlet_pairs_numpy(c, top.bonded_neighbors, 3.25))
    rng = np.random.default_rng(0)
    C = np.repeat(c[None], frames, 0) + 0.02 * rng.standard_normal((frames, *c.shape))
    Q = np.repeat(q[None], frames, 0) + 0.01 * rng.standard_normal((frames, *q.shape))
    Q /= np.linalg.norm(Q, axis=-1, keepdims=True)
    cd = torch.as_tensor(C, dtype=dtype, device=s.device)
    qd = torch.as_tensor(Q, dtype=dtype, device=s.device)
    out = {"bp": bp, "n": top.n_nucleotides, "frames": frames, "dtype": str(dtype).split(".")[-1], "pairs": int(s.neighbor_stats()[1] * top.n_nucleotides / 2)}
    extra = {}
    if obs:
        from mythos_amd.energy.base import _fused_observables, space
        from mythos_amd.observables import PersistenceLength, PitchAngle, PropellerTwist, Rise, get_duplex_quartets
        disp = space.free()[0]
        quartets = get_duplex_quartets(bp)
        pairs = np.stack([np.arange(bp), 2 * bp - 1 - np.arange(bp)], axis=1)[1:-1]
        ob = [PropellerTwist(pairs), Rise(quartets, disp, cfg["geometry"]), PitchAngle(quartets, disp, cfg["geometry"]),
              PersistenceLength(quartets, disp, cfg["geometry"], truncate=10)]
        oset, served = _fused_observables(ob, top.n_nucleotides, dtype, s.device)
        assert oset is not None and len(served) == len(ob)
        extra = {"observables": oset}
        out["observables"] = [type(o).__name__ for o in served]
    for name, kw in (("energy", {}), ("energy+forces", {"grads": True}), ("energy+forces+dU/dtheta", {"grads": True, "param_grads": True})):
        kw = {**kw, **extra}
        s.energy(cd, qd, **kw)
        torch.cuda.synchronize()
        per_call = []  # median of synchronised calls: one allocator or clock hiccup in twenty must not set the figure
        for _ in range(reps):
            t0 = time.perf_counter()
            s.energy(cd, qd, **kw)
            torch.cuda.synchronize()
            per_call.append(time.perf_counter() - t0)
        dt = float(np.median(per_call))
        out[name] = {"ms_per_call": 1e3 * dt, "frames_per_s": frames / dt}
    return out


if __name__ == "__main__":
    if "--obs" in sys.argv[1:]:
        for dtype in (torch.float64, torch.float32):
            print(json.dumps(case(32, 6400, dtype, obs=True)))
        sys.exit(0)
    if "--difftre" in sys.argv[1:]:  # the DiffTRe shape only (profiles: one shape per kernel-time mean)
        for dtype in (torch.float64, torch.float32):
            print(json.dumps(case(32, 6400, dtype)))
        sys.exit(0)
    for bp, frames, dtype in ((32, 6400, torch.float64), (32, 6400, torch.float32), (1000, 64, torch.float64), (1000, 64, torch.float32)):
        print(json.dumps(case(bp, frames, dtype)))
