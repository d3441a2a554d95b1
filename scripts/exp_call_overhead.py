#!/usr/bin/env python3
"""dev: what one mythos_langevin_advance call costs besides its launches - wall clock of advance(n) + synchronize for
n = 1, 2, 5, 10, 20 on the 12 kbp duplex (skin 2.0 and no scheduled rebuild: no list rebuild inside), median of 50 calls each;
a straight line through them gives the per-launch period (slope) and the per-call cost (intercept)."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from mythos_amd import _lib  # noqa: E402
from mythos_amd.energy import flat_params as fp  # noqa: E402
from mythos_amd.hip_system import LangevinIntegrator, OxdnaSystem  # noqa: E402
from mythos_amd.input import defaults  # noqa: E402
from mythos_amd.utils import generators  # noqa: E402

bp = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
top, c0, q0 = generators.ideal_duplex(bp, model=2, seed=1234)
sim, cfg = defaults.default_configs_for("dna2")
flat = fp.pack_flat(fp.derive_flat(2, cfg, kt=sim["kT"], salt_conc=0.5, half_charged_ends=True), _lib.param_names())
dev = torch.device("cuda", 0)
s = OxdnaSystem(2, top.seq, top.is_end, top.bonded_neighbors, box=None, dtype=torch.float32, device=dev)
s.set_params(flat)
kT = sim["kT"]
integ = LangevinIntegrator(s, dt=0.005, kT=kT, gamma_t=kT / 2.5, gamma_r=kT / 7.5, seed=1)
integ.set_neighbor_policy(3.25, 0.6, 25)
c = torch.as_tensor(c0, dtype=torch.float32, device=dev).contiguous()
q = torch.as_tensor(q0, dtype=torch.float32, device=dev).contiguous()
p, L = integ.init_momenta()
integ.load(c, q, p, L)
integ.advance(500)
integ.store(c, q, p, L)
ns, med, rec = [1, 2, 5, 10, 20], [], 0
for n in ns:
    integ.set_neighbor_policy(3.25, 2.0, 1_000_000)  # no scheduled rebuild; a halt would show as an outlier
    integ.load(c, q, p, L)
    integ.advance(5)
    ts = []
    for _ in range(50):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        integ.advance(n)
        torch.cuda.synchronize(dev)
        ts.append(time.perf_counter() - t0)
        rec += integ.last_recoveries()
    med.append(1e6 * float(np.median(ts)))
slope, icpt = np.polyfit(ns, med, 1)
print("n:", ns, "out-of-turn rebuilds:", rec)
print("median us per call:", [round(m, 1) for m in med])
print(f"per launch {slope:.2f} us, per call {icpt:.1f} us")
