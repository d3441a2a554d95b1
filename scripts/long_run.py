#!/usr/bin/env python3
"""Stability soak: the bench system for many steps, energies and kinetic temperature every 2 000 steps."""
import sys
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
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40000
skin = float(sys.argv[3]) if len(sys.argv) > 3 else 0.6
every = int(sys.argv[4]) if len(sys.argv) > 4 else 25
dtype = torch.float64 if (len(sys.argv) > 5 and sys.argv[5] == "f64") else torch.float32
sim, cfg = defaults.default_configs_for("dna2")
kT = sim["kT"]
top, c0, q0 = generators.ideal_duplex(bp, model=2, seed=1234)
flat = fp.pack_flat(fp.derive_flat(2, cfg, kt=kT, salt_conc=0.5, half_charged_ends=True), _lib.param_names())
s = OxdnaSystem(2, top.seq, top.is_end, top.bonded_neighbors, box=None, dtype=dtype)
s.set_params(flat)
integ = LangevinIntegrator(s, dt=0.005, kT=kT, gamma_t=kT / 2.5, gamma_r=kT / 7.5, seed=1)
integ.set_neighbor_policy(3.25, skin, every)
c = torch.as_tensor(c0, dtype=dtype, device=s.device).contiguous()
q = torch.as_tensor(q0, dtype=dtype, device=s.device).contiguous()
p, L = integ.init_momenta()
n = top.n_nucleotides
for blk in range(n_steps // 2000):
    try:
        _, _, et = integ.run(c, q, p, L, 2000, save_every=2000)
    except Exception as exc:  # a skin violation ends the soak with its message
        print(f"step {2000 * blk}..{2000 * (blk + 1)}: {exc}")
        sys.exit(1)
    e = et[-1].cpu().numpy()
    mx, mean = s.neighbor_stats()
    print(f"step {2000 * (blk + 1):6d}  U/N {e[:8].sum() / n:8.4f}  T_kin/T {e[8:].sum() / (3 * n * kT):6.4f}  terms/N "
          + " ".join(f"{v / n:7.4f}" for v in e[:8]) + f"  rows mean {mean:.1f} max {mx}", flush=True)
assert torch.isfinite(c).all()
