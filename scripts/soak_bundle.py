#!/usr/bin/env python3
"""Robustness soak: a bundle of short duplexes in a periodic box (many strands, a hashed or direct cell table that
keeps changing occupancy as the duplexes diffuse), default neighbour policy, fp32.  Prints energies every block and
fails on any error of the run (skin violation, row / spill overflow, NaN).

    python scripts/soak_bundle.py [n_duplexes] [n_bp] [steps] [box]
"""
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

n_dup = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n_bp = int(sys.argv[2]) if len(sys.argv) > 2 else 48
n_steps = int(sys.argv[3]) if len(sys.argv) > 3 else 100000
box_edge = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
sim, cfg = defaults.default_configs_for("dna2")
kT = sim["kT"]
top, c0, q0 = generators.duplex_bundle(n_bp, n_dup, spacing=5.0, seed=7)
box = None
if box_edge > 0:
    box = np.array([box_edge] * 3)
    c0 = c0 - c0.min(0) + 1.0
flat = fp.pack_flat(fp.derive_flat(2, cfg, kt=kT, salt_conc=0.5, half_charged_ends=True), _lib.param_names())
s = OxdnaSystem(2, top.seq, top.is_end, top.bonded_neighbors, box=box, dtype=torch.float32)
s.set_params(flat)
integ = LangevinIntegrator(s, dt=0.005, kT=kT, gamma_t=kT / 2.5, gamma_r=kT / 7.5, seed=11)
integ.set_neighbor_policy(3.25, 0.6, 25)
c = torch.as_tensor(c0, dtype=torch.float32, device=s.device).contiguous()
q = torch.as_tensor(q0, dtype=torch.float32, device=s.device).contiguous()
p, L = integ.init_momenta()
n = top.n_nucleotides
print(f"{n_dup} duplexes x {n_bp} bp = {n} nt, box {box_edge or 'free'}", flush=True)
block = 5000
for blk in range(n_steps // block):
    _, _, et = integ.run(c, q, p, L, block, save_every=block)
    e = et[-1].cpu().numpy()
    mx, mean = s.neighbor_stats()
    print(f"step {block * (blk + 1):7d}  U/N {e[:8].sum() / n:8.4f}  T_kin/T {e[8:].sum() / (3 * n * kT):6.4f}  HB/N {e[4] / n:7.4f}"
          f"  rows mean {mean:.1f} max {mx}", flush=True)
assert torch.isfinite(c).all()
print("soak ok")
