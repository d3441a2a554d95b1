#!/usr/bin/env python3
"""dev (round 3): where the fp32 energy kernel's force error at 12 kbp comes from - per-nucleotide error against the oracle,
the same configuration translated to the origin, and the worst nucleotides with their neighbours' distances."""
import sys
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
from oracle import oxdna_oracle as orc  # noqa: E402
from tests import helpers as H  # noqa: E402

bp = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
top, c0, q0 = generators.ideal_duplex(bp, model=2, seed=1234)
rng = np.random.default_rng(21)
c0 = (c0 + 0.02 * rng.standard_normal(c0.shape)).astype(np.float32).astype(np.float64)
q0 = q0 + 0.01 * rng.standard_normal(q0.shape)
q0 = (q0 / np.linalg.norm(q0, axis=1, keepdims=True)).astype(np.float32).astype(np.float64)
pairs = verlet_pairs_numpy(c0, top.bonded_neighbors, 3.85)
P = H.oracle_params(2, half_charged_ends=True)
tt = (torch.as_tensor(top.seq, dtype=torch.long), torch.as_tensor(top.is_end, dtype=torch.long),
      torch.as_tensor(top.bonded_neighbors, dtype=torch.long), torch.as_tensor(pairs, dtype=torch.long))
def reference(c):
    _, gc, gq = orc.energy_and_grads(2, P, torch.as_tensor(c), torch.as_tensor(q0), *tt, box=None)
    return gc.numpy(), gq.numpy()


sim, cfg = defaults.default_configs_for("dna2")
flat = fp.pack_flat(fp.derive_flat(2, cfg, kt=sim["kT"], salt_conc=0.5, half_charged_ends=True), _lib.param_names())
print("extent", c0.min(0), c0.max(0))
for label, shift in (("as generated", np.zeros(3)), ("centred", -0.5 * (c0.min(0) + c0.max(0)))):
    cs32 = (c0 + shift).astype(np.float32).astype(np.float64)  # what both precisions are given: fp32-representable
    gc_ref, gq_ref = reference(cs32)
    for dtype in (torch.float64, torch.float32):
        s = OxdnaSystem(2, top.seq, top.is_end, top.bonded_neighbors, dtype=dtype)
        s.set_params(flat)
        cs = cs32
        cd = torch.as_tensor(cs, dtype=dtype, device=s.device).contiguous()
        qd = torch.as_tensor(q0, dtype=dtype, device=s.device).contiguous()
        s.build_neighbors(cd, 3.25, 0.0)
        e, gc, gq, _ = s.energy(cd, qd, grads=True)
        g = gc.cpu().double().numpy().reshape(-1, 3)
        err = np.abs(g - gc_ref).max(1)
        mag = np.abs(gc_ref).max(1)
        worst = np.argsort(-err)[:4]
        print(f"{label:13s} {str(dtype):14s} max err {err.max():.3e}  max |F| {mag.max():.1f}  rms |F| {np.sqrt((gc_ref**2).mean()):.2f}  "
              f"worst rel-to-own {np.max(err / (mag + 1e-9)):.2e}")
        for w in worst:
            print(f"     nt {w:6d} x={cs[w, 0]:9.2f} err {err[w]:.3e} |F| {mag[w]:.2f}")
        gqd = gq.cpu().double().numpy().reshape(-1, 4)
        eq = np.abs(gqd - gq_ref).max(1)
        wq = np.argsort(-eq)[:3]
        # split the error of dU/dq into the part along q (a change of |q|: no torque) and the tangential part
        rad = ((gqd - gq_ref) * q0).sum(1) / (q0 * q0).sum(1)
        print(f"     dU/dq: max err {eq.max():.3e} max |gq| {np.abs(gq_ref).max():.1f}; along q {np.abs(rad).max():.3e}; "
              f"|q|-1 max {np.abs(np.linalg.norm(q0, axis=1) - 1).max():.2e}")
        for w in wq:
            print(f"     nt {w:6d} gq err {eq[w]:.3e} gq_ref {gq_ref[w]} got {gqd[w]}")
