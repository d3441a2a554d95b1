#!/usr/bin/env python3
"""dev: fp64 energies of a 1 kbp duplex through MODE 0 / 1 / 2 of the energy kernel, device-built list (they must agree)."""
import sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from mythos_amd import _lib
from mythos_amd.energy import flat_params as fp
from mythos_amd.hip_system import OxdnaSystem
from mythos_amd.input import defaults
from mythos_amd.utils import generators
dev = torch.device("cuda", 0)
sim, cfg = defaults.default_configs_for("dna2")
flat = fp.pack_flat(fp.derive_flat(2, cfg, kt=sim["kT"], salt_conc=0.5, half_charged_ends=True), _lib.param_names())
for bp in (32, 1000):
    top1, c1, q1 = generators.ideal_duplex(bp, model=2, seed=1234)
    ref = None
    for dtype in (torch.float32, torch.float64):
        s = OxdnaSystem(2, top1.seq, top1.is_end, top1.bonded_neighbors, box=None, dtype=dtype, device=dev)
        s.set_params(flat)
        c = torch.as_tensor(c1, dtype=dtype, device=dev); q = torch.as_tensor(q1, dtype=dtype, device=dev)
        for how in ("build", "pairs"):
            if how == "build":
                s.build_neighbors(c, 3.25, 0.1)
            else:
                if bp > 100: continue
                s.set_neighbors(top1.unbonded_neighbors)
            e0 = s.energy(c, q)[0].cpu().numpy()
            e1 = s.energy(c, q, grads=True)[0].cpu().numpy()
            e2 = s.energy(c, q, grads=True, param_grads=True)[0].cpu().numpy()
            e1b = s.energy(c, q, grads=True)[0].cpu().numpy()
            print(bp, dtype, how, "sum e0", e0.sum().round(4), "|e1-e0|", np.abs(e1 - e0).max(), "|e2-e0|", np.abs(e2 - e0).max(), "|e1b-e1|", np.abs(e1b - e1).max())
