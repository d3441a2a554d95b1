#!/usr/bin/env python3
"""oxNA energy path: the reference's hybrid goldens (16 nt, 100 frames tiled to 6 400) through the model-4 instantiation -
all-DNA, all-RNA and DNA-RNA duplexes (the last one diverges over the three code paths) - with the all-DNA duplex through
the oxDNA2 instantiation beside them.  ms per call of energies, + forces, + dU/dtheta."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from mythos_amd import _lib  # noqa: E402
from mythos_amd.energy import flat_params as fp  # noqa: E402
from mythos_amd.hip_system import OxdnaSystem  # noqa: E402
from mythos_amd.input import defaults  # noqa: E402
from tests import helpers as H  # noqa: E402  (golden loader only)

sim, cfg = defaults.default_configs_for("na1")
flat4 = fp.pack_flat_na1(fp.derive_flat_na1(cfg["dna"], cfg["rna"], cfg["drh"], kt=sim["kT"], salt_conc=0.5, half_charged_ends=False), _lib.param_names())
flat2 = fp.pack_flat(fp.derive_flat(2, cfg["dna"], kt=sim["kT"], salt_conc=0.5, half_charged_ends=False), _lib.param_names())
for dtype in (torch.float32, torch.float64):
    for name, model in (("simple-helix-dna-dna", 2), ("simple-helix-dna-dna", 4), ("simple-helix-rna-rna", 4), ("simple-helix-dna-rna", 4)):
        top, traj, _, is_rna = H.load_golden_na1(name)
        s = OxdnaSystem(model, top.seq, top.is_end, top.bonded_neighbors, box=traj.box_size, dtype=dtype, is_rna=is_rna if model == 4 else None)
        s.set_params(flat4 if model == 4 else flat2)
        s.set_neighbors(top.unbonded_neighbors)
        c = torch.as_tensor(np.tile(traj.center, (64, 1, 1)), dtype=dtype, device=s.device)
        q = torch.as_tensor(np.tile(traj.quaternions, (64, 1, 1)), dtype=dtype, device=s.device)
        row = []
        for kw in ({}, {"grads": True}, {"grads": True, "param_grads": True}):
            s.energy(c, q, **kw)
            torch.cuda.synchronize()
            ts = []
            for _ in range(15):
                t0 = time.perf_counter()
                s.energy(c, q, **kw)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            row.append(round(1e3 * float(np.median(ts)), 3))
        print(f"{str(dtype).split('.')[-1]:8s} model {model} {name:22s} {c.shape[0]} frames x {top.n_nucleotides} nt: {row} ms", flush=True)
