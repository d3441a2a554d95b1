#!/usr/bin/env python3
"""dev: oxNA fp64 energies of the golden hybrid helix through MODE 0 / 1 / 2 of the energy kernel (they must agree)."""
import sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from tests import helpers as H
from tests.test_gpu_na1 import _system
for dtype in (torch.float64, torch.float32):
    for name in ("simple-helix-dna-rna", "simple-coax-rna-rna-rna"):
        top, traj, _, is_rna = H.load_golden_na1(name)
        s = _system(top, is_rna, traj.box_size, dtype)
        c = torch.as_tensor(traj.center[[7, 70]], dtype=dtype, device=s.device)
        q = torch.as_tensor(traj.quaternions[[7, 70]], dtype=dtype, device=s.device)
        e0 = s.energy(c, q)[0].cpu().numpy()
        e1 = s.energy(c, q, grads=True)[0].cpu().numpy()
        e2 = s.energy(c, q, grads=True, param_grads=True)[0].cpu().numpy()
        print(dtype, name, "max|e1-e0|", np.abs(e1 - e0).max(), "max|e2-e0|", np.abs(e2 - e0).max())
        if np.abs(e2 - e0).max() > 1e-3:
            print(" e0", e0[0]); print(" e2", e2[0])
