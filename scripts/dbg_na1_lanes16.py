#!/usr/bin/env python3
"""dev: oxNA fp64 fused step kernel at 16 lanes per nucleotide against the oracle, step by step, with and without the energy trace."""
import sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from mythos_amd import _lib
from mythos_amd.hip_system import LangevinIntegrator
from oracle.langevin_oracle import LangevinOracle
from tests import helpers as H
from tests import test_gpu_na1 as T

name = "simple-helix-dna-rna"
top, traj, _, is_rna = H.load_golden_na1(name)
kT = 296.15 * 0.1 / 300.0
gam_t, gam_r, seed = kT / 2.5, kT / 7.5, 0xBADC0FFEE
for lanes in (8, 16):
    _lib.debug_set("md_lanes", lanes)
    s = T._system(top, is_rna, traj.box_size, torch.float64)
    for save in (True, False):
        integ = LangevinIntegrator(s, dt=0.003, kT=kT, gamma_t=gam_t, gamma_r=gam_r, mass=1.0, inertia=(1.0, 1.2, 0.9), seed=seed)
        c = torch.as_tensor(traj.center[2], device=s.device).contiguous(); q = torch.as_tensor(traj.quaternions[2], device=s.device).contiguous()
        p, L = integ.init_momenta()
        x, qq, pp, LL = (t.cpu().numpy().copy() for t in (c, q, p, L))
        lo = LangevinOracle(4, H.oracle_params_na1(), H.topo_tensors(top), traj.box_size, 0.003, kT, gam_t, gam_r, 1.0, (1.0, 1.2, 0.9), seed=seed, is_rna=is_rna)
        integ.load(c, q, p, L)
        errs = []
        for k in range(6):
            if save:
                tc, tq, et = integ.advance(1, save_every=1)
            else:
                integ.advance(1)
            integ.store(c, q, p, L)
            x, qq, pp, LL, u = lo.step(x, qq, pp, LL)
            errs.append(float(np.abs(c.cpu().numpy() - x).max()))
        print("lanes", lanes, "save", save, "max |dx| per step", ["%.1e" % e for e in errs])
_lib.debug_set("md_lanes", 0)
