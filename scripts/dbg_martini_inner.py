#!/usr/bin/env python3
"""dev: the pruned rows written by the plain and by the energy-trace instantiation of the MARTINI step kernel."""
import sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from tests import test_gpu_martini_md as T
from mythos_amd.hip_system import MartiniLangevinIntegrator

dtype = torch.float64
sysm, *_rest, x0, b0 = T._make(dtype)
def fresh(inner=(0.2, 4)):
    integ = MartiniLangevinIntegrator(sysm, dt=0.02, kT=T.KB * T.T, gamma=1.0, seed=23)
    integ.set_neighbor_policy(0.3, 6)
    integ.set_inner_list(*inner)
    pos = torch.as_tensor(x0, dtype=dtype, device=sysm.device).contiguous()
    vel = integ.init_velocities()
    integ.load(pos, vel, b0)
    return integ
a = fresh(); a.advance(5, save_every=1, want_energy=False)   # launches 0..4, launch 4 prunes (plain)
b = fresh(); b.advance(5, save_every=5)                      # launches 0..5 (closing), launch 5 = energy-trace walk; launch 4 plain
c = fresh(); c.advance(4, save_every=4)                      # launches 0..4, launch 4 = energy-trace + closing + prune
ra, la = a.rows(True); rb, lb = b.rows(True); rc, lc = c.rows(True)
ro, lo = a.rows(False)
print("lens equal a/b", np.array_equal(la, lb), " a/c", np.array_equal(la, lc), "mean", la.mean(), lo.mean())
bad = np.nonzero(la != lc)[0]
print("beads whose pruned length differs a/c:", bad[:10], [(int(la[i]), int(lc[i])) for i in bad[:10]])
same = [i for i in range(len(la)) if la[i] == lc[i] and not np.array_equal(ra[i, :la[i]], rc[i, :lc[i]])]
print("same length, other content:", same[:10])
if len(bad):
    i = bad[0]
    print("a:", ra[i, :la[i]].tolist()); print("c:", rc[i, :lc[i]].tolist()); print("outer:", ro[i, :lo[i]].tolist())
