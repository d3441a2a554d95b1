import sys; sys.path.insert(0,'.')
import numpy as np, torch
from mythos_amd import _lib
from mythos_amd.energy import flat_params as fp
from mythos_amd.input import defaults
from mythos_amd.hip_system import OxdnaSystem
from mythos_amd.utils import generators
top, c0, q0 = generators.ideal_duplex(400, seed=5)
sim, cfg = defaults.default_configs_for("dna2")
flat = fp.pack_flat(fp.derive_flat(2, cfg, kt=sim["kT"], half_charged_ends=True), _lib.param_names())
s = OxdnaSystem(2, top.seq, top.is_end, top.bonded_neighbors, box=None, dtype=torch.float64)
s.set_params(flat)
c = torch.as_tensor(c0, dtype=torch.float64, device=s.device); q = torch.as_tensor(q0, dtype=torch.float64, device=s.device)
s.build_neighbors(c, 3.25, 0.4)
e1 = s.energy(c,q)[0]; e2 = s.energy(c,q)[0]
print('same list twice equal:', torch.equal(e1,e2), (e1-e2).abs().max().item())
for k in range(4):
    s.build_neighbors(c, 3.25, 0.4); e3 = s.energy(c,q)[0]
    print('rebuild equal:', torch.equal(e1,e3), (e1-e3).abs().max().item(), s.neighbor_stats())
