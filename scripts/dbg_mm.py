import numpy as np, torch, sys
sys.path.insert(0, '.')
from tests import test_gpu_martini_md as T
from mythos_amd.hip_system import MartiniLangevinIntegrator
sysm, *_r, x0, b0 = T._make(torch.float32)
def run(chunks, seed, every=5):
    integ = MartiniLangevinIntegrator(sysm, dt=0.02, kT=T.KB*T.T, gamma=1.0, seed=seed)
    integ.set_neighbor_policy(0.3, every)
    pos = torch.as_tensor(x0, dtype=torch.float32, device=sysm.device).contiguous()
    vel = integ.init_velocities()
    for n in chunks: integ.run(pos, vel, b0, n)
    return pos.cpu(), vel.cpu()
for chunks in ([2],[1,1]),([10],[5,5]),([6],[5,1]),([6],[3,3]):
    a=run(chunks[0],2); c=run(chunks[1],2)
    print(chunks, (a[0]-c[0]).abs().max().item(), (a[1]-c[1]).abs().max().item())
