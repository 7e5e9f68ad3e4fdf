import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import fem_shell_amd as fsa
from fem_shell_amd import batch
from tests import util
E = int(sys.argv[1]) if len(sys.argv) > 1 else 1250000
props = fsa.build_prop_table(util.standard_props())
rng = np.random.default_rng(0)
xyz = (util.BASE_QUAD[None] + 0.02*rng.standard_normal((E,4,3)))*1e-3 + rng.uniform(0,3,(E,1,3))
pid = ((np.arange(E)//4096) % len(props)).astype(np.int32)
d = torch.device('cuda:0')
x = torch.from_numpy(xyz).to(d); p = torch.from_numpy(pid).to(d); pd = batch.to_device_props(props)
for want in (dict(want_K=True), dict(want_K=False, want_M=True)):
    for _ in range(3): batch.quad_batch(x,p,pd,len(props),status=False,**want)
    torch.cuda.synchronize(); t=time.time(); n=10
    for _ in range(n): batch.quad_batch(x,p,pd,len(props),status=False,**want)
    torch.cuda.synchronize(); dt=(time.time()-t)/n
    print(want, f"{dt*1e3:.3f} ms  {E/dt:.3e} el/s  write {E*2880/dt/1e12:.2f} TB/s  algflops {E*27000/dt/1e12:.2f} TF")
