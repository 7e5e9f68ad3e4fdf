import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import fem_shell_amd as fsa
from fem_shell_amd import batch, _lib
from tests import util
E = 1250000
props = fsa.build_prop_table(util.standard_props())
rng = np.random.default_rng(0)
xyz = (util.BASE_QUAD[None] + 0.02*rng.standard_normal((E,4,3)))*1e-3 + rng.uniform(0,3,(E,1,3))
pid = ((np.arange(E)//4096) % len(props)).astype(np.int32)
d = torch.device('cuda:0')
x = torch.from_numpy(xyz).to(d); p = torch.from_numpy(pid).to(d); pd = batch.to_device_props(props)
res = {}
for layout in (_lib.FS_LAYOUT_SYMBLK, _lib.FS_LAYOUT_SYMBLK_T16):
    out = torch.zeros((batch.n_blocks(E, 10, layout), 36), dtype=torch.float64, device=d)
    f = lambda: batch.quad_batch(x,p,pd,len(props),want_K=False,want_M=True,status=False,validate=False,M_out=out,layout=layout)
    for _ in range(3): f()
    torch.cuda.synchronize(); t=time.time(); n=20
    for _ in range(n): f()
    torch.cuda.synchronize(); dt=(time.time()-t)/n
    res[layout] = batch.expand_blocks(out, 4, 6, layout, E)
    print("M layout", layout, f"{dt*1e3:.3f} ms  write {E*2880/dt/1e12:.2f} TB/s")
print("bitwise equal dense:", bool(torch.equal(res[1], res[2])))
