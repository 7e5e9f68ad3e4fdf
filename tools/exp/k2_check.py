"""A/B of the two-wave stiffness kernel (FEMSHELL_K2=1) against mitc4_k_kernel: bitwise comparison + timing."""
import os, sys, time, subprocess
import numpy as np, torch
sys.path.insert(0, '.')
import fem_shell_amd as fsa
from fem_shell_amd import batch, _lib
from tests import util
E = int(sys.argv[1]) if len(sys.argv) > 1 else 1250000
props = fsa.build_prop_table(util.standard_props())
rng = np.random.default_rng(0)
xyz = (util.BASE_QUAD[None] + 0.02*rng.standard_normal((E,4,3)))*1e-3 + rng.uniform(0,3,(E,1,3))
pid = ((np.arange(E)//37) % len(props)).astype(np.int32)
d = torch.device('cuda:0')
x = torch.from_numpy(xyz).to(d); p = torch.from_numpy(pid).to(d); pd = batch.to_device_props(props)
layout = _lib.FS_LAYOUT_SYMBLK_T16
out = torch.zeros((batch.n_blocks(E, 10, layout), 36), dtype=torch.float64, device=d)
f = lambda: batch.quad_batch(x,p,pd,len(props),status=True,validate=False,K_out=out,layout=layout)
r = f()
for _ in range(3): f()
torch.cuda.synchronize(); t=time.time(); n=20
for _ in range(n): f()
torch.cuda.synchronize(); dt=(time.time()-t)/n
print("K2" if os.environ.get("FEMSHELL_K2") == "1" else "K1", f"{dt*1e3:.3f} ms  {E/dt:.3e} el/s  algflops {E*27000/dt/1e12:.2f} TF  status nonzero {int((r['status']!=0).sum())}")
torch.save(out.cpu(), f"/tmp/k_{os.environ.get('FEMSHELL_K2','0')}.pt")
if os.environ.get("FEMSHELL_K2") == "1" and os.path.exists("/tmp/k_0.pt"):
    a = torch.load("/tmp/k_0.pt"); b = out.cpu()
    print("bitwise equal:", bool(torch.equal(a, b)), "max abs diff", float((a-b).abs().max()), "nan", int(torch.isnan(b).sum()))
