"""Times the MITC3+ stiffness kernel alone (fs_shell_tri_batch, FS_WANT_K) on a 512 x 512 sphere patch of triangles."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
import fem_shell_amd as fsa
from fem_shell_amd import meshes, batch
mesh = meshes.sphere_patch(10.0, 18.0, 90.0, 90.0, 512, triangular=True)
E = mesh.conn.shape[0]
props = fsa.build_prop_table([fsa.ShellProperty(fsa.IsotropicMaterial("s", 6.825e7, 0.3, 1.0), 0.04)])
d = torch.device("cuda:0")
x = torch.from_numpy(mesh.coords[mesh.conn]).to(d)
p = torch.zeros(E, dtype=torch.int32, device=d)
pd = batch.to_device_props(props)
out = torch.empty((E, 6, 36), dtype=torch.float64, device=d)
for _ in range(3):
    batch.shell_batch(3, x, p, pd, 1, want_K=True, status=False, validate=False, K_out=out)
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(10):
    batch.shell_batch(3, x, p, pd, 1, want_K=True, status=False, validate=False, K_out=out)
ev1.record()
torch.cuda.synchronize()
ms = ev0.elapsed_time(ev1) / 10
print(f"MITC3+ K: {E} elements {ms:.3f} ms  {E / ms / 1e6:.3f} G el/s  {E * 12000 / ms / 1e9:.2f} TFLOP/s  write {E * 1728 / ms / 1e9:.2f} TB/s  checksum {float(out.sum()):.6e}")
