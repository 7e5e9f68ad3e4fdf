"""solid_kernel<8> on 1 048 576 jittered bricks: time and checksum of the element blocks."""
import hashlib, sys, numpy as np, torch
sys.path.insert(0, '.')
import bench
import fem_shell_amd as fsa
from fem_shell_amd import batch
from fem_shell_amd.properties import solid_row
rng = np.random.default_rng(0)
E = 1 << 20
ref = np.array([[-1, -1, -1], [1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, -1, 1], [1, -1, 1], [1, 1, 1], [-1, 1, 1]], float)
xyz = (ref[None] + 0.1 * rng.standard_normal((E, 8, 3))) * 1e-3 + rng.uniform(0, 1, (E, 1, 3))
props = torch.from_numpy(solid_row(fsa.IsotropicMaterial("s", 2.1e11, 0.3, 7800.0))[None]).cuda()
x = torch.from_numpy(xyz).cuda(); pid = torch.zeros(E, dtype=torch.int32, device="cuda")
out = batch.solid_batch(8, x, pid, props, 1, want_K=True, status=False, validate=False)
ms = bench.avg_ms(torch, lambda: batch.solid_batch(8, x, pid, props, 1, want_K=True, status=False, validate=False, K_out=out["K"]), 10)
k = out["K"]
print(f"solid_kernel<8> {ms:.3f} ms  sha {hashlib.sha1(k.cpu().numpy().tobytes()).hexdigest()[:16]}")
