"""Times the symbolic phase (host BlockPlan vs device fs_csr_symbolic_*) on the bench strip."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
import bench
from fem_shell_amd import meshes, symbolic
mesh = meshes.jittered_plate(bench.STRIP_NX, bench.STRIP_NY, cell=1e-3, seed=11, n_sets=16)
symbolic.DeviceBlockPlan(mesh.conn[:1000], mesh.node_count, 6, "cuda:0")
torch.cuda.synchronize()
for _ in range(3):
    t = time.perf_counter(); dp = symbolic.DeviceBlockPlan(mesh.conn, mesh.node_count, 6, "cuda:0"); torch.cuda.synchronize()
    td = time.perf_counter() - t
    print(f"device symbolic: {td*1e3:.1f} ms  ({mesh.conn.shape[0]/td:.3e} el/s), nblk {dp.nblk}")
t = time.perf_counter(); hp = symbolic.BlockPlan(mesh.conn, mesh.node_count, 6); th = time.perf_counter() - t
print(f"host symbolic: {th*1e3:.1f} ms")
assert np.array_equal(dp.contrib.cpu().numpy(), hp.contrib) and np.array_equal(dp.nbr.cpu().numpy(), hp.nbr)
