"""Where the set-up time of the fused HEXA8 plan goes (configs[4]'s block): python tools/exp/hex_plan_time.py"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import fem_shell_amd as fsa
from fem_shell_amd import meshes, fused
from fem_shell_amd.assembler import MeshAssembler
from fem_shell_amd.elements import ElementFamily
from types import SimpleNamespace
nx, ny, nz = 128, 128, 64
ids = np.arange((nx + 1) * (ny + 1) * (nz + 1), dtype=np.int64).reshape(nz + 1, ny + 1, nx + 1)
Z, Y, X = np.meshgrid(np.arange(nz + 1.0), np.arange(ny + 1.0), np.arange(nx + 1.0), indexing="ij")
coords = np.stack([X, Y, Z], -1).reshape(-1, 3) * 1e-3
coords += (meshes.hash_uniform(np.arange(coords.size), 5).reshape(-1, 3) - 0.5) * 1e-4
n = lambda dz, dy, dx: ids[dz:nz + dz, dy:ny + dy, dx:nx + dx]
conn = np.stack([n(0, 0, 0), n(0, 0, 1), n(0, 1, 1), n(0, 1, 0), n(1, 0, 0), n(1, 0, 1), n(1, 1, 1), n(1, 1, 0)], -1).reshape(-1, 8)
mat = fsa.OrthotropicMaterial("T300/5208", (181e9, 10.3e9, 10.3e9), (7.17e9, 3.0e9, 7.17e9), (0.28, 0.28, 0.28), 1600.0)
t0 = time.perf_counter()
asm = MeshAssembler(meshes.ArrayMesh(coords, conn), {"elements": {"element_family": ElementFamily.SOLID, "material": mat}})
torch.cuda.synchronize(); print(f"assembler constructor {time.perf_counter() - t0:.3f} s")
for rep in range(2):
    t = [time.perf_counter()]
    patch = fused.partition_nodes_hex(asm._coords, asm._conn); t.append(time.perf_counter())
    p = asm._plan
    host = SimpleNamespace(d=p.d, n_nodes=p.n_nodes, nbr_ptr=p.nbr_ptr.cpu().numpy(), nbr=p.nbr.cpu().numpy()); t.append(time.perf_counter())
    fp = fused.FusedHexPlan(host, asm._conn, asm._coords, node_patch=patch); t.append(time.perf_counter())
    d = fp.to_device(asm.device); torch.cuda.synchronize(); t.append(time.perf_counter())
    names = ["partition_nodes_hex (NumPy)", "node graph D2H", "native builder + export", "upload"]
    print(f"rep {rep}: " + ", ".join(f"{n} {b - a:.3f} s" for n, a, b in zip(names, t[:-1], t[1:])), f"total {t[-1] - t[0]:.3f} s")
