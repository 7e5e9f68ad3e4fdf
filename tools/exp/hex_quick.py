"""fused HEXA8 kernel time on configs[4]'s block (no other legs): python tools/exp/hex_quick.py"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import bench
import fem_shell_amd as fsa
from fem_shell_amd import meshes
from fem_shell_amd.assembler import MeshAssembler
from fem_shell_amd.elements import ElementFamily
nx, ny, nz = 128, 128, 64
ids = np.arange((nx + 1) * (ny + 1) * (nz + 1), dtype=np.int64).reshape(nz + 1, ny + 1, nx + 1)
Z, Y, X = np.meshgrid(np.arange(nz + 1.0), np.arange(ny + 1.0), np.arange(nx + 1.0), indexing="ij")
coords = np.stack([X, Y, Z], -1).reshape(-1, 3) * 1e-3
coords += (meshes.hash_uniform(np.arange(coords.size), 5).reshape(-1, 3) - 0.5) * 1e-4
n = lambda dz, dy, dx: ids[dz:nz + dz, dy:ny + dy, dx:nx + dx]
conn = np.stack([n(0, 0, 0), n(0, 0, 1), n(0, 1, 1), n(0, 1, 0), n(1, 0, 0), n(1, 0, 1), n(1, 1, 1), n(1, 1, 0)], -1).reshape(-1, 8)
mat = fsa.OrthotropicMaterial("T300/5208", (181e9, 10.3e9, 10.3e9), (7.17e9, 3.0e9, 7.17e9), (0.28, 0.28, 0.28), 1600.0)
asm = MeshAssembler(meshes.ArrayMesh(coords, conn), {"elements": {"element_family": ElementFamily.SOLID, "material": mat}})
asm.fused_plan()
print("fused hex8 %.3f ms" % bench.avg_ms(torch, asm.assemble_stiffness_fused, 10))
