"""A/B of the d = 3 assembly kernels on configs[4]'s HEXA8 block (PAIR_ORDER=1: descriptors ordered so that (I,J) and (J,I) are neighbours -- slower, 2.39 vs 2.23 ms).
python tools/exp/hex_bench.py [nx ny nz] -> assembly ms and a checksum of the CSR values (bitwise comparison)."""
import hashlib, sys, numpy as np, torch
sys.path.insert(0, '.')
import bench
import fem_shell_amd as fsa
from fem_shell_amd import meshes
from fem_shell_amd.assembler import MeshAssembler
from fem_shell_amd.elements import ElementFamily
nx, ny, nz = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (128, 128, 64)
ids = np.arange((nx + 1) * (ny + 1) * (nz + 1), dtype=np.int64).reshape(nz + 1, ny + 1, nx + 1)
Z, Y, X = np.meshgrid(np.arange(nz + 1.0), np.arange(ny + 1.0), np.arange(nx + 1.0), indexing="ij")
coords = np.stack([X, Y, Z], -1).reshape(-1, 3) * 1e-3
coords += (meshes.hash_uniform(np.arange(coords.size), 5).reshape(-1, 3) - 0.5) * 1e-4
n = lambda dz, dy, dx: ids[dz:nz + dz, dy:ny + dy, dx:nx + dx]
conn = np.stack([n(0, 0, 0), n(0, 0, 1), n(0, 1, 1), n(0, 1, 0), n(1, 0, 0), n(1, 0, 1), n(1, 1, 1), n(1, 1, 0)], -1).reshape(-1, 8)
mat = fsa.IsotropicMaterial("steel", 2.1e11, 0.3, 7800.0)
asm = MeshAssembler(meshes.ArrayMesh(coords, conn), {"elements": {"element_family": ElementFamily.SOLID, "material": mat}})
K = asm.assemble_stiffness_matrix(recompute=True)
import os
if os.environ.get("PAIR_ORDER"):      # descriptors ordered so that (I,J) and (J,I) are neighbours: shared source blocks meet in L2
    pl = asm._plan
    I, J = pl.blk_row.long(), pl.nbr.long()
    key = torch.minimum(I, J) * (2 * pl.n_nodes) + torch.maximum(I, J) * 2 + (I > J).long()
    perm = torch.argsort(key)
    asm._desc_d = asm._desc_d.view(-1, 48)[perm].contiguous().view(-1)
a_ms = bench.avg_ms(torch, lambda: asm.assemble_stiffness_matrix(recompute=False), 10)
K = asm.assemble_stiffness_matrix(recompute=False)
h = hashlib.sha1(K.data.cpu().numpy().tobytes()).hexdigest()[:16]
E = conn.shape[0]
print(f"E={E} nnz={K.nnz} assembly {a_ms:.3f} ms  alg {(36*72*E + 8*K.nnz)/a_ms/1e9:.2f} TB/s  sha {h}")
