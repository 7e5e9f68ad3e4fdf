"""MITC3+ hemisphere 256 x 256 (configs[1]): element kernel + assembly timing and a checksum of the CSR values."""
import hashlib, sys, torch
sys.path.insert(0, '.')
import bench
import fem_shell_amd as fsa
from fem_shell_amd import meshes
from fem_shell_amd.assembler import MeshAssembler
from fem_shell_amd.elements import ElementFamily
mat = fsa.IsotropicMaterial("Ko2017_Sphere", 6.825e7, 0.3, 1.0)
mesh = meshes.sphere_patch(10.0, 18.0, 90.0, 90.0, 256, triangular=True)
asm = MeshAssembler(mesh, {"elements": {"element_family": ElementFamily.SHELL, "material": mat, "thickness": 0.04}})
K = asm.assemble_stiffness_matrix(recompute=True)
k_ms = bench.avg_ms(torch, lambda: asm.refresh_element_matrices(want_K=True, want_M=False, check=False), 20)
a_ms = bench.avg_ms(torch, lambda: asm.assemble_stiffness_matrix(recompute=False), 20)
K = asm.assemble_stiffness_matrix(recompute=False)
print(f"E={mesh.element_count} element kernel {k_ms:.4f} ms  assembly {a_ms:.4f} ms  sha {hashlib.sha1(K.data.cpu().numpy().tobytes()).hexdigest()[:16]}")
