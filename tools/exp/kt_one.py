"""A few launches of the Newton kernels on the 1.25 M strip (for rocprofv3 passes): python3 tools/exp/kt_one.py [flags]"""
import sys, torch
sys.path.insert(0, '.')
import fem_shell_amd as fsa
from fem_shell_amd import meshes
from fem_shell_amd.assembler import MeshAssembler
from fem_shell_amd.elements import ElementFamily
mat = fsa.IsotropicMaterial("cant", 1.2e6, 0.0, 1.0)
mesh = meshes.flat_patch(0.0, 10.0, 0.0, 1.0 * 3125 / 40, 400, 3125)
asm = MeshAssembler(mesh, {"elements": {"element_family": ElementFamily.SHELL, "material": mat, "thickness": 0.1,
                                        "nonlinear": True}}, cache_blocks=False)
if len(sys.argv) > 1:
    asm.tangent_flags = int(sys.argv[1])
xs = torch.from_numpy(mesh.coords[:, 0]).to(asm.device)
field = torch.zeros((mesh.node_count, 6), dtype=torch.float64, device=asm.device)
field[:, 2] = xs ** 2 / 100.0; field[:, 4] = -2.0 * xs / 100.0
u = (field * 1e-3).reshape(-1).contiguous()
for k in range(3):
    asm.assemble_tangent_and_forces(u)
asm.assemble_tangent_stiffness(u)
asm.assemble_internal_forces(u)
torch.cuda.synchronize()
