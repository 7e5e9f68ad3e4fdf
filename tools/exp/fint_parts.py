"""Parts of f_int on the assembled matrix (1.25 M elements): python tools/exp/fint_parts.py"""
import sys, torch
sys.path.insert(0, '.')
import bench
import fem_shell_amd as fsa
from fem_shell_amd import meshes, _lib
from fem_shell_amd.assembler import MeshAssembler
from fem_shell_amd.elements import ElementFamily
mat = fsa.IsotropicMaterial("cant", 1.2e6, 0.0, 1.0)
lib = _lib.load()
mesh = meshes.flat_patch(0.0, 10.0, 0.0, 1.0 * 3125 / 40, 400, 3125)
asm = MeshAssembler(mesh, {"elements": {"element_family": ElementFamily.SHELL, "material": mat, "thickness": 0.1,
                                        "nonlinear": True}}, cache_blocks=False)
xs = torch.from_numpy(mesh.coords[:, 0]).to(asm.device)
field = torch.zeros((mesh.node_count, 6), dtype=torch.float64, device=asm.device)
field[:, 2] = xs ** 2 / 100.0; field[:, 4] = -2.0 * xs / 100.0
u = (field * 1e-3).reshape(-1).contiguous()
k_lin = asm._linear_csr()
g, pd = asm._groups[0], asm._plan_d
fe = torch.empty((g["E"], 24), dtype=torch.float64, device=asm.device)
def fnl():
    _lib.check(lib.fs_shell_quad_fint_nl(_lib.ptr(g["xyz_d"]), _lib.ptr(g["conn_d"]), _lib.ptr(u), _lib.ptr(g["pid_d"]),
                                         _lib.ptr(asm._props_d), len(asm._prop_table), g["E"], _lib.ptr(fe), _lib.stream_ptr()), "fnl")
fnl()
f_nl = asm._scatter_vector(fe.reshape(-1))
f = torch.empty_like(u)
def mv():
    _lib.check(lib.fs_csr_blk_matvec(_lib.ptr(pd["nbr_ptr"]), _lib.ptr(pd["nbr"]), _lib.ptr(pd["blk_row"]), int(asm._plan.n_nodes), 6,
                                     _lib.ptr(k_lin), _lib.ptr(u), _lib.ptr(f_nl), _lib.ptr(f), _lib.stream_ptr()), "mv")
print("fnl kernel %.4f ms" % bench.avg_ms(torch, fnl, 20))
print("vector scatter %.4f ms" % bench.avg_ms(torch, lambda: asm._scatter_vector(fe.reshape(-1)), 20))
print("matvec %.4f ms" % bench.avg_ms(torch, mv, 20))
print("f_int total %.4f ms" % bench.avg_ms(torch, lambda: asm._forces_on_assembled(u, k_lin), 20))
for fl in (0, 1):
    asm.tangent_flags = fl
    print("K_T flags", fl, "%.4f ms" % bench.avg_ms(torch, lambda: asm._tangent_on_assembled(u, k_lin), 20))
