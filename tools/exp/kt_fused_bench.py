"""fs_fused_assemble_quad_tangent against the two-kernel Newton route (element kernel on the cached linear blocks +
descriptor assembly): python tools/exp/kt_fused_bench.py [small]"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import bench
import fem_shell_amd as fsa
from fem_shell_amd import meshes, _lib
from fem_shell_amd.assembler import MeshAssembler
from fem_shell_amd.elements import ElementFamily
mat = fsa.IsotropicMaterial("cant", 1.2e6, 0.3, 1.0)
lib = _lib.load()
sizes = ((40, 30),) if "small" in sys.argv else ((40, 30), (400, 3125))
for nx, ny in sizes:
    mesh = meshes.flat_patch(0.0, 10.0, 0.0, 1.0 * ny / 40, nx, ny)
    asm = MeshAssembler(mesh, {"elements": {"element_family": ElementFamily.SHELL, "material": mat, "thickness": 0.1,
                                            "nonlinear": True}})
    xs = torch.from_numpy(mesh.coords[:, 0]).to(asm.device)
    ys = torch.from_numpy(mesh.coords[:, 1]).to(asm.device)
    field = torch.zeros((mesh.node_count, 6), dtype=torch.float64, device=asm.device)
    field[:, 0] = 0.01 * xs * ys; field[:, 1] = -0.02 * ys + 0.003 * xs
    field[:, 2] = xs ** 2 / 100.0; field[:, 4] = -2.0 * xs / 100.0
    u = (field * 0.3).reshape(-1).contiguous()
    asm.newton_on_assembled = False
    KT_ref = asm.assemble_tangent_stiffness(u)
    asm.newton_on_assembled = True
    K = asm.assemble_stiffness_matrix(recompute=True)
    asm._tangent_on_assembled(u, K.data)          # builds pat_conn
    fp = asm.fused_plan(); d = asm._fplan_d; g = asm._groups[0]
    out = torch.empty_like(K.data)
    for flags in (0, 1):
        asm.tangent_flags = flags
        run = lambda: asm._tangent_on_assembled(u, K.data)
        out = run().data; torch.cuda.synchronize()
        ref = KT_ref.data
        err = (out - ref).abs().max().item() / ref.abs().max().item()
        dk = (ref - K.data).abs().max().item() / ref.abs().max().item()
        ms = bench.avg_ms(torch, run, 20)
        print(f"{nx}x{ny} flags={flags}: max|KT - ref|/max|ref| = {err:.3e} (correction itself {dk:.3e}), nan {int(torch.isnan(out).sum())}, {ms:.4f} ms")
    asm.newton_on_assembled = False
    ms2 = bench.avg_ms(torch, lambda: asm.assemble_tangent_stiffness(u), 10)
    print(f"  two-kernel K_T: {ms2:.4f} ms")
