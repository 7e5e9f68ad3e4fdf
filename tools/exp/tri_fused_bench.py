"""Fused MITC3+ kernel against element kernel + assembly kernel: hemisphere 256 x 256 (configs[1]) and a 1 M-triangle
cylinder: python tools/exp/tri_fused_bench.py"""
import sys, time, torch
sys.path.insert(0, '.')
import bench
import fem_shell_amd as fsa
from fem_shell_amd import meshes
from fem_shell_amd.assembler import MeshAssembler
from fem_shell_amd.elements import ElementFamily
mat = fsa.IsotropicMaterial("Ko2017_Sphere", 6.825e7, 0.3, 1.0)
for name, mesh in (("hemisphere 256", meshes.sphere_patch(10.0, 18.0, 90.0, 90.0, 256, triangular=True)),
                   ("cylinder 1000 x 512", meshes.cylindrical_patch(25.0, 25.0, 40.0, 1000, 512, triangular=True))):
    asm = MeshAssembler(mesh, {"elements": {"element_family": ElementFamily.SHELL, "material": mat, "thickness": 0.04}})
    t0 = time.perf_counter(); asm.fused_plan(); t_plan = time.perf_counter() - t0
    f_ms = bench.avg_ms(torch, lambda: asm.assemble_stiffness_fused(), 20)
    step_f = bench.avg_ms(torch, lambda: asm.assemble_stiffness_matrix(recompute=True), 20)
    step_2 = bench.avg_ms(torch, lambda: asm.assemble_stiffness_matrix(recompute=True, fused=False), 20)
    print(f"{name}: E={mesh.element_count} fused kernel {f_ms:.4f} ms, step fused {step_f:.4f} ms, step two-kernel {step_2:.4f} ms, "
          f"plan {t_plan:.3f} s {asm.fused_plan().stats()}")
