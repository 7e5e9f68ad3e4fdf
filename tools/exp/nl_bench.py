"""Timing of the Newton re-assembly path (K_T and f_int) at config C4 size and at the C5 strip size."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
import fem_shell_amd as fsa
from fem_shell_amd import meshes
from fem_shell_amd.assembler import MeshAssembler
from fem_shell_amd.elements import ElementFamily

def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3

mat = fsa.IsotropicMaterial("cant", 1.2e6, 0.0, 1.0)
for nx, ny in ((400, 40), (400, 3125)):
    mesh = meshes.flat_patch(0.0, 10.0, 0.0, 1.0 * ny / 40, nx, ny)
    t0 = time.perf_counter()
    asm = MeshAssembler(mesh, {"elements": {"element_family": ElementFamily.SHELL, "material": mat, "thickness": 0.1, "nonlinear": True}})
    torch.cuda.synchronize(); setup = time.perf_counter() - t0
    E = mesh.conn.shape[0]
    u = torch.from_numpy(1e-3 * np.random.default_rng(0).standard_normal(asm.dofs_count)).to(asm.device)
    a = timeit(lambda: asm.assemble_tangent_and_forces(u))
    b = timeit(lambda: asm.assemble_tangent_stiffness(u))
    c = timeit(lambda: asm.assemble_internal_forces(u))
    d = timeit(lambda: asm.assemble_stiffness_matrix(recompute=True))
    m = timeit(lambda: asm.assemble_mass_matrix(recompute=True))
    print(f"{nx}x{ny}={E}: setup {setup*1e3:.1f} ms | K_T+f_int {a:.3f} ms ({E/a*1e3:.3g} el/s) | K_T {b:.3f} | f_int {c:.3f} | K {d:.3f} | M {m:.3f}", flush=True)
