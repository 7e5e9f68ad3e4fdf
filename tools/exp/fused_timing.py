"""Phase timeline of the fused kernel (library built with -DFS_FUSED_TIMING): shader-clock cycles summed
over all waves per phase.  FEMSHELL_HIP_LIB=tools/exp/variants/lib_timing.so python tools/exp/fused_timing.py"""
import ctypes, sys, numpy as np, torch
sys.path.insert(0, '.')
import bench
from fem_shell_amd import meshes, _lib
from fem_shell_amd.assembler import MeshAssembler
from fem_shell_amd.elements import ElementFamily
lams = bench.c5_laminates()
mesh = meshes.jittered_plate(bench.STRIP_NX, bench.STRIP_NY, cell=1e-3, seed=11, n_sets=16)
asm = MeshAssembler(mesh, {"elements": {"element_family": ElementFamily.SHELL,
                                        "properties": {mesh.set_names[i]: lams[i] for i in range(16)}}})
asm.fused_plan()
lib = ctypes.CDLL(_lib.LIB_PATH)
out = (ctypes.c_ulonglong * 24)()
fn = asm.assemble_mass_fused if 'mass' in sys.argv else asm.assemble_stiffness_fused
for _ in range(3): fn()
torch.cuda.synchronize()
lib.fs_debug_fused_timing(None, 1)
ms = bench.avg_ms(torch, fn, 5)
torch.cuda.synchronize()
lib.fs_debug_fused_timing(out, 0)
t = np.array(list(out)[:7], dtype=np.float64)
t2 = np.array(list(out)[14:17], dtype=np.float64)
t[5] += t2.sum()
t1 = np.array(list(out)[7:14], dtype=np.float64)
names = ["stage 1 (+ record fetch issue)", "barrier after stage 1", "header prefetch, record decode, evaluation 0", "evaluation 1",
         "element-order sums (DPP)", "staging + stores", "end-of-patch barrier + hand-over"]
print(f"kernel {ms:.3f} ms")
for n, v in zip(names, t):
    print(f"{n:48s} {100 * v / t.sum():5.1f} %   {ms * v / t.sum():.3f} ms")
names1 = ["geometry", "tying tangents", "Jacobian store, blend, W, S partial", "coupling cores", "shear constants", "bubble + Knb partials and sums", "S / Kbb sums, Kbb^-1"]
print("inside stage 1:")
for n, v in zip(names1, t1):
    print(f"  {n:46s} {100 * v / t.sum():5.1f} %   {ms * v / t.sum():.3f} ms")
print(f"  {'property-row loads issued (phase 7)':46s} {100 * out[17] / t.sum():5.1f} %")
print(f"  {'loop top .. quad_stage1 (prefetch issue, waits)':46s} {100 * out[18] / t.sum():5.1f} %")
print("inside staging + stores:")
for n, v in zip(["slot bookkeeping before the passes", "staging writes (+ wave sync)", "read-back + global stores (+ wave sync)"], t2):
    print(f"  {n:46s} {100 * v / t.sum():5.1f} %   {ms * v / t.sum():.3f} ms")
