"""Fused evaluate-and-assemble kernel vs element kernel + assembly kernel on the C5 strip (bench workload).
usage: python tools/fused_bench.py [nx ny] [max_patch_elems]"""
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, '.')
import bench
from fem_shell_amd import meshes
from fem_shell_amd.assembler import MeshAssembler
from fem_shell_amd.elements import ElementFamily

nx = int(sys.argv[1]) if len(sys.argv) > 2 else bench.STRIP_NX
ny = int(sys.argv[2]) if len(sys.argv) > 2 else bench.STRIP_NY
mpe = int(sys.argv[3]) if len(sys.argv) > 3 else 64
lams = bench.c5_laminates()
import os
if os.environ.get("FUSED_BENCH_UNCOUPLED"):
    lams = [lams[i] for i in (0, 1, 2, 3, 8, 9, 10, 11, 12, 13, 14, 15, 0, 1, 2, 3)]     # symmetric stacks only
mesh = meshes.jittered_plate(nx, ny, cell=1e-3, seed=11, n_sets=16)
asm = MeshAssembler(mesh, {"elements": {"element_family": ElementFamily.SHELL,
                                        "properties": {mesh.set_names[i]: lams[i] for i in range(16)}}})
asm.fused_max_patch_elems = mpe
E = mesh.element_count
t0 = time.time(); fp = asm.fused_plan(); t_plan = time.time() - t0
print("fused plan: %.2f s" % t_plan, json.dumps({k: round(v, 3) for k, v in fp.stats().items()}))


def timed(fn, n=10):
    # (the first ~25 launches after an idle period run up to 20 % slower -- clocks / power management; FUSED_BENCH_PREWARM=60
    # times the steady state instead)
    for _ in range(int(os.environ.get("FUSED_BENCH_PREWARM", "3"))): fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(n): fn()
    ev[1].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / n


def two():
    asm.refresh_element_matrices(want_K=True, want_M=False, check=False)
    return asm.assemble_stiffness_matrix(recompute=False)


if os.environ.get("FUSED_BENCH_ONLY"):
    print("fused only: %.3f ms" % timed(asm.assemble_stiffness_fused, n=int(os.environ["FUSED_BENCH_ONLY"])))
    if os.environ.get("FUSED_BENCH_MASS"):
        print("fused mass: %.3f ms" % timed(asm.assemble_mass_fused, n=int(os.environ["FUSED_BENCH_ONLY"])))
    if os.environ.get("FUSED_BENCH_STATUS"):       # the same with the status words written (what a bench step does)
        st = torch.zeros(E, dtype=torch.int32, device="cuda")
        print("fused with status: %.3f ms" % timed(lambda: asm.assemble_stiffness_fused(status=st), n=int(os.environ["FUSED_BENCH_ONLY"])))
    sys.exit(0)
t2 = timed(two)
tf = timed(asm.assemble_stiffness_fused)
K2, Kf = two(), asm.assemble_stiffness_fused()
same = bool((K2.data.view(torch.int64) == Kf.data.view(torch.int64)).all())
d_ = (K2.data - Kf.data).abs().max().item() / K2.data.abs().max().item()
print("max |fused - two-kernel| / max|K| = %.2e, symmetric: %s" % (d_, "n/a"))
print(f"E={E} two-kernel {t2:.3f} ms  fused {tf:.3f} ms  ({E / tf * 1e3:.3e} el/s)  bitwise equal: {same}")
print(f"fused step roofline: {E * 2708 / tf * 1e3 / 1e12:.2f} TB/s of 8 ({E * 2708 / tf * 1e3 / 8e12:.3f}), "
      f"{E * 27750 / tf * 1e3 / 1e12:.1f} TFLOP/s of 78.6 ({E * 27750 / tf * 1e3 / 78.6e12:.3f})")


def two_m():
    asm.refresh_element_matrices(want_K=False, want_M=True, check=False)
    return asm.assemble_mass_matrix(recompute=False, fused=False)


tm2, tmf = timed(two_m), timed(asm.assemble_mass_fused)
M2, Mf = two_m(), asm.assemble_mass_fused()
print(f"mass: two-kernel {tm2:.3f} ms  fused {tmf:.3f} ms  bitwise equal: {bool((M2.data.view(torch.int64) == Mf.data.view(torch.int64)).all())}")
