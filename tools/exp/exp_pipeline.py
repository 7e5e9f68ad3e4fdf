"""Experiment: cross-step software pipeline -- element kernel of step s+1 (stream A, buffer (s+1)%2) runs while
the assembly of step s (stream B) reads buffer s%2.  No dependency between the two, so no chunking.
usage: python tools/exp/exp_pipeline.py [steps]"""
import sys, time
import torch
sys.path.insert(0, '.')
import bench
from fem_shell_amd import meshes, _lib
from fem_shell_amd.assembler import MeshAssembler
from fem_shell_amd.elements import ElementFamily

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
lams = bench.c5_laminates()
mesh = meshes.jittered_plate(bench.STRIP_NX, bench.STRIP_NY, cell=1e-3, seed=11, n_sets=16)
model = {"elements": {"element_family": ElementFamily.SHELL,
                      "properties": {mesh.set_names[i]: lams[i] for i in range(16)}}}
asm = MeshAssembler(mesh, model)
E = mesh.element_count
p = asm._plan
lib = _lib.load()
dev = asm.device
ref = asm.assemble_stiffness_matrix(recompute=True).data.clone()
torch.cuda.synchronize()
layout = asm._layout
g = asm._groups[0]
contrib_d, desc_d = asm._plan_d["contrib"], asm._plan_d["desc"]
bufs = [asm._K_blk, torch.empty_like(asm._K_blk)]
data = torch.empty(p.nnz, dtype=torch.float64, device=dev)
nblk = p.nblk


def k_launch(buf, stream):
    _lib.check(lib.fs_shell_quad_batch(_lib.ptr(g["xyz_d"]), _lib.ptr(g["pid_d"]), _lib.ptr(asm._props_d),
                                       len(asm._prop_table), None, E, 1, layout, _lib.ptr(buf), None, None, None,
                                       None, stream.cuda_stream), "k")


def a_launch(buf, stream):
    _lib.check(lib.fs_csr_numeric_desc(_lib.ptr(buf), _lib.ptr(desc_d), _lib.ptr(contrib_d), nblk, 6, _lib.ptr(data),
                                       stream.cuda_stream), "a")


def run(pipelined, n):
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    t = time.perf_counter()
    a_done = [None, None]
    for s in range(n):
        b = s & 1
        if pipelined:
            if a_done[b] is not None:
                sA.wait_event(a_done[b])          # buffer b free again: its reader (step s-2) has finished
            k_launch(bufs[b], sA)
            ev = torch.cuda.Event(); ev.record(sA); sB.wait_event(ev)
            a_launch(bufs[b], sB)
            a_done[b] = torch.cuda.Event(); a_done[b].record(sB)
        else:
            k_launch(bufs[b], sA)
            a_launch(bufs[b], sA)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


for mode in (False, True, False, True):
    run(mode, 5)
    ms = run(mode, steps)
    ok = torch.equal(data, ref)
    print(f"pipelined={mode}: {ms:.3f} ms/step  {E / ms * 1e3:.4g} el/s  bitwise_equal={ok}", flush=True)
