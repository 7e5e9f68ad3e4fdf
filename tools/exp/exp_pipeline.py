"""Experiment: overlap the (VALU-bound) element kernel with the (HBM-bound) block assembly on two
streams by element chunks.  usage: python tools/exp_pipeline.py [chunks ...]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
import bench
import fem_shell_amd as fsa
from fem_shell_amd import meshes, batch, _lib
from fem_shell_amd.assembler import MeshAssembler
from fem_shell_amd.elements import ElementFamily

lams = bench.c5_laminates()
mesh = meshes.jittered_plate(bench.STRIP_NX, bench.STRIP_NY, cell=1e-3, seed=11, n_sets=16)
model = {"elements": {"element_family": ElementFamily.SHELL,
                      "properties": {mesh.set_names[i]: lams[i] for i in range(16)}}}
asm = MeshAssembler(mesh, model)
E = mesh.element_count
p = asm._plan
lib = _lib.load()
ref = asm.assemble_stiffness_matrix(recompute=True).data.clone()
torch.cuda.synchronize()


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def serial():
    asm.refresh_element_matrices(want_K=True, want_M=False, check=False)
    return asm.assemble_stiffness_matrix(recompute=False)


print("serial ms", timeit(serial))
desc_h = p.descriptors()
cnt = np.diff(p.blk_ptr)
last = np.where(cnt > 0, p.contrib[np.maximum(p.blk_ptr[1:] - 1, 0)] >> 1, 0)
last_el = last // 10
for C in [int(a) for a in sys.argv[1:]] or [4, 8, 16]:
    ce = -(-E // C)
    ce = -(-ce // 16) * 16
    binid = last_el // ce
    order = np.argsort(binid, kind="stable")
    dsorted = desc_h[order]
    offs = np.searchsorted(binid[order], np.arange(C + 1))
    desc_d = torch.from_numpy(dsorted.view(np.uint8).reshape(-1)).to(asm.device)
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
    data = torch.empty(p.nnz, dtype=torch.float64, device=asm.device)
    g = asm._groups[0]
    contrib_d = asm._plan_d["contrib"]
    evs = [torch.cuda.Event() for _ in range(C)]
    dptr = desc_d.data_ptr()

    def piped():
        cur = torch.cuda.current_stream()
        start = torch.cuda.Event(); start.record(cur)
        sA.wait_event(start); sB.wait_event(start)
        for c in range(C):
            e0, e1 = c * ce, min(E, (c + 1) * ce)
            with torch.cuda.stream(sA):
                _lib.check(lib.fs_shell_quad_batch(g["xyz_d"].data_ptr() + e0 * 96, g["pid_d"].data_ptr() + e0 * 4,
                                                   _lib.ptr(asm._props_d), len(asm._prop_table), None, e1 - e0, 1, 1,
                                                   asm._K_blk.data_ptr() + e0 * 2880, None, None, None, None,
                                                   sA.cuda_stream), "k")
                evs[c].record(sA)
            sB.wait_event(evs[c])
            nb = int(offs[c + 1] - offs[c])
            if nb:
                _lib.check(lib.fs_csr_numeric_desc(_lib.ptr(asm._K_blk), dptr + int(offs[c]) * 48, _lib.ptr(contrib_d),
                                                   nb, 6, _lib.ptr(data), sB.cuda_stream), "a")
        done = torch.cuda.Event(); done.record(sB)
        cur.wait_event(done)

    ms = timeit(piped)
    torch.cuda.synchronize()
    ok = torch.equal(data, ref)
    print(f"chunks {C}: {ms:.3f} ms/step  {E / ms * 1e3:.4g} el/s  bitwise_equal={ok}  bins {np.diff(offs)[:4]}...")
