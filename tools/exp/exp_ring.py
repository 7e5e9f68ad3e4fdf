"""Experiment: chunked element kernel + block assembly with the element matrices living in a small RING
buffer that stays resident in the 256 MB Infinity Cache, captured in a HIP graph (two branches: element
kernel chunks / assembly chunks).  usage: python tools/exp/exp_ring.py [chunk_elements ...]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
import bench
from fem_shell_amd import meshes, _lib
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
dev = asm.device
ref = asm.assemble_stiffness_matrix(recompute=True).data.clone()
torch.cuda.synchronize()
layout = asm._layout
assert layout == _lib.FS_LAYOUT_SYMBLK_T16
g = asm._groups[0]
contrib_d = asm._plan_d["contrib"]


def timeit(fn, n=20):
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


print("serial (python launches) ms", timeit(serial), flush=True)

desc_h = asm._plan_d["desc"].cpu().numpy().view(np.dtype([("dst", "<i8"), ("stride", "<i4"), ("cnt", "<i4"), ("c", "<i8", (4,))]))
cnt = desc_h["cnt"]
assert cnt.max() <= 4
cc = desc_h["c"]
src = cc >> 1                                  # block index in the tiled buffer
valid = np.arange(4)[None, :] < cnt[:, None]
el_of = (src // 160) * 16                      # first element of the tile (tile granularity is enough)
last_el = np.where(valid, el_of + 15, -1).max(axis=1)
first_el = np.where(valid, el_of, E + 16).min(axis=1)


def k_launch(e0, e1, out_ptr, stream):
    _lib.check(lib.fs_shell_quad_batch(g["xyz_d"].data_ptr() + e0 * 96, g["pid_d"].data_ptr() + e0 * 4,
                                       _lib.ptr(asm._props_d), len(asm._prop_table), None, e1 - e0, 1, layout,
                                       out_ptr, None, None, None, None, stream.cuda_stream), "k")


def a_launch(vals_ptr, dptr, nb, data, stream):
    _lib.check(lib.fs_csr_numeric_desc(vals_ptr, dptr, _lib.ptr(contrib_d), nb, 6, _lib.ptr(data), stream.cuda_stream), "a")


for ce in [int(a) for a in sys.argv[1:]] or [16384, 32768, 65536]:
    assert ce % 16 == 0
    C = -(-E // ce)
    # rows (descriptors are in block order) become ready after the chunk holding their last element;
    # prefix-max keeps the ranges contiguous
    ready = np.maximum.accumulate(last_el // ce)
    offs = np.searchsorted(ready, np.arange(C + 1))
    span = (ready - first_el // ce).max()      # how many chunks back a range reaches
    for NS in (0, 4):                          # 0: full buffer, 4: ring of 4 chunk slots
        if NS and span > NS - 2:
            print(f"chunk {ce}: ring of {NS} too small (span {span})"); continue
        if NS:
            bpc = ce // 16 * 160               # blocks per chunk slot
            ring = torch.empty((NS * bpc, 36), dtype=torch.float64, device=dev)
            slot_src = ((src // bpc) % NS) * bpc + src % bpc
            d2 = desc_h.copy()
            d2["c"] = np.where(valid, (slot_src << 1) | (cc & 1), 0)
            desc_d = torch.from_numpy(d2.view(np.uint8).reshape(-1)).to(dev)
            vals_ptr = ring.data_ptr()
        else:
            desc_d = asm._plan_d["desc"]
            vals_ptr = asm._K_blk.data_ptr()
        dptr = desc_d.data_ptr()
        data = torch.empty(p.nnz, dtype=torch.float64, device=dev)
        for two_streams in ((False,) if NS else (False, True)):
            sA, sB = torch.cuda.Stream(), torch.cuda.Stream()

            def body():
                cur = torch.cuda.current_stream()
                if two_streams:
                    sA.wait_stream(cur); sB.wait_stream(cur)
                a_ev = [None] * C
                for c in range(C):
                    e0, e1 = c * ce, min(E, (c + 1) * ce)
                    ka = sA if two_streams else cur
                    kb = sB if two_streams else cur
                    if two_streams and NS and c >= NS - 1 and a_ev[c - (NS - 1)] is not None:
                        ka.wait_event(a_ev[c - (NS - 1)])      # slot reuse: its last reader has finished
                    out_ptr = (vals_ptr + (c % NS) * (ce // 16 * 160) * 288) if NS else (vals_ptr + e0 * 2880)
                    k_launch(e0, e1, out_ptr, ka)
                    if two_streams:
                        ev = torch.cuda.Event(); ev.record(ka); kb.wait_event(ev)
                    nb = int(offs[c + 1] - offs[c])
                    if nb:
                        a_launch(vals_ptr, dptr + int(offs[c]) * 48, nb, data, kb)
                    if two_streams:
                        a_ev[c] = torch.cuda.Event(); a_ev[c].record(kb)
                if two_streams:
                    cur.wait_stream(sA); cur.wait_stream(sB)

            try:
                gr = torch.cuda.CUDAGraph()
                cap = torch.cuda.Stream()
                with torch.cuda.graph(gr, stream=cap):
                    body()
                ms = timeit(gr.replay)
                torch.cuda.synchronize()
                ok = torch.equal(data, ref)
                print(f"chunk {ce:6d} x{C:3d} ring={NS} streams={2 if two_streams else 1}: {ms:.3f} ms/step "
                      f"{E / ms * 1e3:.4g} el/s bitwise_equal={ok}", flush=True)
            except Exception as ex:   # noqa: BLE001
                print(f"chunk {ce} ring={NS} two={two_streams}: FAILED {type(ex).__name__}: {str(ex)[:200]}", flush=True)
            data.zero_()
