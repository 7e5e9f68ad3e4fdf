"""Experiment: order of the block descriptors.  The assembly kernel issues 4 value-load instructions per wave whenever
any of its lanes has 4 contributions; grouping blocks of equal contribution count (stable sort by cnt inside windows
of W blocks) makes waves uniform.  usage: python tools/exp/exp_desc_order.py [W ...]"""
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
asm.use_row_staging = False
p = asm._plan
lib = _lib.load()
ref = asm.assemble_stiffness_matrix(recompute=True).data.clone()
desc = asm._desc_d.view(torch.int64).view(-1, 6)          # dst, (stride | cnt << 32), c0..c3
cnt = (desc[:, 1] >> 32).to(torch.int64)
nblk = desc.shape[0]
data = torch.empty_like(ref)
contrib = asm._plan_d["contrib"]
print("cnt histogram", torch.bincount(cnt).tolist())


def run(dd):
    _lib.check(lib.fs_csr_numeric_desc(_lib.ptr(asm._K_blk), _lib.ptr(dd), _lib.ptr(contrib), nblk, 6, _lib.ptr(data),
                                       _lib.stream_ptr()), "a")


def timeit(dd, n=20):
    for _ in range(3):
        run(dd)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(n):
        run(dd)
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / n


print(f"block order: {timeit(desc.contiguous()):.3f} ms")
# order by the source block of the first contribution inside windows: adjacent lanes then read adjacent source blocks
c0 = desc[:, 2] >> 1
for W in (126, 1008, 8064, 64512):
    key = (torch.arange(nblk, device=desc.device) // W) * (1 << 40) + c0
    perm = torch.argsort(key, stable=True)
    dd = desc[perm].contiguous()
    ms = timeit(dd)
    data.zero_()
    run(dd)
    torch.cuda.synchronize()
    print(f"sorted by first source block, window {W:6d}: {ms:.3f} ms  bitwise_equal={torch.equal(data, ref)}", flush=True)
idx = torch.arange(nblk, device=desc.device)
for W in [int(a) for a in sys.argv[1:]] or [28, 56, 126, 252, 1008, 4032]:
    key = (idx // W) * 8 + cnt.clamp(max=7)
    perm = torch.argsort(key, stable=True)
    dd = desc[perm].contiguous()
    ms = timeit(dd)
    data.zero_()
    run(dd)
    torch.cuda.synchronize()
    print(f"window {W:5d}: {ms:.3f} ms  bitwise_equal={torch.equal(data, ref)}", flush=True)
