"""fs_csr_blk_matvec against SciPy on a small plate and its rate on the 400 x 3125 strip:
python tools/exp/matvec_bench.py"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import bench
import fem_shell_amd as fsa
from fem_shell_amd import meshes, _lib
from fem_shell_amd.assembler import MeshAssembler
from fem_shell_amd.elements import ElementFamily
mat = fsa.IsotropicMaterial("cant", 1.2e6, 0.3, 1.0)
lib = _lib.load()
for nx, ny in ((40, 30), (400, 3125)):
    mesh = meshes.flat_patch(0.0, 10.0, 0.0, 1.0 * ny / 40, nx, ny)
    asm = MeshAssembler(mesh, {"elements": {"element_family": ElementFamily.SHELL, "material": mat, "thickness": 0.1}},
                        cache_blocks=False)
    K = asm.assemble_stiffness_matrix(recompute=True)
    p, d = asm._plan, asm._plan_d
    x = torch.randn(asm.dofs_count, dtype=torch.float64, device=asm.device)
    y = torch.empty_like(x)
    def run():
        _lib.check(lib.fs_csr_blk_matvec(_lib.ptr(d["nbr_ptr"]), _lib.ptr(d["nbr"]), _lib.ptr(d["blk_row"]), p.n_nodes, 6,
                                         _lib.ptr(K.data), _lib.ptr(x), None, _lib.ptr(y), _lib.stream_ptr()), "matvec")
    run(); torch.cuda.synchronize()
    if nx == 40:
        ref = K.to_scipy() @ x.cpu().numpy()
        print("max rel err", np.abs(y.cpu().numpy() - ref).max() / np.abs(ref).max())
    ms = bench.avg_ms(torch, run, 20)
    gb = (K.data.numel() * 8 + d["nbr"].numel() * 8 + x.numel() * 16) / 1e9
    print(f"{nx}x{ny}: {ms:.4f} ms, {gb:.3f} GB -> {gb / ms * 1e3:.0f} GB/s")
