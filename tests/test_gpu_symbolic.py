"""GPU: device symbolic phase (fs_csr_symbolic_begin / _finish) vs the host BlockPlan, which
tests/test_host.py pins against the reference's set-union pattern (core/assembler.py:222-241).
Integer work: everything must be bit-exact."""
import numpy as np
import pytest

from fem_shell_amd import meshes, symbolic

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _same(dp, hp):
    assert dp.nblk == hp.nblk and dp.nnz == hp.nnz and dp.n_src_blocks == hp.n_src_blocks
    for k in ("nbr_ptr", "nbr", "blk_row", "blk_ptr", "contrib"):
        a, b = getattr(dp, k).cpu().numpy(), getattr(hp, k)
        assert a.dtype == b.dtype and np.array_equal(a, b), k
    assert np.array_equal(dp.descriptors(), hp.descriptors())
    assert np.array_equal(dp.row_nnz(), hp.row_nnz())


@pytest.mark.parametrize("case", ["quads", "quads_t16", "tris", "mixed", "hexa", "extra", "shuffled"])
def test_device_plan_equals_host_plan(case):
    d = 6
    extra = None
    tiled = case in ("quads_t16", "extra")
    if case in ("quads", "quads_t16"):
        mesh = meshes.jittered_plate(37, 23, n_sets=4)
        conn, n = mesh.conn, mesh.node_count
    elif case == "tris":
        mesh = meshes.cylindrical_patch(25.0, 25.0, 40.0, 12, 10, triangular=True)
        conn, n = mesh.conn, mesh.node_count
    elif case == "mixed":
        conn, n = meshes.mixed_conn(9, 7), 10 * 8
    elif case == "hexa":
        from tests.test_gpu_assembly import _hex_block
        coords, conn = _hex_block(6, 5, 4)
        n, d = coords.shape[0], 3
    elif case == "extra":
        mesh = meshes.jittered_plate(11, 9, n_sets=2)
        conn, n = mesh.conn, mesh.node_count
        rng = np.random.default_rng(0)
        extra = np.stack([rng.integers(0, n, 200), rng.integers(0, n, 200)], 1)
        extra = np.concatenate([extra, extra[:, ::-1], extra[:5]])      # symmetric, with duplicates
    else:   # element and node numbering without locality
        mesh = meshes.jittered_plate(29, 31, n_sets=2)
        rng = np.random.default_rng(1)
        perm = rng.permutation(mesh.node_count)
        conn, n = perm[mesh.conn][rng.permutation(mesh.conn.shape[0])], mesh.node_count
    hp = symbolic.BlockPlan(conn, n, d, extra_pairs=extra, tiled=tiled)
    dp = symbolic.DeviceBlockPlan(conn, n, d, DEV, extra_pairs=extra, tiled=tiled)
    if tiled:   # 37*23 = 851 elements: the last tile is partial
        assert hp.n_src_blocks == -(-conn.shape[0] // 16) * 160 and (hp.contrib >> 1).max() < hp.n_src_blocks
    _same(dp, hp)
    assert np.array_equal(dp.host_pattern()[1], hp.host_pattern()[1])


def test_device_plan_errors_and_empty():
    from fem_shell_amd._lib import FemShellHipError
    conn = np.array([[0, 1, 2, 7]])
    with pytest.raises(FemShellHipError):
        symbolic.DeviceBlockPlan(conn, 4, 6, DEV)           # node 7 outside [0, 4)
    dp = symbolic.DeviceBlockPlan(np.zeros((0, 4), dtype=np.int64), 5, 6, DEV)
    assert dp.nblk == 0 and dp.nnz == 0 and not dp.nbr_ptr.cpu().numpy().any()


@pytest.mark.parametrize("kind", ["tris", "hexa"])
def test_packed_and_listed_descriptors_give_the_same_sums(kind):
    """Blocks with 5..8 contributions: codes inline (FS_BLK_PACKED, what fs_csr_symbolic_finish writes) against the
    same blocks described through the `contrib` list (flag clear, c[0] = offset) -- the assembly must be bit-identical,
    d = 6 (triangles: 6 elements around a node) and d = 3 (bricks: 8)."""
    import torch
    from fem_shell_amd import _lib
    rng = np.random.default_rng(4)
    if kind == "tris":
        mesh = meshes.cylindrical_patch(25.0, 25.0, 40.0, 14, 11, triangular=True)
        conn, n, d = mesh.conn, mesh.node_count, 6
    else:
        from tests.test_gpu_assembly import _hex_block
        coords, conn = _hex_block(7, 5, 4)
        n, d = coords.shape[0], 3
    dp = symbolic.DeviceBlockPlan(conn, n, d, DEV)
    desc = dp.descriptors().copy()
    packed = (desc["cnt"] & symbolic.FS_BLK_PACKED) != 0
    assert packed.any() and int((desc["cnt"][packed] & (symbolic.FS_BLK_PACKED - 1)).max()) == (6 if kind == "tris" else 8)
    listed = desc.copy()
    listed["cnt"][packed] &= symbolic.FS_BLK_PACKED - 1
    listed["c"][packed] = 0
    listed["c"][packed, 0] = dp.blk_ptr.cpu().numpy()[:-1][packed]
    vals = torch.from_numpy(rng.standard_normal((dp.n_src_blocks, d * d))).to(DEV)
    lib, out = _lib.load(), []
    for dsc in (desc, listed):
        dd = torch.from_numpy(dsc.view(np.uint8).reshape(-1).copy()).to(DEV)
        data = torch.full((dp.nnz,), np.nan, dtype=torch.float64, device=DEV)
        _lib.check(lib.fs_csr_numeric_desc(_lib.ptr(vals), _lib.ptr(dd), _lib.ptr(dp.contrib), dp.nblk, d, _lib.ptr(data),
                                           _lib.stream_ptr()), "fs_csr_numeric_desc")
        out.append(data)
    torch.cuda.synchronize()
    assert not torch.isnan(out[0]).any()
    assert bool((out[0].view(torch.int64) == out[1].view(torch.int64)).all())
