"""CPU: the native patch-plan builder of the fused kernel (fs_fused_plan_build, csrc/femshell_fused_plan.cpp -- host code
of the C-ABI library, no GPU needed) against its NumPy statement (tests/fused_plan_numpy.py): the four plan arrays must
be equal bit for bit on structured, curved, irregular (valence 3 / 5 / 7: half-lane tasks, 3-term groups, chains of
more than 4 contributions) and ghost-column (distributed) patterns; and the plan's own invariants."""
import numpy as np
import pytest

from fem_shell_amd import meshes
from fem_shell_amd.fused import FusedQuadPlan, partition_nodes
from fem_shell_amd.symbolic import BlockPlan
from tests.fused_plan_numpy import NumpyFusedQuadPlan


def _fans():
    from tests.test_gpu_fused import irregular_mesh
    return irregular_mesh()


CASES = {
    "plate_37x23": lambda: meshes.jittered_plate(37, 23, n_sets=8),
    "hemisphere_32": lambda: meshes.sphere_patch(10.0, 18.0, 90.0, 90.0, 32),
    "twisted_beam_24x4": lambda: meshes.twisted_beam(12.0, 1.1, 90.0, 24, 4),
    "tube_16x32": lambda: meshes.tapered_tube(16, 32, n_sets_around=4, n_sets_span=2),
    "irregular_fans": _fans,
}


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("max_elems", [64, 40])
def test_native_plan_equals_numpy_plan(name, max_elems):
    m = CASES[name]()
    plan = BlockPlan(m.conn, m.coords.shape[0], 6, tiled=True)
    npatch = partition_nodes(m.coords, m.conn, max_elems)
    a = FusedQuadPlan(plan, m.conn, m.coords, max_elems, node_patch=npatch)
    b = NumpyFusedQuadPlan(plan, m.conn, m.coords, max_elems, node_patch=npatch)
    assert a.n_patches == b.n_patches and a.n_evals == b.n_evals
    for k in ("pat_eptr", "pat_elems", "pat_rptr"):
        assert np.array_equal(getattr(a, k), getattr(b, k)), k
    assert a.recs.shape == b.recs.shape
    for f in a.recs.dtype.names:
        assert np.array_equal(a.recs[f], b.recs[f]), f
    assert int(np.diff(a.pat_eptr).max()) <= max_elems
    assert (a.pat_rptr % 64 == 0).all()
    # every CSR block is written exactly once: directly or as a transposed copy
    dst = np.concatenate([a.recs["dst"].ravel(), a.recs["dst_t"].ravel()])
    dst = np.sort(dst[dst >= 0])
    assert np.array_equal(dst, 36 * plan.nbr_ptr[plan.blk_row] + 6 * (np.arange(plan.nblk) - plan.nbr_ptr[plan.blk_row]))


def test_native_plan_with_ghost_columns_and_errors():
    """Pattern-only blocks (ghost columns of a distributed plan: no contributions, no transposed partner) become tasks
    of their own; broken inputs are FS_EINVAL, never a crash."""
    from fem_shell_amd._lib import FemShellHipError
    m = meshes.jittered_plate(9, 8, n_sets=1)
    N = m.coords.shape[0] + 3                                   # three ghost nodes without elements
    extra = np.array([[0, N - 3], [5, N - 2], [5, N - 1], [11, N - 1]])
    plan = BlockPlan(m.conn, N, 6, extra_pairs=extra, tiled=True)
    coords = np.vstack([m.coords, np.full((3, 3), np.nan)])
    npatch = partition_nodes(coords, m.conn, 64)
    a = FusedQuadPlan(plan, m.conn, coords, 64, node_patch=npatch)
    b = NumpyFusedQuadPlan(plan, m.conn, coords, 64, node_patch=npatch)
    for f in a.recs.dtype.names:
        assert np.array_equal(a.recs[f], b.recs[f]), f
    assert np.array_equal(a.pat_rptr, b.pat_rptr) and np.array_equal(a.pat_elems, b.pat_elems)
    with pytest.raises(FemShellHipError, match="max_patch_elems"):
        FusedQuadPlan(plan, m.conn, coords, 64, node_patch=np.zeros(N, dtype=np.int64))      # one patch for everything
    bad = m.conn.copy()
    bad[3, 2] = N + 7
    with pytest.raises(FemShellHipError, match="out of range"):
        FusedQuadPlan(plan, bad, coords, 64, node_patch=npatch)
