"""GPU: the patch plan of the fused kernel built on the device (fs_fused_plan_device_begin / _finish,
csrc/femshell_fused_plan_dev.hip) against the native host builder (fs_fused_plan_build, itself held equal to the NumPy
statement by tests/test_fused_plan_host.py): the four plan arrays bit for bit, on structured, curved, closed, irregular
(valence 3 / 5 / 7, chains of more than 4 contributions), randomly renumbered and ghost-column patterns; the errors."""
import numpy as np
import pytest
import torch

from fem_shell_amd import meshes
from fem_shell_amd._lib import FemShellHipError
from fem_shell_amd.fused import FusedQuadPlan, FusedQuadPlanDevice, partition_nodes
from fem_shell_amd.symbolic import BlockPlan, DeviceBlockPlan

pytestmark = pytest.mark.gpu


def _fans():
    from tests.test_gpu_fused import irregular_mesh
    return irregular_mesh()


def _renumbered():
    m = meshes.jittered_plate(41, 29, n_sets=4)
    perm = np.random.default_rng(5).permutation(m.coords.shape[0])
    coords = np.empty_like(m.coords)
    coords[perm] = m.coords
    return meshes.ArrayMesh(coords, perm[m.conn])


CASES = {
    "plate_37x23": lambda: meshes.jittered_plate(37, 23, n_sets=8),
    "plate_300x70": lambda: meshes.jittered_plate(300, 70, n_sets=8),
    "hemisphere_32": lambda: meshes.sphere_patch(10.0, 18.0, 90.0, 90.0, 32),
    "twisted_beam_24x4": lambda: meshes.twisted_beam(12.0, 1.1, 90.0, 24, 4),
    "tube_16x32": lambda: meshes.tapered_tube(16, 32, n_sets_around=4, n_sets_span=2),
    "irregular_fans": _fans,
    "renumbered_41x29": _renumbered,
    "tiny_2x1": lambda: meshes.jittered_plate(2, 1, n_sets=1),
}


def _both(conn, coords, n_nodes, max_elems, extra=None, node_patch=None):
    dev = torch.device("cuda", 0)
    conn = np.ascontiguousarray(conn, dtype=np.int64)
    host_plan = BlockPlan(conn, n_nodes, 6, extra_pairs=extra, tiled=True)
    if node_patch is None:
        node_patch = partition_nodes(coords, conn, max_elems)
    a = FusedQuadPlan(host_plan, conn, coords, max_elems, node_patch=node_patch)
    conn_d = torch.from_numpy(conn).to(dev)
    dplan = DeviceBlockPlan(conn, n_nodes, 6, dev, extra_pairs=extra, tiled=True)
    assert np.array_equal(dplan.nbr_ptr.cpu().numpy(), host_plan.nbr_ptr) and np.array_equal(dplan.nbr.cpu().numpy(), host_plan.nbr)
    b = FusedQuadPlanDevice(dplan, conn_d, torch.from_numpy(np.asarray(node_patch, dtype=np.int64)).to(dev), max_elems)
    return a, b


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("max_elems", [64, 40])
def test_device_plan_equals_native_plan(name, max_elems):
    m = CASES[name]()
    a, b = _both(m.conn, m.coords, m.coords.shape[0], max_elems)
    assert (a.n_patches, a.n_elem_slots, a.n_evals, a.recs.shape[0]) == (b.n_patches, b.n_elem_slots, b.n_evals, b.n_recs)
    h = b.to_host()
    for k in ("pat_eptr", "pat_elems", "pat_rptr"):
        assert np.array_equal(getattr(a, k), h[k]), k
    for f in a.recs.dtype.names:
        assert np.array_equal(a.recs[f], h["recs"][f]), f
    sa, sb = a.stats(), b.stats()
    assert sa.keys() == sb.keys() and all(abs(sa[k] - sb[k]) <= 1e-12 * abs(sa[k]) for k in sa)


def test_device_plan_with_ghost_columns_and_errors():
    m = meshes.jittered_plate(9, 8, n_sets=1)
    N = m.coords.shape[0] + 3                                   # three ghost nodes without elements
    extra = np.array([[0, N - 3], [5, N - 2], [5, N - 1], [11, N - 1]], dtype=np.int64)
    coords = np.vstack([m.coords, np.full((3, 3), np.nan)])
    a, b = _both(m.conn, coords, N, 64, extra=extra)
    h = b.to_host()
    for f in a.recs.dtype.names:
        assert np.array_equal(a.recs[f], h["recs"][f]), f
    assert np.array_equal(a.pat_rptr, h["pat_rptr"]) and np.array_equal(a.pat_elems, h["pat_elems"])
    dev = torch.device("cuda", 0)
    conn_d = torch.from_numpy(np.ascontiguousarray(m.conn, dtype=np.int64)).to(dev)
    dplan = DeviceBlockPlan(np.ascontiguousarray(m.conn, dtype=np.int64), N, 6, dev, extra_pairs=extra, tiled=True)
    with pytest.raises(FemShellHipError, match="max_patch_elems"):
        FusedQuadPlanDevice(dplan, conn_d, torch.zeros(N, dtype=torch.int64, device=dev), 64)     # one patch for everything
    with pytest.raises(FemShellHipError, match="negative patch id"):
        FusedQuadPlanDevice(dplan, conn_d, torch.full((N,), -1, dtype=torch.int64, device=dev), 64)
    bad = conn_d.clone()
    bad[3, 2] = N + 7
    with pytest.raises(FemShellHipError, match="out of range"):
        FusedQuadPlanDevice(dplan, bad, torch.zeros(N, dtype=torch.int64, device=dev), 64)


def test_device_plan_reports_a_node_pair_in_more_than_32_elements():
    from tests.test_gpu_fused import fan_mesh
    coords, conn = fan_mesh(40, seed=2)
    conn = np.ascontiguousarray(conn, dtype=np.int64)
    dev = torch.device("cuda", 0)
    conn_d = torch.from_numpy(conn).to(dev)
    dplan = DeviceBlockPlan(conn, coords.shape[0], 6, dev, tiled=True)
    # every node its own patch: no patch is too large, the pole's diagonal block still has 40 contributions
    npatch = torch.arange(coords.shape[0], dtype=torch.int64, device=dev)
    with pytest.raises(FemShellHipError, match="more than 32 elements"):
        FusedQuadPlanDevice(dplan, conn_d, npatch, 64)
