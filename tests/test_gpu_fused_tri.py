"""GPU: the fused MITC3+ kernel (fs_fused_assemble_tri) against the two-kernel path it replaces (fs_shell_tri_batch +
fs_csr_numeric_desc): the same block function (shared tri_pair_block) and the same element-order sums -- the CSR values
agree to 1e-13 of the largest entry, the fused result is exactly symmetric and bitwise repeatable -- on structured, curved, laminate (coupled and uncoupled rows mixed), randomly renumbered and
irregular triangulations (valence 4 ... 8), plus the oracle as an independent check; fall-backs, status words and the
plan check."""
import numpy as np
import pytest
import torch

import fem_shell_amd as fsa
from fem_shell_amd import _lib, meshes
from fem_shell_amd.elements import ElementFamily
from tests import util

pytestmark = pytest.mark.gpu


def _bits(t):
    return t.contiguous().view(torch.int64)


def tri_fan(sectors, seed=0):
    """Triangles around a centre node of valence `sectors`, one ring of quads split in two outside."""
    rng = np.random.default_rng(seed)
    n = sectors
    ang = 2 * np.pi * np.arange(n) / n
    ring1 = np.stack([np.cos(ang), np.sin(ang), 0.1 * np.sin(3 * ang)], -1)
    ring2 = 2.0 * np.stack([np.cos(ang), np.sin(ang), 0.15 * np.cos(2 * ang)], -1)
    coords = np.vstack([[0.0, 0.0, 0.2], ring1, ring2]) + 0.02 * rng.standard_normal((2 * n + 1, 3))
    r1 = lambda k: 1 + (k % n)
    r2 = lambda k: 1 + n + (k % n)
    tris = [(0, r1(s), r1(s + 1)) for s in range(n)]
    for k in range(n):
        tris.append((r1(k), r2(k), r2(k + 1)))
        tris.append((r1(k), r2(k + 1), r1(k + 1)))
    return coords, np.array(tris, dtype=np.int64)


def irregular_tris():
    cs, ts, off = [], [], 0
    for sectors, shift in ((4, 0.0), (8, 6.0), (5, 12.0), (7, 18.0)):
        c, t = tri_fan(sectors, seed=sectors)
        c[:, 0] += shift
        cs.append(c); ts.append(t + off)
        off += c.shape[0]
    conn = np.vstack(ts)
    perm = np.random.default_rng(1).permutation(conn.shape[0])          # element order is not the spatial order
    return meshes.ArrayMesh(np.vstack(cs), conn[perm], set_id=(np.arange(conn.shape[0]) % 8).astype(np.int32),
                            set_names=[f"s{i}" for i in range(8)])


def _renumbered():
    m = meshes.cylindrical_patch(25.0, 25.0, 40.0, 23, 17, triangular=True)
    perm = np.random.default_rng(5).permutation(m.coords.shape[0])
    coords = np.empty_like(m.coords)
    coords[perm] = m.coords
    return meshes.ArrayMesh(coords, perm[m.conn])


def _with_sets(mesh, n=8):
    E = mesh.conn.shape[0]
    return meshes.ArrayMesh(mesh.coords, mesh.conn, set_id=((np.arange(E) // 7) % n).astype(np.int32),
                            set_names=[f"s{i}" for i in range(n)])


def _cases():
    iso = {"material": fsa.IsotropicMaterial("m", 6.825e7, 0.3, 1.0), "thickness": 0.04}
    yield "cylinder_37x23_8props", _with_sets(meshes.cylindrical_patch(25.0, 25.0, 40.0, 37, 23, triangular=True)), {"properties": "sets8"}
    yield "flat_axis_aligned_40x20", meshes.flat_patch(0.0, 2.0, 0.0, 1.0, 40, 20, triangular=True), iso
    yield "hemisphere_48", meshes.sphere_patch(10.0, 18.0, 90.0, 90.0, 48, triangular=True), iso
    yield "twisted_beam_96x8", meshes.twisted_beam(12.0, 1.1, 90.0, 96, 8, triangular=True), \
        {"material": fsa.IsotropicMaterial("t", 29.0e6, 0.22, 1.0), "thickness": 0.0032}
    yield "renumbered_23x17", _renumbered(), iso
    yield "irregular_fans", irregular_tris(), {"properties": "sets8"}


@pytest.mark.parametrize("name", [c[0] for c in _cases()])
def test_fused_tri_is_bitwise_the_two_kernel_path(oracle, name):
    from fem_shell_amd.assembler import MeshAssembler
    mesh, model = next((m, md) for n, m, md in _cases() if n == name)
    model = dict(model)
    if model.get("properties") == "sets8":
        p8 = util.standard_props()
        model["properties"] = {nm: p8[i % 8] for i, nm in enumerate(mesh.set_names)}
    model["element_family"] = ElementFamily.SHELL
    asm = MeshAssembler(mesh, {"elements": model})
    assert asm.can_fuse and asm._groups[0]["nn"] == 3
    K2 = asm.assemble_stiffness_matrix(recompute=True, fused=False)
    st = torch.full((mesh.conn.shape[0],), -1, dtype=torch.int32, device=asm.device)
    Kf = asm.assemble_stiffness_fused(status=st)
    torch.cuda.synchronize()
    s = asm.fused_plan().stats()
    print(name, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in s.items()})
    assert not torch.isnan(Kf.data).any()
    # the blocks come from the same device function (tri_pair_block) in another inlining context: where the compiler
    # forms FMAs may differ (fp contract(fast)), so the two routes agree to rounding, not to the bit
    scale = float(K2.data.abs().max())
    dev_ = float((Kf.data - K2.data).abs().max())
    assert dev_ <= 1e-13 * scale, (dev_, scale)
    assert torch.equal(st, asm._status[0])
    again = asm.assemble_stiffness_matrix(recompute=True)            # default route == fused; repeatable
    assert asm._fplan is not None and bool((_bits(again.data) == _bits(Kf.data)).all())
    A = Kf.to_scipy()
    assert (A != A.T).nnz == 0                                       # exactly symmetric
    if name in ("cylinder_37x23_8props", "irregular_fans"):          # independent check against the oracle
        Ke, _, _ = oracle.mitc3_batch(mesh.coords[mesh.conn], asm._prop_id, asm._prop_table)
        dofs = (mesh.conn[:, :, None] * 6 + np.arange(6)).reshape(mesh.conn.shape[0], -1)
        ip, ix, kd = oracle.csr_assemble(dofs, mesh.node_count * 6, Ke)
        assert np.array_equal(ip, Kf.indptr.cpu().numpy()) and np.array_equal(ix, Kf.indices.cpu().numpy())
        assert np.abs(Kf.data.cpu().numpy() - kd).max() <= 1e-12 * np.abs(kd).max()
    # the mass matrix through the same plan (fs_fused_assemble_tri_mass): the bits of mitc3_m_kernel + the descriptor
    # assembly (contraction is off in both: same statements, same element-order sums)
    M2 = asm.assemble_mass_matrix(recompute=True, fused=False)
    Mf = asm.assemble_mass_fused()
    torch.cuda.synchronize()
    assert bool((_bits(Mf.data) == _bits(M2.data)).all()), int((_bits(Mf.data) != _bits(M2.data)).sum())
    assert bool((_bits(asm.assemble_mass_matrix(recompute=True).data) == _bits(M2.data)).all())      # default route
    if name in ("cylinder_37x23_8props", "irregular_fans"):
        _, Me, _ = oracle.mitc3_batch(mesh.coords[mesh.conn], asm._prop_id, asm._prop_table, want_K=False, want_M=True)
        _, _, md = oracle.csr_assemble(dofs, mesh.node_count * 6, Me)
        assert np.abs(Mf.data.cpu().numpy() - md).max() <= 1e-13 * np.abs(md).max()
    # the Newton path keeps the two-kernel route and still works after a fused evaluation
    M = asm.assemble_mass_matrix()
    assert torch.isfinite(M.data).all() and asm._K_blk is None
    u = 1e-4 * torch.ones(asm.dofs_count, dtype=torch.float64, device=asm.device)
    KT = asm.assemble_tangent_stiffness(u)
    assert torch.isfinite(KT.data).all() and asm._K_blk is not None


def test_fused_tri_fallbacks_status_and_small_meshes():
    from fem_shell_amd.assembler import MeshAssembler
    iso = {"element_family": ElementFamily.SHELL, "material": fsa.IsotropicMaterial("m", 6.825e7, 0.3, 1.0), "thickness": 0.04}
    # a node of valence 12: its diagonal block has 12 contributions, beyond the plan's 8 -> two-kernel route, no error
    coords, conn = tri_fan(12, seed=2)
    asm = MeshAssembler(meshes.ArrayMesh(coords, conn), {"elements": iso})
    K = asm.assemble_stiffness_matrix(recompute=True)
    assert not asm.fused_ready() and asm._fplan is None and "more than 8 elements" in str(asm._fplan_error)
    assert torch.isfinite(K.data).all()
    with pytest.raises(NotImplementedError):
        asm.assemble_stiffness_fused()
    # one element, two elements
    for E in (1, 2):
        c = np.array([[0.0, 0.0, 0.0], [1.0, 0.1, 0.0], [0.9, 1.0, 0.2], [-0.1, 1.1, 0.1]])
        t = np.array([[0, 1, 2], [0, 2, 3]], dtype=np.int64)[:E]
        a = MeshAssembler(meshes.ArrayMesh(c[: 3 + (E - 1)], t), {"elements": iso})
        Kf, K2 = a.assemble_stiffness_matrix(recompute=True), a.assemble_stiffness_matrix(recompute=True, fused=False)
        assert a._fplan is not None and float((Kf.data - K2.data).abs().max()) <= 1e-13 * float(K2.data.abs().max())
    # a degenerate triangle is reported through the status words of the fused evaluation
    mesh = meshes.flat_patch(0.0, 1.0, 0.0, 1.0, 6, 5, triangular=True)
    a = MeshAssembler(mesh, {"elements": iso})
    a._groups[0]["xyz_d"][7, 2] = a._groups[0]["xyz_d"][7, 1]              # collapse element 7 in place
    a.assemble_stiffness_matrix(recompute=True)
    with pytest.raises(ValueError):
        a.check_status()


def test_tri_plan_check_rejects_malformed_plans():
    from fem_shell_amd.assembler import MeshAssembler
    mesh = meshes.cylindrical_patch(25.0, 25.0, 40.0, 20, 14, triangular=True)
    iso = {"element_family": ElementFamily.SHELL, "material": fsa.IsotropicMaterial("m", 6.825e7, 0.3, 1.0), "thickness": 0.04}
    asm = MeshAssembler(mesh, {"elements": iso})
    fp = asm.fused_plan()
    lib, dev, nnz, E = _lib.load(), asm.device, asm._plan.nnz, mesh.conn.shape[0]

    def check(**kw):
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).to(dev)
        arrs = [t(kw.get(k, getattr(fp, k))) for k in ("pat_eptr", "pat_elems", "pat_nptr", "nodes", "pat_lptr", "steps")]
        rc = lib.fs_fused_tri_plan_check(fp.n_patches, *[_lib.ptr(x) for x in arrs], E, nnz, _lib.stream_ptr())
        return rc, lib.fs_last_error().decode()

    assert check()[0] == 0
    el = fp.pat_elems.copy(); el[3] = E
    rc, msg = check(pat_elems=el); assert rc == -22 and "element index" in msg
    nd = fp.nodes.copy(); nd["gbase"][2] = nnz - 30
    rc, msg = check(nodes=nd); assert rc == -22 and "slab" in msg
    st = fp.steps.copy()
    act = np.argwhere(st & (1 << 13))
    st[act[5][0], act[5][1]] |= 255                                        # slot 255
    rc, msg = check(steps=st); assert rc == -22 and "slot" in msg
    st = fp.steps.copy()
    fl = np.argwhere(st & (1 << 14))
    st[fl[4][0], fl[4][1]] |= np.uint32(127 << 22)                         # neighbour rank 127
    rc, msg = check(steps=st); assert rc == -22 and "rank" in msg
    lp = fp.pat_lptr.copy(); lp[1] += 1
    assert check(pat_lptr=lp)[0] == -22
