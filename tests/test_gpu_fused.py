"""GPU: the fused evaluate-and-assemble kernel (fs_fused_assemble_quad) against the two-kernel path it replaces
(fs_shell_quad_batch + fs_csr_numeric_desc): same element-order sums of the same block function; the stiffness values
agree to rounding (util.assert_fused_close: a transposed contribution is formed as the block of the swapped node pair
since round 3), the fused matrix is EXACTLY symmetric and bitwise repeatable, the mass matrix is the same bits -- on
structured, curved, laminate (coupled and uncoupled property rows mixed), flat axis-aligned (exact zeros) and irregular
meshes (valence 3 and valence > 4 nodes: 3-term groups and the serial path), plus the oracle as an independent check."""
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


def fan_mesh(sectors, seed=0):
    """Quads around a centre node of valence `sectors` (a second ring outside): irregular valences 3 / 5 / 7 ..."""
    rng = np.random.default_rng(seed)
    n = sectors
    ang = 2 * np.pi * np.arange(2 * n) / (2 * n)
    ring1 = np.stack([np.cos(ang), np.sin(ang), 0.1 * np.sin(3 * ang)], -1)
    ring2 = 2.0 * np.stack([np.cos(ang), np.sin(ang), 0.15 * np.cos(2 * ang)], -1)
    coords = np.vstack([[0.0, 0.0, 0.2], ring1, ring2]) + 0.02 * rng.standard_normal((4 * n + 1, 3))
    r1 = lambda k: 1 + (k % (2 * n))
    r2 = lambda k: 1 + 2 * n + (k % (2 * n))
    quads = []
    for s in range(n):                 # centre, spoke, mid, next spoke
        quads.append((0, r1(2 * s), r1(2 * s + 1), r1(2 * s + 2)))
    for k in range(2 * n):             # outer ring
        quads.append((r1(k), r2(k), r2(k + 1), r1(k + 1)))
    return coords, np.array(quads, dtype=np.int64)


def irregular_mesh():
    parts, off = [], 0
    cs, qs = [], []
    for sectors, shift in ((3, 0.0), (7, 6.0), (5, 12.0)):
        c, q = fan_mesh(sectors, seed=sectors)
        c[:, 0] += shift
        cs.append(c); qs.append(q + off)
        off += c.shape[0]
    conn = np.vstack(qs)
    rng = np.random.default_rng(1)
    perm = rng.permutation(conn.shape[0])          # element order is not the spatial order
    return meshes.ArrayMesh(np.vstack(cs), conn[perm], set_id=(np.arange(conn.shape[0]) % 8).astype(np.int32),
                            set_names=[f"s{i}" for i in range(8)])


def _cases():
    props8 = util.standard_props()
    iso = {"material": fsa.IsotropicMaterial("m", 6.825e7, 0.3, 1.0), "thickness": 0.04}
    yield "plate_37x23_8props", meshes.jittered_plate(37, 23, n_sets=8), {"properties": "sets8"}
    yield "plate_64x50_iso", meshes.jittered_plate(64, 50, n_sets=1), iso
    yield "scordelis_32_flat_axes", meshes.cylindrical_patch(25.0, 25.0, 40.0, 32, 32), \
        {"material": fsa.IsotropicMaterial("s", 4.32e8, 0.0, 360.0), "thickness": 0.25}
    yield "flat_plate_axis_aligned", meshes.flat_patch(0.0, 2.0, 0.0, 1.0, 40, 20), iso
    yield "hemisphere_64", meshes.sphere_patch(10.0, 18.0, 90.0, 90.0, 64), iso
    yield "twisted_beam_96x16", meshes.twisted_beam(12.0, 1.1, 90.0, 96, 16), \
        {"material": fsa.IsotropicMaterial("t", 29.0e6, 0.22, 1.0), "thickness": 0.0032}
    yield "tube_32x64_laminates", meshes.tapered_tube(32, 64, n_sets_around=4, n_sets_span=2), {"properties": "sets8"}
    yield "irregular_fans", irregular_mesh(), {"properties": "sets8"}


@pytest.mark.parametrize("name", [c[0] for c in _cases()])
@pytest.mark.parametrize("max_elems", [64, 40])
def test_fused_against_the_two_kernel_path(oracle, name, max_elems):
    from fem_shell_amd.assembler import MeshAssembler
    mesh, model = next((m, md) for n, m, md in _cases() if n == name)
    model = dict(model)
    if model.get("properties") == "sets8":
        p8 = util.standard_props()
        model["properties"] = {nm: p8[i % 8] for i, nm in enumerate(mesh.set_names)}
    model["element_family"] = ElementFamily.SHELL
    asm = MeshAssembler(mesh, {"elements": model})
    asm.fused_max_patch_elems = max_elems
    assert asm.can_fuse
    K2 = asm.assemble_stiffness_matrix(recompute=True, fused=False)
    st = torch.full((mesh.conn.shape[0],), -1, dtype=torch.int32, device=asm.device)
    Kf = asm.assemble_stiffness_fused(status=st)
    torch.cuda.synchronize()
    fp = asm.fused_plan()
    s = fp.stats()
    print(name, "patches", s["patches"], "recompute %.3f" % s["stage1_recompute"], "lane fill %.3f" % s["stage2_lane_fill"],
          "long chains", int((fp.recs["flags"] & 8 != 0).sum()))
    assert int(np.diff(fp.pat_eptr).max()) <= max_elems
    assert not torch.isnan(Kf.data).any()
    d = util.assert_fused_close(Kf, K2, name)
    util.assert_exactly_symmetric(Kf)
    print(name, "fused vs two-kernel route", d)
    assert torch.equal(st, asm._status[0])
    again = asm.assemble_stiffness_matrix(recompute=True)            # default route == fused; repeatable
    assert bool((_bits(again.data) == _bits(Kf.data)).all())
    # the mass matrix through the same kernel skeleton (FS_FUSED_MASS) against element kernel + assembly kernel
    M2 = asm.assemble_mass_matrix(recompute=True, fused=False)
    Mf = asm.assemble_mass_fused()
    torch.cuda.synchronize()
    assert bool((_bits(Mf.data) == _bits(M2.data)).all()), int((_bits(Mf.data) != _bits(M2.data)).sum())
    assert bool((_bits(asm.assemble_mass_matrix(recompute=True).data) == _bits(M2.data)).all())     # default route
    # independent check of sampled rows against the oracle
    if name in ("plate_37x23_8props", "irregular_fans"):
        Ke, _ = oracle.mitc4_K_batch(mesh.coords[mesh.conn], asm._prop_id, asm._prop_table)
        dofs = (mesh.conn[:, :, None] * 6 + np.arange(6)).reshape(mesh.conn.shape[0], -1)
        ip, ix, kd = oracle.csr_assemble(dofs, mesh.node_count * 6, Ke)
        assert np.array_equal(ip, Kf.indptr.cpu().numpy()) and np.array_equal(ix, Kf.indices.cpu().numpy())
        assert np.abs(Kf.data.cpu().numpy() - kd).max() <= 1e-12 * np.abs(kd).max()


def test_fused_reports_bad_property_ids():
    """prop_id outside the table: row 0 is used and FS_ST_BAD_PROP is set -- never an out-of-bounds read."""
    from fem_shell_amd.assembler import MeshAssembler
    mesh = meshes.jittered_plate(20, 10, n_sets=1)
    iso = {"element_family": ElementFamily.SHELL, "material": fsa.IsotropicMaterial("m", 6.825e7, 0.3, 1.0), "thickness": 0.04}
    asm = MeshAssembler(mesh, {"elements": iso})
    ref = asm.assemble_stiffness_fused()
    pid = asm._groups[0]["pid_d"]
    pid[7], pid[100] = 12345, -3
    st = torch.zeros(mesh.conn.shape[0], dtype=torch.int32, device=asm.device)
    K = asm.assemble_stiffness_fused(status=st)
    torch.cuda.synchronize()
    assert int(st[7]) & 8 and int(st[100]) & 8 and int((st != 0).sum()) == 2
    assert bool((_bits(K.data) == _bits(ref.data)).all())          # one property row: clamping changes nothing


def test_fused_small_and_ragged_meshes():
    """Tiny and odd-sized plates (1 x 1 ... 23 x 9: fewer patches than XCDs, partly filled tiles and waves, single
    rows): K (to rounding, exactly symmetric) and M (bit for bit) of the fused route against the two-kernel route."""
    from fem_shell_amd.assembler import MeshAssembler
    p8 = util.standard_props()
    rng = np.random.default_rng(3)
    sizes = [(1, 1), (2, 1), (1, 7), (3, 3), (16, 1), (5, 4), (17, 2), (23, 9)] + \
            [(int(a), int(b)) for a, b in rng.integers(1, 21, size=(12, 2))]
    for nx, ny in sizes:
        mesh = meshes.jittered_plate(nx, ny, n_sets=8)
        model = {"element_family": ElementFamily.SHELL,
                 "properties": {nm: p8[i % 8] for i, nm in enumerate(mesh.set_names)}}
        asm = MeshAssembler(mesh, {"elements": model})
        assert asm.can_fuse
        K2 = asm.assemble_stiffness_matrix(recompute=True, fused=False)
        M2 = asm.assemble_mass_matrix(recompute=True, fused=False)
        Kf, Mf = asm.assemble_stiffness_fused(), asm.assemble_mass_fused()
        torch.cuda.synchronize()
        util.assert_fused_close(Kf, K2, (nx, ny, "K"))
        util.assert_exactly_symmetric(Kf)
        assert bool((_bits(Mf.data) == _bits(M2.data)).all()), (nx, ny, "M")


def test_mesh_beyond_the_plan_limits_falls_back_to_two_kernels():
    """A pole node of valence 40: its diagonal block has 40 contributions (fs_fused_plan_build allows 32) -- the default
    entry points must not raise, they remember the mesh as not fusable and take element kernel + assembly kernel."""
    from fem_shell_amd.assembler import MeshAssembler
    coords, conn = fan_mesh(40, seed=2)
    mesh = meshes.ArrayMesh(coords, conn)
    iso = {"element_family": ElementFamily.SHELL, "material": fsa.IsotropicMaterial("m", 6.825e7, 0.3, 1.0), "thickness": 0.04}
    ref = MeshAssembler(mesh, {"elements": iso})
    K2 = ref.assemble_stiffness_matrix(recompute=True, fused=False)
    M2 = ref.assemble_mass_matrix(recompute=True, fused=False)
    for cache in (True, False):
        asm = MeshAssembler(mesh, {"elements": iso}, cache_blocks=cache)
        M = asm.assemble_mass_matrix()                     # default route: tries the fused plan first
        K = asm.assemble_stiffness_matrix(recompute=True)
        assert not asm.can_fuse and asm._fplan_error is not None and "32" in str(asm._fplan_error)
        assert bool((_bits(K.data) == _bits(K2.data)).all()) and bool((_bits(M.data) == _bits(M2.data)).all())
        with pytest.raises(NotImplementedError):
            asm.assemble_stiffness_fused()


def test_cache_blocks_false_keeps_no_element_matrices(oracle):
    """MeshAssembler(cache_blocks=False): the constructor checks the geometry through fs_shell_quad_geom_status, K and M
    come from the fused kernel, the element blocks appear only when `_ke_array` or the Newton path ask for them."""
    from fem_shell_amd.assembler import MeshAssembler
    p8 = util.standard_props()
    mesh = meshes.jittered_plate(23, 17, n_sets=8)
    model = {"element_family": ElementFamily.SHELL, "nonlinear": True,
             "properties": {nm: p8[i % 8] for i, nm in enumerate(mesh.set_names)}}
    full = MeshAssembler(mesh, {"elements": model})
    lazy = MeshAssembler(mesh, {"elements": model}, cache_blocks=False)
    assert lazy._K_blk is None and torch.equal(lazy._status[0], full._status[0])
    K, Kl = full.assemble_stiffness_matrix(recompute=True), lazy.assemble_stiffness_matrix()
    assert lazy._K_blk is None and bool((_bits(K.data) == _bits(Kl.data)).all())
    assert bool((_bits(full.assemble_mass_matrix().data) == _bits(lazy.assemble_mass_matrix().data)).all())
    lazy.check_status()
    u = 1e-3 * torch.from_numpy(np.random.default_rng(0).standard_normal(lazy.dofs_count)).to(lazy.device)
    (KT, f), (KTl, fl) = full.assemble_tangent_and_forces(u), lazy.assemble_tangent_and_forces(u)
    assert lazy._K_blk is None                             # the Newton path works on the assembled matrix
    assert bool((_bits(KT.data) == _bits(KTl.data)).all()) and torch.equal(f, fl)
    full.newton_on_assembled = lazy.newton_on_assembled = False     # the element-block route
    (KT, f), (KTl, fl) = full.assemble_tangent_and_forces(u), lazy.assemble_tangent_and_forces(u)
    assert lazy._K_blk is not None                         # evaluated on first use
    assert bool((_bits(KT.data) == _bits(KTl.data)).all()) and torch.equal(f, fl)
    assert np.array_equal(full._ke_array[:5], lazy._ke_array[:5])
    # a degenerate element is still caught by the constructor
    bad = meshes.jittered_plate(6, 5, n_sets=8)
    bad.coords[bad.conn[7]] = bad.coords[bad.conn[7, 0]]
    with pytest.raises(ValueError, match="degenerate"):
        MeshAssembler(bad, {"elements": model}, cache_blocks=False)
    # the status words of a fused re-evaluation replace the constructor's (inputs mutated in place)
    pid = lazy._groups[0]["pid_d"]
    pid[3] = 999
    lazy.assemble_stiffness_matrix(recompute=True)
    with pytest.raises(ValueError, match="element 3"):
        lazy.check_status()


def test_plan_check_rejects_malformed_plans():
    """fs_fused_plan_check: a plan of fs_fused_plan_build passes; corrupted ones (empty patch, slot beyond the patch,
    destination outside the value array, rank byte out of range, odd destination) return FS_EINVAL -- no launch."""
    from fem_shell_amd.assembler import MeshAssembler
    from fem_shell_amd.fused import REC_DTYPE
    mesh = meshes.jittered_plate(30, 20, n_sets=1)
    iso = {"element_family": ElementFamily.SHELL, "material": fsa.IsotropicMaterial("m", 6.825e7, 0.3, 1.0), "thickness": 0.04}
    asm = MeshAssembler(mesh, {"elements": iso})
    fp = asm.fused_plan()
    lib, dev, nnz = _lib.load(), asm.device, asm._plan.nnz

    def check(eptr=None, elems=None, rptr=None, recs=None):
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        a = [t(fp.pat_eptr if eptr is None else eptr), t(fp.pat_elems if elems is None else elems),
             t(fp.pat_rptr if rptr is None else rptr), t((fp.recs if recs is None else recs).view(np.uint8).reshape(-1))]
        rc = lib.fs_fused_plan_check(fp.n_patches, *[_lib.ptr(x) for x in a], mesh.conn.shape[0], fp.max_patch_elems, nnz,
                                     _lib.stream_ptr())
        return rc, lib.fs_last_error().decode()

    assert check()[0] == 0
    e = fp.pat_eptr.copy(); e[2] = e[1]                                    # patch 1 holds no element
    assert check(eptr=e)[0] == -22
    el = fp.pat_elems.copy(); el[5] = mesh.conn.shape[0]                   # element index == E
    rc, msg = check(elems=el); assert rc == -22 and "element index" in msg
    r = fp.pat_rptr.copy(); r[3] += 1                                      # not whole waves
    assert check(rptr=r)[0] == -22
    act = np.flatnonzero(fp.recs["ev"][:, 0] & (1 << 13))
    rc_ = fp.recs.copy(); rc_["ev"][act[10], 0] |= 255                     # slot 255 of a patch of <= 64
    rc, msg = check(recs=rc_); assert rc == -22 and "slot" in msg
    out = np.flatnonzero(fp.recs["flags"] & 2)
    rc_ = fp.recs.copy(); rc_["dst"][out[7], 0] = nnz - 8                  # block would run past the end
    rc, msg = check(recs=rc_); assert rc == -22 and "destination" in msg
    rc_ = fp.recs.copy(); rc_["dst"][out[7], 0] += 1                       # misaligned 16-byte stores
    assert check(recs=rc_)[0] == -22
    rc_ = fp.recs.copy(); rc_["slots"][out[9]] = (rc_["slots"][out[9]] & ~np.uint32(0xFF)) | np.uint32(200)
    rc, msg = check(recs=rc_); assert rc == -22 and "slot" in msg
    rc_ = fp.recs.copy(); rc_["flags"][out[3]] |= np.uint32(15 << 16)      # chain position beyond the chain
    assert check(recs=rc_)[0] == -22


def test_device_partition_equals_the_numpy_partition_on_lattice_meshes():
    """fused.partition_nodes_device (torch, on the device) against fused.partition_nodes (NumPy): the same node -> patch
    map on lattice-numbered meshes -- flat, curved, closed tube --, None on meshes that are not numbered like a grid."""
    from fem_shell_amd import fused
    dev = torch.device("cuda", 0)
    cases = [meshes.jittered_plate(37, 23, n_sets=4), meshes.cylindrical_patch(25.0, 25.0, 40.0, 19, 31),
             meshes.jittered_plate(200, 9, seed=4), meshes.tapered_tube(16, 32, n_sets_around=4, n_sets_span=2)]
    for k, mesh in enumerate(cases):
        for max_elems in (64, 48):
            want = fused.partition_nodes(mesh.coords, mesh.conn, max_elems)
            got = fused.partition_nodes_device(mesh.node_count, torch.from_numpy(np.asarray(mesh.conn)).to(dev), max_elems)
            if got is None and k == 3:
                continue          # the wrap-around column of the closed tube may need the bisection pass of the NumPy version
            assert got is not None and got.dtype == np.int64
            assert np.array_equal(got, want)
    coords, conn = fan_mesh(12, seed=1)          # not numbered like a grid: the NumPy version takes over
    assert fused.partition_nodes_device(coords.shape[0], torch.from_numpy(conn).to(dev), 64) is None
