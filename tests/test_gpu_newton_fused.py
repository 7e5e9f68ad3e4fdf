"""GPU: the Newton path on the ASSEMBLED linear matrix -- fs_fused_assemble_quad_tangent (K_T = K_csr + element-order
sums of the geometric corrections), fs_csr_blk_matvec + fs_shell_quad_fint_nl (f_int = K_csr u + nonlinear membrane
forces) -- against the route it replaces (fs_shell_quad_batch(FS_WANT_KT | FS_WANT_FINT) on the element blocks +
fs_csr_numeric_desc + fs_vec_gather_sum, itself pinned on tests/golden/elem_nl.npz and the oracle loop) and against
the oracle loop 'update_configuration(u[dofs]); compute_tangent_stiffness(); compute_internal_forces()' directly
(elements/MITC4.py:1067-1084, 1246-1341)."""
import numpy as np
import pytest
import torch

import fem_shell_amd as fsa
from fem_shell_amd import _lib, meshes
from fem_shell_amd.elements import ElementFamily
from tests import util
from tests.test_gpu_fused import irregular_mesh

pytestmark = pytest.mark.gpu


def _bits(t):
    return t.contiguous().view(torch.int64)


def _cases():
    iso = {"material": fsa.IsotropicMaterial("m", 6.825e7, 0.3, 1.0), "thickness": 0.04}
    yield "plate_37x23_8props", meshes.jittered_plate(37, 23, n_sets=8), {"properties": "sets8"}
    yield "flat_plate_axis_aligned", meshes.flat_patch(0.0, 2.0, 0.0, 1.0, 40, 20), iso
    yield "hemisphere_64", meshes.sphere_patch(10.0, 18.0, 90.0, 90.0, 64), iso
    yield "tube_32x64_laminates", meshes.tapered_tube(32, 64, n_sets_around=4, n_sets_span=2), {"properties": "sets8"}
    yield "irregular_fans", irregular_mesh(), {"properties": "sets8"}


def _assembler(name, nonlinear=True, **kw):
    from fem_shell_amd.assembler import MeshAssembler
    mesh, model = next((m, md) for n, m, md in _cases() if n == name)
    model = dict(model)
    if model.get("properties") == "sets8":
        p8 = util.standard_props()
        model["properties"] = {nm: p8[i % 8] for i, nm in enumerate(mesh.set_names)}
    model["element_family"] = ElementFamily.SHELL
    model["nonlinear"] = nonlinear
    return mesh, MeshAssembler(mesh, {"elements": model}, **kw)


def _field(mesh, scale, seed=0):
    """Smooth displacement field (in-plane stretch, out-of-plane bending, rotations) + noise, size ~ scale * mesh size."""
    rng = np.random.default_rng(seed)
    c = mesh.coords - mesh.coords.mean(axis=0)
    L = np.abs(c).max()
    u = np.zeros((mesh.node_count, 6))
    u[:, 0] = 0.6 * c[:, 0] + 0.2 * c[:, 1] * c[:, 0] / L
    u[:, 1] = -0.3 * c[:, 1] + 0.1 * c[:, 2]
    u[:, 2] = 0.5 * (c[:, 0] ** 2 + c[:, 1] ** 2) / L
    u[:, 3:] = 0.3 * rng.standard_normal((mesh.node_count, 3))
    u += 0.05 * L * rng.standard_normal(u.shape) * np.array([1, 1, 1, 0, 0, 0])
    return (scale * u).reshape(-1)


@pytest.mark.parametrize("name", [c[0] for c in _cases()])
def test_tangent_and_forces_on_assembled_matrix_vs_element_block_route(name):
    mesh, asm = _assembler(name)
    assert asm._newton_fused()
    u = torch.from_numpy(_field(mesh, 2e-2, seed=3)).to(asm.device)
    asm.newton_on_assembled = False
    K2, f2 = asm.assemble_tangent_and_forces(u)
    asm.newton_on_assembled = True
    K = asm.assemble_stiffness_matrix(recompute=True)
    for flags in (0, _lib.FS_TANGENT_TWO_WGS_PER_CU):
        asm.tangent_flags = flags
        KT, f = asm.assemble_tangent_and_forces(u)
        torch.cuda.synchronize()
        scale = float(K2.data.abs().max())
        corr = float((K2.data - K.data).abs().max()) / scale
        errK = float((KT.data - K2.data).abs().max()) / scale
        # f_int = K u + ...: the rounding of both routes is set by the magnitude of the summed products
        terms = scale * float(u.abs().max())
        errF = float((f - f2).abs().max())
        print(name, "flags", flags, "correction / K", corr, "errK", errK, "errF / |K||u|", errF / terms, "errF / |f|",
              errF / float(f2.abs().max()))
        assert corr > 1e-6                       # the test field does bend the tangent
        assert errK <= 1e-13 and errF <= 1e-13 * terms
        # exactly symmetric, repeatable bit for bit
        S = KT.to_scipy()
        assert abs(S - S.T).max() == 0.0
        KT_again, f_again = asm.assemble_tangent_and_forces(u)
        assert bool((_bits(KT_again.data) == _bits(KT.data)).all()) and bool((_bits(f_again) == _bits(f)).all())
        assert bool((_bits(asm.assemble_tangent_stiffness(u).data) == _bits(KT.data)).all())
        # f_int alone: nonlinear forces from the element kernel + vector scatter instead of the tangent kernel's sums
        f_only = asm.assemble_internal_forces(u)
        assert float((f_only - f).abs().max()) <= 1e-14 * terms
        assert bool((_bits(asm.assemble_internal_forces(u)) == _bits(f_only)).all())
    asm.tangent_flags = 0
    # u = 0: the tangent is the linear matrix, the force vanishes
    KT0, f0 = asm.assemble_tangent_and_forces(torch.zeros_like(u))
    assert bool((KT0.data == K.data).all()) and float(f0.abs().max()) == 0.0
    # the packed copy of the linear blocks follows the linear matrix
    assert asm._K_lin_packed is not None and asm._K_lin_packed[0] is asm._K_lin_csr


def test_tangent_and_forces_on_assembled_matrix_vs_oracle(oracle):
    """The oracle loop directly, mixed property table (isotropic rows, coupled and uncoupled laminates)."""
    mesh, asm = _assembler("plate_37x23_8props")
    u = _field(mesh, 1e-2, seed=5)
    KT, f = asm.assemble_tangent_and_forces(u)
    assert asm._K_blk is None or asm._cache_blocks       # the Newton path itself keeps no element blocks
    dofs = asm._dofs_array
    E = mesh.conn.shape[0]
    kt_e = np.empty((E, 24, 24)); f_e = np.empty((E, 24))
    for e in range(E):
        pid = int(asm._prop_id[e])
        kt_e[e], f_e[e], _ = oracle.mitc4_NL(mesh.coords[mesh.conn[e]], asm._prop_table[pid:pid + 1], u[dofs[e]])
    ip, ix, data = oracle.csr_assemble(dofs, mesh.node_count * 6, kt_e)
    assert np.array_equal(KT.indptr.cpu().numpy(), ip) and np.array_equal(KT.indices.cpu().numpy(), ix)
    errK = np.abs(KT.data.cpu().numpy() - data).max() / np.abs(data).max()
    fo = np.zeros(mesh.node_count * 6)
    for e in range(E):
        fo[dofs[e]] += f_e[e]
    terms = np.abs(data).max() * np.abs(u).max()
    errF = np.abs(f.cpu().numpy() - fo).max()
    print("oracle: errK", errK, "errF / |K||u|", errF / terms, "errF / |f|", errF / np.abs(fo).max())
    assert errK <= 1e-12 and errF <= 1e-12 * terms


def test_linear_rows_keep_the_linear_matrix():
    """Property rows without the geometric terms (nonlinear == 0): K_T is K, f_int is K u."""
    mesh, asm = _assembler("plate_37x23_8props", nonlinear=False)
    u = torch.from_numpy(_field(mesh, 1e-2, seed=1)).to(asm.device)
    K = asm.assemble_stiffness_matrix(recompute=True)
    KT, f = asm.assemble_tangent_and_forces(u)
    assert bool((KT.data == K.data).all())
    ref = K.to_scipy() @ u.cpu().numpy()
    assert np.abs(f.cpu().numpy() - ref).max() <= 1e-13 * float(K.data.abs().max()) * float(u.abs().max())


def test_lazy_assembler_runs_newton_without_element_blocks():
    """cache_blocks=False: the Newton path works on the assembled matrix and never evaluates the element blocks; a
    recompute=True assembly refreshes the linear matrix it uses."""
    mesh, asm = _assembler("hemisphere_64", cache_blocks=False)
    u = torch.from_numpy(_field(mesh, 1e-2)).to(asm.device)
    KT, f = asm.assemble_tangent_and_forces(u)
    assert asm._K_blk is None and asm._K_lin_csr is not None
    g = asm._groups[0]
    g["xyz_d"].mul_(1.01)                                  # inputs changed in place ...
    asm.assemble_stiffness_matrix(recompute=True)          # ... and re-evaluated
    assert asm._K_lin_csr is None
    KT2, f2 = asm.assemble_tangent_and_forces(u)
    assert float((KT2.data - KT.data).abs().max()) > 0.0
    asm.newton_on_assembled = False
    KT3, f3 = asm.assemble_tangent_and_forces(u)
    assert float((KT3.data - KT2.data).abs().max()) <= 1e-13 * float(KT3.data.abs().max())


@pytest.mark.parametrize("d", [6, 3])
def test_blk_matvec_vs_scipy(d):
    from fem_shell_amd.assembler import MeshAssembler
    if d == 6:
        mesh, asm = _assembler("irregular_fans")
    else:
        from tests.test_gpu_fused_hex import brick_block
        asm = MeshAssembler(meshes.ArrayMesh(*brick_block(7, 5, 4, shear=0.2)),
                            {"elements": {"element_family": ElementFamily.SOLID, "material": util.t300()}})
    K = asm.assemble_stiffness_matrix()
    lib, pd = _lib.load(), asm._plan_d
    rng = np.random.default_rng(0)
    x = torch.from_numpy(rng.standard_normal(asm.dofs_count)).to(asm.device)
    y0 = torch.from_numpy(rng.standard_normal(asm.dofs_count)).to(asm.device)
    S = K.to_scipy()
    for add in (None, y0):
        y = torch.full_like(x, float("nan"))
        _lib.check(lib.fs_csr_blk_matvec(_lib.ptr(pd["nbr_ptr"]), _lib.ptr(pd["nbr"]), _lib.ptr(pd["blk_row"]),
                                         int(asm._plan.n_nodes), d, _lib.ptr(K.data), _lib.ptr(x), _lib.ptr(add),
                                         _lib.ptr(y), _lib.stream_ptr()), "fs_csr_blk_matvec")
        ref = S @ x.cpu().numpy() + (0.0 if add is None else add.cpu().numpy())
        assert np.abs(y.cpu().numpy() - ref).max() <= 1e-14 * np.abs(S).max() * asm.dofs_count ** 0.5
    with pytest.raises(_lib.FemShellHipError, match="alias"):
        _lib.check(lib.fs_csr_blk_matvec(_lib.ptr(pd["nbr_ptr"]), _lib.ptr(pd["nbr"]), _lib.ptr(pd["blk_row"]),
                                         int(asm._plan.n_nodes), d, _lib.ptr(K.data), _lib.ptr(x), None, _lib.ptr(x),
                                         _lib.stream_ptr()), "fs_csr_blk_matvec")
