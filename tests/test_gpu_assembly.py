"""GPU parity of the full path: MeshAssembler (HIP element kernels + block CSR assembly) vs the
oracle's element matrices assembled in element order (core/assembler.py:222-303 semantics)."""
import numpy as np
import pytest

import fem_shell_amd as fsa
from fem_shell_amd import meshes
from fem_shell_amd.elements import ElementFamily
from tests import util

pytestmark = pytest.mark.gpu


def _oracle_csr(oracle, mesh, table, pid, which):
    xyz = mesh.coords[mesh.conn]
    if which == "K":
        vals, _ = oracle.mitc4_K_batch(xyz, pid, table)
    else:
        vals, _ = oracle.mitc4_M_batch(xyz, pid, table)
    dofs = (mesh.conn[:, :, None] * 6 + np.arange(6)).reshape(mesh.conn.shape[0], -1)
    return oracle.csr_assemble(dofs, mesh.node_count * 6, vals)


@pytest.mark.parametrize("case", ["scordelis32", "plate_sets"])
def test_assembled_K_M(case, oracle):
    from fem_shell_amd.assembler import MeshAssembler
    if case == "scordelis32":
        mesh = meshes.cylindrical_patch(25.0, 25.0, 40.0, 32, 32)
        model = {"elements": {"element_family": ElementFamily.SHELL,
                              "material": fsa.IsotropicMaterial("sc", 4.32e8, 0.0, 360.0), "thickness": 0.25}}
    else:
        mesh = meshes.jittered_plate(40, 23, n_sets=8)
        props = util.standard_props()
        model = {"elements": {"element_family": ElementFamily.SHELL,
                              "properties": {f"band_{i}": props[i] for i in range(8)}}}
    asm = MeshAssembler(mesh, model)
    assert asm.dofs_count == mesh.node_count * 6 and asm.dofs_per_node == 6
    for which in ("K", "M"):
        A = asm.assemble_stiffness_matrix() if which == "K" else asm.assemble_mass_matrix()
        ip, ix, data = _oracle_csr(oracle, mesh, asm._prop_table, asm._prop_id, which)
        assert np.array_equal(A.indptr.cpu().numpy(), ip)        # pattern: bit-exact
        assert np.array_equal(A.indices.cpu().numpy(), ix)
        assert np.array_equal(asm._row_nnz, np.diff(ip).astype(np.int32))
        err = np.abs(A.data.cpu().numpy() - data).max() / np.abs(data).max()
        print(case, which, "nnz", A.nnz, "rel err", err)
        assert err < 1e-12
    if case == "scordelis32":
        assert A.nnz == 36 * 97 * 97                              # SURVEY.md a17 closed form
        assert np.array_equal(asm._dofs_array[5], (mesh.conn[5][:, None] * 6 + np.arange(6)).ravel())
    # determinism: two assemblies are bitwise identical (segmented reduction, no atomics)
    a = asm.assemble_stiffness_matrix(recompute=True).data.clone()
    b = asm.assemble_stiffness_matrix(recompute=True).data
    assert bool((a == b).all())


def test_element_classes_match_batch(oracle):
    from fem_shell_amd.elements import MITC4, MITC4Composite
    xyz = util.BASE_QUAD
    mat = fsa.IsotropicMaterial("steel", 2.1e11, 0.3, 7800.0)
    el = MITC4(xyz, (0, 1, 2, 3), mat, 0.01)
    table = fsa.build_prop_table([fsa.ShellProperty(mat, 0.01)])
    Ko, _ = oracle.mitc4_K_batch(xyz[None], [0], table)
    assert np.abs(el.K - Ko[0]).max() / np.abs(Ko).max() < 1e-12
    assert el.K.shape == (24, 24) and el.dofs_count == 24 and el.dofs_per_node == 6
    lam = fsa.create_laminate_from_angles(util.t300(), 0.125e-3, [0, 90])
    ec = MITC4Composite(xyz, (0, 1, 2, 3), lam)
    assert ec.has_coupling
    tl = fsa.build_prop_table([fsa.CompositeShellProperty(lam)])
    Kc, _ = oracle.mitc4_K_batch(xyz[None], [0], tl)
    assert np.abs(ec.K - Kc[0]).max() / np.abs(Kc).max() < 1e-12
    bad = np.zeros((4, 3))
    with pytest.raises(ValueError):
        MITC4(bad, (0, 1, 2, 3), mat, 0.01).K
