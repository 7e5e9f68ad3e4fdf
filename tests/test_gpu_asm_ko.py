"""GPU parity against the reference's UNMODIFIED MeshAssembler on the meshes SURVEY.md 8(c)-6 names
(tests/golden/asm_scordelis_32.npz = config C1 itself, asm_hemisphere_16.npz, asm_twisted_16x96.npz; made by
tools/gen_golden_asm_ko.py): dofs_count, _dofs_array, _row_nnz and the CSR pattern bit-exact, values 1e-12, and the
linear-static displacement of the benchmark case (device Dirichlet reduction -> SciPy solve) against the reference's
own solve within max(1e-12, 10 x floor), floor = the reference's own sensitivity of u to a +-1 ulp change of its
input coordinates (stored in the fixture, printed beside the achieved error)."""
import numpy as np
import pytest

import fem_shell_amd as fsa
from fem_shell_amd import meshes
from fem_shell_amd.bc import BoundaryConditionManager, DirichletCondition
from fem_shell_amd.elements import ElementFamily
from tests.util_golden import golden

pytestmark = pytest.mark.gpu

EXPECTED = {"asm_scordelis_32": 0.9973, "asm_hemisphere_16": 1.009, "asm_twisted_16x96": 1.02}


@pytest.mark.parametrize("name", sorted(EXPECTED))
def test_reference_assembler_ko_golden(name):
    from scipy.sparse.linalg import spsolve
    from fem_shell_amd.assembler import MeshAssembler
    g = golden(name)
    mesh = meshes.ArrayMesh(g["coords"], g["conn"])
    mat = fsa.IsotropicMaterial("m", float(g["E"]), float(g["nu"]), float(g["rho"]))
    asm = MeshAssembler(mesh, {"elements": {"element_family": ElementFamily.SHELL, "material": mat,
                                            "thickness": float(g["t"])}})
    assert asm.dofs_count == int(g["dofs_count"])
    assert np.array_equal(asm._dofs_array, g["dofs_array"])
    assert np.array_equal(asm._row_nnz, g["row_nnz"])
    K = asm.assemble_stiffness_matrix()
    assert np.array_equal(K.indptr.cpu().numpy(), g["K_indptr"])
    assert np.array_equal(K.indices.cpu().numpy(), g["K_indices"])
    eK = np.abs(K.data.cpu().numpy() - g["K_data"]).max() / np.abs(g["K_data"]).max()
    bcm = BoundaryConditionManager(K, g["F"], dof_per_node=6)
    bcm.apply_dirichlet([DirichletCondition(g["fixed"].tolist(), 0.0)])
    K_red, F_red, _ = bcm.reduced_system
    u = bcm.expand_solution(spsolve(K_red.to_scipy().tocsc(), F_red.cpu().numpy())).cpu().numpy()
    eu = np.abs(u - g["u"]).max() / np.abs(g["u"]).max()
    floor = float(g["u_floor"])
    print(name, "K", eK, "(reference 1-ulp floor", float(g["K_floor"]), ") u", eu, "(reference 1-ulp floor", floor, ")")
    assert eK <= 1e-12
    assert eu <= max(1e-12, 10.0 * floor)
    if name == "asm_hemisphere_16":
        disp = (abs(u[int(g["node0"]) * 6]) + abs(u[int(g["node90"]) * 6 + 1])) / 2.0
    else:
        disp = abs(u[int(g["node"]) * 6 + (2 if name == "asm_twisted_16x96" else 0)])
    assert abs(disp / float(g["wref"]) - float(g["norm_disp"])) <= 1e-8
    assert abs(disp / float(g["wref"]) - EXPECTED[name]) < 0.05 * EXPECTED[name]   # Ko-Lee-Bathe 2017 table value
