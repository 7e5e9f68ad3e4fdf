"""GPU: fs_inertial_forces / InertialForcesCalculator against the reference fixture (tests/golden/inertial_ref.npz) and,
for configs[2] (blade-like laminate tube), the rotor load chain M -> lumped nodal masses -> centrifugal forces at
omega = 0.79 rad/s about z against the oracle."""
import numpy as np
import pytest
import torch

import fem_shell_amd as fsa
from fem_shell_amd import meshes
from fem_shell_amd.elements import ElementFamily
from fem_shell_amd.inertial import InertialForcesCalculator
from oracle import inertial as I
from tests import util
from tests.util_golden import golden

pytestmark = pytest.mark.gpu


def test_inertial_forces_vs_reference_fixture():
    g = golden("inertial_ref")
    calc = InertialForcesCalculator(g["axis"], g["center"])
    assert np.allclose(np.linalg.norm(calc.axis), 1.0) and np.array_equal(calc.center, g["center"])
    omega, alpha = float(g["omega"]), float(g["alpha"])
    scale = np.abs(g["F_all"]).max()
    for got, key in ((calc.compute_centrifugal_force(g["coords"], g["mass"], omega), "F_cf"),
                     (calc.compute_coriolis_force(g["vel"], g["mass"], omega), "F_cor"),
                     (calc.compute_euler_force(g["coords"], g["mass"], alpha), "F_eul")):
        assert np.abs(got.cpu().numpy() - g[key]).max() <= 2e-15 * scale, key
    F, diag = calc.compute_all_inertial_forces(g["coords"], g["vel"], g["mass"], omega, alpha)
    assert np.abs(F.cpu().numpy() - g["F_all"]).max() <= 2e-15 * scale
    for k in ("centrifugal", "coriolis", "euler", "total_inertial"):
        ref = g["diag_" + k]
        assert np.allclose([diag[k]["total"], diag[k]["max_nodal"], diag[k]["mean_nodal"]], ref, rtol=1e-12)
    # omega = 0 / alpha = 0: zero forces (corotational.py:374-375, 418-419, 456-457); part selection
    assert float(calc.compute_centrifugal_force(g["coords"], g["mass"], 0.0).abs().max()) == 0.0
    assert float(calc.compute_euler_force(g["coords"], g["mass"], 0.0).abs().max()) == 0.0
    F2, d2 = calc.compute_all_inertial_forces(g["coords"], g["vel"], g["mass"], omega, alpha, include_coriolis=False)
    assert "coriolis" not in d2 and np.abs(F2.cpu().numpy() - (g["F_cf"] + g["F_eul"])).max() <= 2e-15 * scale
    with pytest.raises(ValueError):
        InertialForcesCalculator([0.0, 0.0, 0.0])


def test_c3_rotor_centrifugal_load_from_lumped_mass():
    from fem_shell_amd.assembler import MeshAssembler
    mesh = meshes.tapered_tube(64, 128)
    lam = fsa.create_laminate_from_angles(util.t300(), 0.5e-3, [0.0, 45.0, -45.0, 90.0, 90.0, -45.0, 45.0, 0.0])
    asm = MeshAssembler(mesh, {"elements": {"element_family": ElementFamily.SHELL, "laminate": lam}})
    lumped = asm.assemble_mass_matrix().lump_rows()                   # row sums of M (linear_dynamic.py:227-234)
    masses = lumped[0::6].contiguous()                                # translational mass per node
    calc = InertialForcesCalculator([0.0, 0.0, 1.0], [0.0, 0.0, 0.0])
    F = calc.compute_centrifugal_force(torch.from_numpy(mesh.coords).to(masses.device), masses, 0.79)
    ref, _ = I.inertial_forces(mesh.coords, None, masses.cpu().numpy(), 0.79, 0.0, [0, 0, 1], [0, 0, 0], coriolis=False,
                               euler=False)
    assert np.abs(F.cpu().numpy() - ref).max() <= 2e-15 * np.abs(ref).max()
    assert float(F[:, 2].abs().max()) == 0.0                          # no force along the axis
    # total mass of the tube = rho h area: sum of the lumped masses against sum m_t * element area
    c = mesh.coords[mesh.conn]
    area = 0.5 * (np.linalg.norm(np.cross(c[:, 1] - c[:, 0], c[:, 2] - c[:, 0]), axis=1)
                  + np.linalg.norm(np.cross(c[:, 2] - c[:, 0], c[:, 3] - c[:, 0]), axis=1))
    assert abs(float(masses.sum()) / (float(asm._prop_table["m_t"][0]) * area.sum()) - 1.0) < 2e-3   # warped quads
