"""CPU: the oracle's inertial-force restatement against the fixture generated from the unmodified reference."""
import numpy as np

from oracle import inertial as I
from tests.util_golden import golden


def test_inertial_oracle_vs_reference_fixture():
    g = golden("inertial_ref")
    F, parts = I.inertial_forces(g["coords"], g["vel"], g["mass"], float(g["omega"]), float(g["alpha"]), g["axis"], g["center"])
    scale = np.abs(g["F_all"]).max()
    assert np.abs(parts["centrifugal"] - g["F_cf"]).max() <= 1e-15 * scale
    assert np.abs(parts["coriolis"] - g["F_cor"]).max() <= 1e-15 * scale
    assert np.abs(parts["euler"] - g["F_eul"]).max() <= 1e-15 * scale
    assert np.abs(F - g["F_all"]).max() <= 1e-15 * scale
