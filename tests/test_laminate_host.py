"""CPU: the known-answer checks of the reference's tests/test_laminate.py (Q / Qbar identities :114-235, laminate
ABD properties :262-368, unidirectional closed forms and consistency :777-834) re-expressed against
fem_shell_amd.laminate -- the host code that fills the fs_shell_prop rows the kernels read."""
import numpy as np
import pytest

import fem_shell_amd as fsa
from fem_shell_amd.laminate import (Laminate, Ply, compute_Q, compute_Qbar, compute_shear_Cbar, create_laminate_from_angles,
                                    create_symmetric_laminate, cross_ply_layup, quasi_isotropic_layup)

T = 0.125e-3


@pytest.fixture
def cfrp():
    return fsa.OrthotropicMaterial(name="Carbon/Epoxy T300/5208", E=(181e9, 10.3e9, 10.3e9), G=(7.17e9, 3.78e9, 7.17e9),
                                   nu=(0.28, 0.28, 0.28), rho=1600)


def test_Q(cfrp):
    Q = compute_Q(cfrp)
    E1, E2, G12, nu12 = 181e9, 10.3e9, 7.17e9, 0.28
    nu21 = nu12 * E2 / E1
    den = 1.0 - nu12 * nu21
    assert np.allclose(Q, Q.T)
    assert np.isclose(Q[0, 0], E1 / den, rtol=1e-12) and np.isclose(Q[1, 1], E2 / den, rtol=1e-12)
    assert np.isclose(Q[0, 1], nu12 * E2 / den, rtol=1e-12) and np.isclose(Q[2, 2], G12, rtol=1e-12)
    assert np.all(np.linalg.eigvalsh(Q) > 0)


def test_Qbar(cfrp):
    Q = compute_Q(cfrp)
    assert np.allclose(compute_Qbar(cfrp, 0.0), Q, rtol=1e-12)
    Q90 = compute_Qbar(cfrp, 90.0)
    assert np.isclose(Q90[0, 0], Q[1, 1], rtol=1e-10) and np.isclose(Q90[1, 1], Q[0, 0], rtol=1e-10)
    for ang in (15.0, 30.0, 45.0, 60.0):
        Qb = compute_Qbar(cfrp, ang)
        assert np.allclose(Qb, Qb.T)
        Qm = compute_Qbar(cfrp, -ang)                              # +-theta: Q16, Q26 change sign, the rest agrees
        assert np.isclose(Qb[0, 0], Qm[0, 0]) and np.isclose(Qb[0, 2], -Qm[0, 2]) and np.isclose(Qb[1, 2], -Qm[1, 2])
    assert abs(compute_Qbar(cfrp, 45.0)[0, 2]) > 0.0
    tol = 1e-10 * Q.max()
    assert abs(compute_Qbar(cfrp, 0.0)[0, 2]) < tol and abs(Q90[0, 2]) < tol and abs(Q90[1, 2]) < tol
    C0 = compute_shear_Cbar(cfrp, 0.0)
    assert np.allclose(C0, C0.T) and np.allclose(compute_shear_Cbar(cfrp, 30.0), compute_shear_Cbar(cfrp, 30.0).T)


def test_laminate_structure(cfrp):
    lam = Laminate([Ply(cfrp, T, 0), Ply(cfrp, T, 90)])
    assert lam.n_plies == 2 and np.isclose(lam.total_thickness, 2 * T)
    assert not lam.is_symmetric and not np.allclose(lam.B, 0)
    plies = [Ply(cfrp, T, a) for a in (0, 90, 90, 0)]
    sym = Laminate(plies)
    h = sym.total_thickness
    assert np.isclose(plies[0].z_bottom, -h / 2) and np.isclose(plies[0].z_top, -h / 2 + T) and np.isclose(plies[-1].z_top, h / 2)
    assert sym.is_symmetric and np.allclose(sym.B, 0, atol=1e-10 * np.abs(sym.A).max())
    bal = Laminate([Ply(cfrp, T, a) for a in (45, -45, -45, 45)])
    assert bal.is_balanced
    assert abs(bal.A[0, 2]) < 1e-10 * np.abs(bal.A).max() and abs(bal.A[1, 2]) < 1e-10 * np.abs(bal.A).max()
    qi = create_laminate_from_angles(cfrp, T, quasi_isotropic_layup(1))
    assert np.isclose(qi.A[0, 0], qi.A[1, 1], rtol=0.01) and qi.is_balanced and qi.is_symmetric
    assert np.all(np.linalg.eigvalsh(qi.A) > 0) and np.all(np.linalg.eigvalsh(qi.D) > 0)
    s = create_symmetric_laminate([Ply(cfrp, T, 0), Ply(cfrp, T, 90)])
    assert s.n_plies == 4 and s.is_symmetric


def test_closed_forms(cfrp):
    ud = create_laminate_from_angles(cfrp, T, [0, 0, 0, 0])
    h, Q = ud.total_thickness, compute_Q(cfrp)
    assert np.allclose(ud.A, Q * h, rtol=1e-10) and np.allclose(ud.D, Q * h ** 3 / 12, rtol=1e-10)
    cp = create_laminate_from_angles(cfrp, T, cross_ply_layup(4))
    assert np.isclose(cp.A[0, 0], cp.A[1, 1], rtol=0.01)
    qi = create_laminate_from_angles(cfrp, T, quasi_isotropic_layup(1))
    a = np.linalg.inv(qi.A)
    assert np.isclose(qi.get_equivalent_properties()["Ex_membrane"], 1.0 / (qi.total_thickness * a[0, 0]), rtol=1e-10)
