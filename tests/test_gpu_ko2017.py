"""GPU known-answer tests: the reference's own Ko-Lee-Bathe (2017) benchmark cases
(tests/benchmarks/test_ko2017_performance.py: square plate :585-749, circular plate :752-849, pinched cylinder
:919-1024, Scordelis-Lo :1031-1185, twisted beam :1195-1357, hemisphere with cut-out :1663-1797, full hemisphere
:1800-1913, hyperbolic paraboloid :1916-2030; the Raasch hook needs a gmsh mesh and only asserts norm > 0.1) re-expressed against this package's boundary -- MeshAssembler (HIP element
kernels + CSR assembly) -> BoundaryConditionManager (device Dirichlet reduction) -> SciPy spsolve,
the solve route of the reference tests (:401-408).  Same meshes (N = 16), loads, constraints, table
values and the same 5 % acceptance band."""
import numpy as np
import pytest

import fem_shell_amd as fsa
from fem_shell_amd import meshes
from fem_shell_amd.assembler import MeshAssembler
from fem_shell_amd.bc import BoundaryConditionManager, DirichletCondition
from fem_shell_amd.elements import ElementFamily

pytestmark = pytest.mark.gpu
DOF = 6


def _solve(mesh, mat, t, F, fixed):
    from scipy.sparse.linalg import spsolve
    asm = MeshAssembler(mesh, {"elements": {"element_family": ElementFamily.SHELL, "material": mat, "thickness": t}})
    bcm = BoundaryConditionManager(asm.assemble_stiffness_matrix(), F, dof_per_node=DOF)
    bcm.apply_dirichlet([DirichletCondition(fixed, 0.0)])
    K_red, F_red, _ = bcm.reduced_system
    return bcm.expand_solution(spsolve(K_red.to_scipy().tocsc(), F_red.cpu().numpy())).cpu().numpy()


def _nearest(mesh, p):
    return int(np.argmin(np.abs(mesh.coords - np.asarray(p, float)).sum(axis=1)))


def _cyl_fixed(mesh, length, with_x0_plane):
    fixed = []
    for i, (x, y, z) in enumerate(mesh.coords):
        if abs(z) < 1e-6:
            fixed += [i * DOF + 2, i * DOF + 3, i * DOF + 4]
        if abs(z - length) < 1e-6:
            fixed += [i * DOF + 0, i * DOF + 1, i * DOF + 5]
        if abs(y) < 1e-6 and x > 0.0:
            fixed += [i * DOF + 1, i * DOF + 3, i * DOF + 5]
        if with_x0_plane and abs(x) < 1e-6 and y > 0.0:
            fixed += [i * DOF + 0, i * DOF + 4, i * DOF + 5]
    return sorted(set(fixed))


@pytest.mark.parametrize("element,expected", [("MITC4", 0.9313), ("MITC3", 0.9308)])
def test_pinched_cylinder(element, expected):
    R, L, t, wref = 300.0, 300.0, 3.0, 1.8248e-5
    mesh = meshes.cylindrical_patch(R, L, 90.0, 16, 16, triangular=(element == "MITC3"))
    F = np.zeros(mesh.node_count * DOF)
    node = _nearest(mesh, (R, 0.0, 0.0))
    F[node * DOF] = -0.25
    u = _solve(mesh, fsa.IsotropicMaterial("Ko2017_Cylinder", 3.0e6, 0.3, 1.0), t, F, _cyl_fixed(mesh, L, True))
    norm = abs(u[node * DOF]) / wref
    print("pinched cylinder", element, norm)
    assert np.isclose(norm, expected, rtol=0.05)


@pytest.mark.parametrize("element,expected", [("MITC4", 0.9973), ("MITC3", 0.9550)])
def test_scordelis_lo(element, expected):
    R, L, ang, t, wref, rho = 25.0, 25.0, 40.0, 0.25, 3.0240e-1, 360.0
    mesh = meshes.cylindrical_patch(R, L, ang, 16, 16, triangular=(element == "MITC3"))
    F = np.zeros(mesh.node_count * DOF)
    for row in mesh.conn:                     # gravity along -x, lumped equally to the element's nodes
        c = mesh.coords[row]
        area = 0.5 * np.linalg.norm(np.cross(c[1] - c[0], c[2] - c[0]))
        if len(row) == 4:
            area += 0.5 * np.linalg.norm(np.cross(c[2] - c[0], c[3] - c[0]))
        F[row * DOF] -= area * t * rho / len(row)
    u = _solve(mesh, fsa.IsotropicMaterial("Ko2017_Scordelis", 4.32e8, 0.0, rho), t, F, _cyl_fixed(mesh, L, False))
    node = _nearest(mesh, (R * np.cos(np.radians(ang)), R * np.sin(np.radians(ang)), 0.0))
    norm = abs(u[node * DOF]) / wref
    print("scordelis-lo", element, norm)
    assert np.isclose(norm, expected, rtol=0.05)


@pytest.mark.parametrize("element", ["MITC4", "MITC3"])
@pytest.mark.parametrize("t_over_L,case,P,wref,exp3,exp4", [
    (0.02667, "In-plane", 1.0, 5.4240e-3, 0.9965, 1.02),
    (0.02667, "Out-of-plane", 1.0, 1.7540e-3, 0.9912, 0.99),
    (0.0002667, "In-plane", 1.0e-6, 5.2560e-3, 0.9963, 0.92),
    (0.0002667, "Out-of-plane", 1.0e-6, 1.2940e-3, 0.9975, 0.92),
])
def test_twisted_beam(element, t_over_L, case, P, wref, exp3, exp4):
    length, width = 12.0, 1.1
    mesh = meshes.twisted_beam(length, width, 90.0, 96, 16, triangular=(element == "MITC3"))
    dof = 2 if case == "In-plane" else 1
    tip = _nearest(mesh, (length, 0.0, 0.0))
    F = np.zeros(mesh.node_count * DOF)
    F[tip * DOF + dof] = P
    fixed = [i * DOF + d for i in np.flatnonzero(np.abs(mesh.coords[:, 0]) < 1e-6) for d in range(DOF)]
    u = _solve(mesh, fsa.IsotropicMaterial("Ko2017_TwistedBeam", 29.0e6, 0.22, 1.0), length * t_over_L, F, fixed)
    norm = abs(u[tip * DOF + dof]) / wref
    print("twisted beam", element, t_over_L, case, norm)
    assert np.isclose(norm, exp3 if element == "MITC3" else exp4, rtol=0.05)


# ---------------------------------------------------------------------------------------------------
# The remaining cases of the reference's benchmark file, same set-ups: square plate (Tables 2-5, :585-749),
# circular plate (Tables 6-7, :752-849), hemisphere with an 18 degree cut-out (Tables 15-16, :1663-1797),
# full hemisphere (Table 17, :1800-1913), hyperbolic paraboloid (Tables 18-19, :1916-2030).
# ---------------------------------------------------------------------------------------------------
def _ratio_positions(n):
    """Node positions in [0, 1] with segment lengths 1 : 2 : ... : n (the paper's distorted meshes, :123-129)."""
    cum = np.concatenate([[0.0], np.cumsum(np.arange(1, n + 1, dtype=float))])
    return cum / cum[-1]


def _pressure_load(mesh, pressure, dof=2, sign=1.0):
    """Uniform pressure / gravity lumped equally to the element's nodes (:411-441, :1061-1106)."""
    F = np.zeros(mesh.node_count * DOF)
    for row in mesh.conn:
        c = mesh.coords[row]
        area = 0.5 * np.linalg.norm(np.cross(c[1] - c[0], c[2] - c[0]))
        if len(row) == 4:
            area += 0.5 * np.linalg.norm(np.cross(c[2] - c[0], c[3] - c[0]))
        F[row * DOF + dof] += sign * pressure * area / len(row)
    return F


def _sym_fixed(mesh, x_sym, y_sym, tol=1e-9):
    fixed = []
    for i, (x, y, _) in enumerate(mesh.coords):
        if abs(x - x_sym) < tol:
            fixed += [i * DOF + 0, i * DOF + 4, i * DOF + 5]
        if abs(y - y_sym) < tol:
            fixed += [i * DOF + 1, i * DOF + 3, i * DOF + 5]
    return fixed


def _edge_fixed(mesh, dofs, x=None, y=None, tol=1e-9):
    fixed = []
    for i, (xx, yy, _) in enumerate(mesh.coords):
        if (x is not None and abs(xx - x) < tol) or (y is not None and abs(yy - y) < tol):
            fixed += [i * DOF + d for d in dofs]
    return fixed


PLATE = dict(E=1.0e4, nu=0.3)


@pytest.mark.parametrize("t_over_L,pressure", [(1 / 100, 1.0e2), (1 / 1000, 1.0e5), (1 / 10000, 1.0e8)])
@pytest.mark.parametrize("distorted", [False, True])
@pytest.mark.parametrize("element,exp_clamped,exp_ss", [
    ("MITC4", {False: {1 / 100: 0.9984, 1 / 1000: 0.9980, 1 / 10000: 0.9979}, True: {1 / 100: 1.002, 1 / 1000: 1.001, 1 / 10000: 1.001}},
     {False: {1 / 100: 1.000, 1 / 1000: 0.9998, 1 / 10000: 0.9998}, True: {1 / 100: 1.003, 1 / 1000: 1.003, 1 / 10000: 1.003}}),
    ("MITC3", {False: {1 / 100: 0.9947, 1 / 1000: 0.9943, 1 / 10000: 0.9942}, True: {1 / 100: 1.000, 1 / 1000: 1.000, 1 / 10000: 1.000}},
     {False: {1 / 100: 1.002, 1 / 1000: 1.001, 1 / 10000: 1.001}, True: {1 / 100: 1.004, 1 / 1000: 1.003, 1 / 10000: 1.003}}),
])
def test_square_plate(t_over_L, pressure, distorted, element, exp_clamped, exp_ss):
    n, L, half = 16, 1.0, 0.5
    t = L * t_over_L
    mesh = meshes.flat_patch(0.0, half, 0.0, half, n, n, triangular=(element == "MITC3"))
    if distorted:
        lev = _ratio_positions(n) * half
        X, Y = np.meshgrid(lev, lev)
        mesh.coords[:, 0], mesh.coords[:, 1] = X.reshape(-1), Y.reshape(-1)
    mat = fsa.IsotropicMaterial("Ko2017_Plate", PLATE["E"], PLATE["nu"], 1.0)
    D = PLATE["E"] * t ** 3 / (12.0 * (1.0 - PLATE["nu"] ** 2))
    F = _pressure_load(mesh, pressure)
    centre = _nearest(mesh, (half, half, 0.0))
    for name, alpha, dofs, expected in (("clamped", 1.267e-3, range(6), exp_clamped), ("ss", 4.062e-3, (0, 1, 2), exp_ss)):
        fixed = _edge_fixed(mesh, dofs, x=0.0) + _edge_fixed(mesh, dofs, y=0.0) + _sym_fixed(mesh, half, half)
        u = _solve(mesh, mat, t, F, sorted(set(fixed)))
        norm = abs(u[centre * DOF + 2]) / (alpha * pressure * L ** 4 / D)
        print("square plate", name, element, "distorted" if distorted else "regular", t_over_L, norm)
        assert np.isclose(norm, expected[distorted][t_over_L], rtol=0.05)


def _quarter_disk(radius, n, triangular):
    """Unit square -> quarter disk, concentric map (:311-362)."""
    mesh = meshes.flat_patch(0.0, 1.0, 0.0, 1.0, n, n, triangular=triangular)
    for k, (x, y, _) in enumerate(mesh.coords.copy()):
        if x == 0.0 and y == 0.0:
            xx = yy = 0.0
        elif x >= y:
            rr, phi = radius * x, (np.pi / 4.0) * (y / x)
            xx, yy = rr * np.cos(phi), rr * np.sin(phi)
        else:
            rr, phi = radius * y, (np.pi / 2.0) - (np.pi / 4.0) * (x / y)
            xx, yy = rr * np.cos(phi), rr * np.sin(phi)
        mesh.coords[k] = (xx, yy, 0.0)
    return mesh


@pytest.mark.parametrize("t_over_L,pressure,exp3,exp4", [(1 / 100, 1.0e2, 0.9989, 1.001), (1 / 1000, 1.0e5, 0.9974, 0.9997),
                                                       (1 / 10000, 1.0e8, 0.9974, 0.9997)])
@pytest.mark.parametrize("clamped", [True, False])
@pytest.mark.parametrize("element", ["MITC3", "MITC4"])
def test_circular_plate(t_over_L, pressure, exp3, exp4, clamped, element):
    n, R = 16, 1.0
    t = R * t_over_L
    mesh = _quarter_disk(R, n, element == "MITC3")
    fixed = _sym_fixed(mesh, 0.0, 0.0)
    for i, (x, y, _) in enumerate(mesh.coords):
        if abs(np.hypot(x, y) - R) < 1e-6:
            fixed += [i * DOF + d for d in (range(6) if clamped else (0, 1, 2))]
    alpha = 1.0 / 64.0 if clamped else (5.0 + 0.3) / (64.0 * (1.0 + 0.3))
    D = PLATE["E"] * t ** 3 / (12.0 * (1.0 - PLATE["nu"] ** 2))
    u = _solve(mesh, fsa.IsotropicMaterial("Ko2017_Plate", PLATE["E"], PLATE["nu"], 1.0), t, _pressure_load(mesh, pressure),
               sorted(set(fixed)))
    norm = abs(u[_nearest(mesh, (0.0, 0.0, 0.0)) * DOF + 2]) / (alpha * pressure * R ** 4 / D)
    print("circular plate", "clamped" if clamped else "ss", element, t_over_L, norm)
    assert np.isclose(norm, exp3 if element == "MITC3" else exp4, rtol=0.05)


def _sphere_fixed(mesh, R, tol=1e-6):
    fixed = []
    for i, (x, y, z) in enumerate(mesh.coords):
        if abs(y) < tol and x >= 0.0:
            fixed += [i * DOF + 1, i * DOF + 3, i * DOF + 5]
        if abs(x) < tol and y >= 0.0:
            fixed += [i * DOF + 0, i * DOF + 4, i * DOF + 5]
        if abs(x - R) < 1e-2 and abs(y) < 1e-2 and abs(z) < 1e-2:
            fixed.append(i * DOF + 2)
    return sorted(set(fixed))


def _hemisphere(theta_min, n, triangular, distorted, R=10.0):
    mesh = meshes.sphere_patch(R, theta_min, 90.0, 90.0, n, triangular=triangular)
    if distorted:                                    # theta / phi levels remapped to the 1:2:...:n spacing (:1607-1636)
        th = np.radians(theta_min) + _ratio_positions(n) * (np.radians(90.0) - np.radians(theta_min))
        ph = _ratio_positions(n) * np.radians(90.0)
        TH, PH = np.meshgrid(th, ph, indexing="ij")
        mesh.coords[:] = np.stack([R * np.sin(TH) * np.cos(PH), R * np.sin(TH) * np.sin(PH), R * np.cos(TH)], -1).reshape(-1, 3)
    return mesh


@pytest.mark.parametrize("distorted", [False, True])
@pytest.mark.parametrize("t_over_R,P,exp3,exp4", [(4 / 1000, 2.0, {False: 1.007, True: 1.009}, {False: 1.009, True: 0.9958}),
                                                  (4 / 10000, 2.0e-3, {False: 0.9994, True: 0.9949}, {False: 0.9811, True: 0.9736})])
@pytest.mark.parametrize("element", ["MITC3", "MITC4"])
def test_hemisphere_cutout(distorted, t_over_R, P, exp3, exp4, element):
    R, uref = 10.0, 9.3000e-2
    mesh = _hemisphere(18.0, 16, element == "MITC3", distorted)
    a, b = _nearest(mesh, (R, 0.0, 0.0)), _nearest(mesh, (0.0, R, 0.0))
    F = np.zeros(mesh.node_count * DOF)
    F[a * DOF + 0] = P / 2.0
    F[b * DOF + 1] = -P / 2.0
    u = _solve(mesh, fsa.IsotropicMaterial("Ko2017_Sphere", 6.825e7, 0.3, 1.0), R * t_over_R, F, _sphere_fixed(mesh, R))
    norm = 0.5 * (abs(u[a * DOF]) + abs(u[b * DOF + 1])) / uref
    print("hemisphere cut-out", element, "distorted" if distorted else "regular", t_over_R, norm)
    assert np.isclose(norm, (exp3 if element == "MITC3" else exp4)[distorted], rtol=0.05)


@pytest.mark.parametrize("t_over_R,P,exp3,exp4", [(4 / 1000, 2.0, 0.9994, 0.9960), (4 / 10000, 2.0e-3, 0.9956, 0.9798)])
@pytest.mark.parametrize("element", ["MITC3", "MITC4"])
def test_full_hemisphere(t_over_R, P, exp3, exp4, element):
    R, uref = 10.0, 9.2400e-2
    mesh = _hemisphere(2.0, 16, element == "MITC3", False)       # theta_min = 2 degrees: no degenerate pole elements
    a, b = _nearest(mesh, (R, 0.0, 0.0)), _nearest(mesh, (0.0, R, 0.0))
    F = np.zeros(mesh.node_count * DOF)
    F[a * DOF + 0] = P / 2.0
    F[b * DOF + 1] = -P / 2.0
    u = _solve(mesh, fsa.IsotropicMaterial("Ko2017_Sphere", 6.825e7, 0.3, 1.0), R * t_over_R, F, _sphere_fixed(mesh, R))
    norm = abs(u[a * DOF]) / uref
    print("full hemisphere", element, t_over_R, norm)
    assert np.isclose(norm, exp3 if element == "MITC3" else exp4, rtol=0.05)


@pytest.mark.parametrize("distorted", [False, True])
@pytest.mark.parametrize("t_over_L,rho,exp3,exp4,wref", [
    (1 / 1000, 360.0, {False: 0.9728, True: 0.9483}, {False: 0.9762, True: 0.9904}, 2.8780e-4),
    (1 / 10000, 3.6, {False: 0.9828, True: 0.9358}, {False: 0.9777, True: 0.9936}, 2.3856e-4),
])
@pytest.mark.parametrize("element", ["MITC3", "MITC4"])
def test_hyperbolic_paraboloid(distorted, t_over_L, rho, exp3, exp4, wref, element):
    n = 32
    t = 1.0 * t_over_L
    surf = lambda x, y: y * y - x * x
    mesh = meshes.flat_patch(-0.5, 0.5, -0.5, 0.5, n, n, triangular=(element == "MITC3"), surface_z=surf)
    if distorted:
        lev = -0.5 + _ratio_positions(n)
        X, Y = np.meshgrid(lev, lev)
        mesh.coords[:] = np.stack([X, Y, surf(X, Y)], -1).reshape(-1, 3)
    F = _pressure_load(mesh, rho * 1.0 * t, sign=-1.0)
    fixed = _edge_fixed(mesh, range(6), x=-0.5, tol=1e-6)
    u = _solve(mesh, fsa.IsotropicMaterial("Ko2017_Hyperbolic", 2.0e11, 0.3, 1.0), t, F, sorted(set(fixed)))
    node = _nearest(mesh, (0.5, 0.0, surf(0.5, 0.0)))
    norm = abs(u[node * DOF + 2]) / wref
    print("hyperbolic paraboloid", element, "distorted" if distorted else "regular", t_over_L, norm)
    # Two distorted cases miss the paper's 5 % band IN THE REFERENCE ITSELF (its own MITC4 / MITC3 classes driven through
    # tools/ref_shim.py on this mesh give 1.0423614 and 0.9862832, i.e. 5.2 % / 5.4 % off the table): parity is with the
    # reference, so these two are pinned to the reference's own numbers (ill-conditioned solves: 1e-4).
    own = {("MITC4", 1 / 1000, True): 1.0423614, ("MITC3", 1 / 10000, True): 0.9862832}.get((element, t_over_L, distorted))
    if own is not None:
        assert np.isclose(norm, own, rtol=1e-4)
    else:
        assert np.isclose(norm, (exp3 if element == "MITC3" else exp4)[distorted], rtol=0.05)
