"""CPU: pin the oracle (oracle/femshell_oracle.c) against golden vectors produced by the
unmodified reference (tools/gen_golden.py).  Geometry stage bit-exact; matrices <= 1e-13."""
import os

import numpy as np
import pytest

from tests.util_golden import golden, golden_props

TOL = 1e-13


def _rel(a, b):
    scale = np.abs(b).reshape(b.shape[0], -1).max(axis=1)
    return (np.abs(a - b).reshape(b.shape[0], -1).max(axis=1) / scale).max()


def test_geometry_bitwise(oracle):
    g = golden("elem_mitc4")
    for e in range(g["xyz"].shape[0]):
        fr, xl, dj, ac, st = oracle.mitc4_geom(g["xyz"][e])
        assert st == 0
        assert np.array_equal(fr, g["frame"][e]), e
        assert np.array_equal(xl, g["xl"][e]), e
        assert np.array_equal(dj, g["detJ"][e]), e
        assert np.abs(ac - g["acoef"][e]).max() <= 1e-12 * max(1.0, np.abs(g["acoef"][e]).max()), e


def test_K_M_vs_reference(oracle):
    g = golden("elem_mitc4")
    props = golden_props(g)
    pid = np.arange(g["xyz"].shape[0], dtype=np.int32)
    K, st = oracle.mitc4_K_batch(g["xyz"], pid, props)
    M, _ = oracle.mitc4_M_batch(g["xyz"], pid, props)
    assert not st.any()
    print("oracle K vs reference", _rel(K, g["Ke"]), " M", _rel(M, g["Me"]))
    assert _rel(K, g["Ke"]) < TOL
    assert _rel(M, g["Me"]) < TOL


def test_local_parts_vs_reference(oracle):
    g = golden("elem_mitc4")
    props = golden_props(g)
    for e in range(0, g["xyz"].shape[0], 3):
        km, kbs, kd, kc = oracle.mitc4_K_parts(g["xyz"][e], props[e:e + 1])
        ref = g["parts"][e]
        scale = np.abs(ref[:3]).max()
        for a, b in zip((km, kbs, kd, kc), ref):
            assert np.abs(a - b).max() <= TOL * scale, e
    # the coupling routine equals the nodal-only symmetric form: covered by kc above
    assert (g["prop_flags"] & 1).any()


def test_nonlinear_vs_reference(oracle):
    g = golden("elem_nl")
    props = golden_props(g)
    assert props["nonlinear"].min() == 0 and props["nonlinear"].max() == 1
    for e in range(g["xyz"].shape[0]):
        KT, f, st = oracle.mitc4_NL(g["xyz"][e], props[e:e + 1], g["u"][e])
        assert st == 0
        assert np.abs(KT - g["KT"][e]).max() <= 1e-12 * np.abs(g["KT"][e]).max(), e
        assert np.abs(f - g["f_int"][e]).max() <= 1e-11 * np.abs(g["f_int"][e]).max(), e


def test_assembly_scordelis(oracle):
    """Pattern bit-exact and values vs the reference tests' COO->CSR route
    (tests/benchmarks/test_ko2017_performance.py:365-398), plus the known answer 0.9973 +-5 %."""
    import fem_shell_amd as fsa
    g = golden("asm_scordelis_16")
    table = fsa.build_prop_table([fsa.ShellProperty(
        fsa.IsotropicMaterial("sc", float(g["E"]), float(g["nu"]), float(g["rho"])), float(g["t"]))])
    conn, coords = g["conn"], g["coords"]
    pid = np.zeros(conn.shape[0], dtype=np.int32)
    Ke, _ = oracle.mitc4_K_batch(coords[conn], pid, table)
    Me, _ = oracle.mitc4_M_batch(coords[conn], pid, table)
    dofs = (conn[:, :, None] * 6 + np.arange(6)).reshape(conn.shape[0], -1)
    ip, ix, kd = oracle.csr_assemble(dofs, coords.shape[0] * 6, Ke)
    _, _, md = oracle.csr_assemble(dofs, coords.shape[0] * 6, Me)
    assert np.array_equal(ip, g["K_indptr"]) and np.array_equal(ix, g["K_indices"])
    assert ix.shape[0] == 36 * 49 * 49 == 86436
    assert np.abs(kd - g["K_data"]).max() <= 1e-13 * np.abs(g["K_data"]).max()
    assert np.abs(md - g["M_data"]).max() <= 1e-13 * np.abs(g["M_data"]).max()
    assert abs(float(g["norm_disp"]) - 0.9973) < 0.05 * 0.9973
    # linear-static displacement from the oracle-assembled K (north_star: displacements match)
    import scipy.sparse as sp
    from scipy.sparse.linalg import spsolve
    n = coords.shape[0] * 6
    K = sp.csr_matrix((kd, ix, ip), shape=(n, n))
    free = np.setdiff1d(np.arange(n), g["fixed"])
    u = np.zeros(n)
    u[free] = spsolve(K[free, :][:, free].tocsc(), g["F"][free])
    err = np.abs(u - g["u"]).max() / np.abs(g["u"]).max()
    # the reference's own sensitivity of u to a +-1 ulp change of its input coordinates (tools/gen_golden_asm_ko.py)
    floor = float(golden("asm_scordelis_16_floor")["u_floor"])
    print("displacement rel diff vs reference solve", err, "reference 1-ulp floor", floor)
    assert err <= max(1e-12, 10.0 * floor)


@pytest.mark.parametrize("name", ["asm_scordelis_32", "asm_hemisphere_16", "asm_twisted_16x96"])
def test_assembly_ko_goldens(oracle, name):
    """SURVEY.md 8(c)-6 fixtures from the reference's UNMODIFIED MeshAssembler (tools/gen_golden_asm_ko.py): DOF map,
    _row_nnz and pattern bit-exact, values 1e-12, displacement within max(1e-12, 10 x the reference's own 1-ulp
    sensitivity), known answers of Ko-Lee-Bathe 2017 within the reference's 5 %."""
    import fem_shell_amd as fsa
    import scipy.sparse as sp
    from scipy.sparse.linalg import spsolve
    g = golden(name)
    table = fsa.build_prop_table([fsa.ShellProperty(
        fsa.IsotropicMaterial("m", float(g["E"]), float(g["nu"]), float(g["rho"])), float(g["t"]))])
    conn, coords = g["conn"], g["coords"]
    n = coords.shape[0] * 6
    assert n == int(g["dofs_count"])
    pid = np.zeros(conn.shape[0], dtype=np.int32)
    Ke, st = oracle.mitc4_K_batch(coords[conn], pid, table)
    assert not st.any()
    dofs = (conn[:, :, None] * 6 + np.arange(6)).reshape(conn.shape[0], -1)
    assert np.array_equal(dofs, g["dofs_array"])
    ip, ix, kd = oracle.csr_assemble(dofs, n, Ke)
    assert np.array_equal(np.diff(ip), g["row_nnz"])
    assert np.array_equal(ip, g["K_indptr"]) and np.array_equal(ix, g["K_indices"])
    eK = np.abs(kd - g["K_data"]).max() / np.abs(g["K_data"]).max()
    K = sp.csr_matrix((kd, ix, ip), shape=(n, n))
    free = np.setdiff1d(np.arange(n), g["fixed"])
    u = np.zeros(n)
    u[free] = spsolve(K[free, :][:, free].tocsc(), g["F"][free])
    eu = np.abs(u - g["u"]).max() / np.abs(g["u"]).max()
    print(name, "K", eK, "(reference 1-ulp floor", float(g["K_floor"]), ") u", eu, "(floor", float(g["u_floor"]), ")")
    assert eK <= 1e-12
    assert eu <= max(1e-12, 10.0 * float(g["u_floor"]))
    expected = {"asm_scordelis_32": 0.9973, "asm_hemisphere_16": 1.009, "asm_twisted_16x96": 1.02}[name]
    assert abs(float(g["norm_disp"]) - expected) < 0.05 * expected      # test_ko2017_performance.py:1115,1672,1256


def test_mitc3_K_M_vs_reference(oracle):
    """MITC3+ / MITC3Composite (elements/MITC3.py:1077-1138, MITC3_composite.py:165-240)."""
    g = golden("elem_mitc3")
    props = golden_props(g)
    assert (g["prop_flags"] & 1).any() and (g["kind"] == 0).any()
    pid = np.arange(g["xyz"].shape[0], dtype=np.int32)
    K, M, st = oracle.mitc3_batch(g["xyz"], pid, props, want_K=True, want_M=True)
    assert not st.any()
    print("oracle MITC3 K vs reference", _rel(K, g["Ke"]), " M", _rel(M, g["Me"]))
    assert _rel(K, g["Ke"]) < TOL
    assert _rel(M, g["Me"]) < TOL
    # property rows: the MITC3 classes evaluate the same constants as the MITC4 ones
    import fem_shell_amd as fsa
    assert np.allclose(g["Ke"], np.transpose(g["Ke"], (0, 2, 1)), rtol=0, atol=0)


@pytest.mark.parametrize("nn", [8, 4, 6, 5])
def test_solids_vs_reference(oracle, nn):
    """HEXA8 / TETRA4 / WEDGE6 / PYRAMID5 K, M, body_load, K_G (elements/SOLID.py:322-374,543-631),
    including HEXA8 reduced integration, rotated orthotropic C and the WEDGE6 x2 weight quirk."""
    g = golden("elem_solid")
    xyz, props, red = g[f"xyz{nn}"], g[f"props{nn}"], g[f"reduced{nn}"]
    pid = np.arange(xyz.shape[0], dtype=np.int32)
    for r in (False, True):
        sel = np.flatnonzero(red == r)
        if not sel.size:
            continue
        o = oracle.solid_batch(nn, xyz[sel], pid[sel], props, want_K=True, want_M=True, body=g["body"],
                               sigma=g[f"sigma{nn}"][sel], reduced=r)
        assert not o["status"].any()
        for k, ref in (("K", "K"), ("M", "M"), ("f", "F"), ("KG", "KG")):
            err = _rel(o[k], g[f"{ref}{nn}"][sel])
            print(nn, "reduced" if r else "full", k, err)
            assert err < 5e-13      # the reference's own J = dN @ x on absolute coordinates is ~1e-13
    if nn == 6:   # quirk pin: reference WEDGE6 mass is twice rho * volume
        x = xyz[0]
        vol = 0.5 * abs(np.linalg.det(np.stack([x[1] - x[0], x[2] - x[0], x[3] - x[0]])))
        assert abs(g["M6"][0][0::3, 0::3].sum() / (props[0, 36] * vol) - 2.0) < 1e-12


@pytest.mark.parametrize("nn", [10, 13, 15, 20])
def test_quadratic_solids_vs_reference(oracle, nn):
    """TETRA10 / PYRAMID13 / WEDGE15 / HEXA20 K, M, body_load, K_G (elements/SOLID.py:832-1059, 1221-1537, 1600-1708,
    1821-1946).  HEXA20 uses the reference's forward-difference derivatives (h = 1e-8): parity needs bit-identical
    shape values, which the restatement reproduces (same operation order, libm pow)."""
    g = golden("elem_pyr13" if nn == 13 else "elem_solid2")
    xyz, props = g[f"xyz{nn}"], g[f"props{nn}"]
    pid = np.arange(xyz.shape[0], dtype=np.int32)
    o = oracle.solid_batch(nn, xyz, pid, props, want_K=True, want_M=True, body=g["body"], sigma=g[f"sigma{nn}"])
    assert not o["status"].any()
    for k, ref in (("K", "K"), ("M", "M"), ("f", "F"), ("KG", "KG")):
        err = _rel(o[k], g[f"{ref}{nn}"])
        print(nn, k, err)
        assert err < 1e-12


def test_quadratic_solid_tables_bitwise():
    """The host-built shape tables of the HIP library (fs_solid2_table) equal the reference's
    shape_functions / shape_function_derivatives / integration_points bit for bit."""
    import ctypes
    from fem_shell_amd import _lib
    lib = _lib.load()
    for nn in (10, 13, 15, 20):
        g = golden("elem_pyr13" if nn == 13 else "solid2_tables")
        ng = ctypes.c_int32(0)
        pts, w, v = np.zeros((27, 3)), np.zeros(27), np.zeros(27 * nn * 4)
        assert lib.fs_solid2_table(nn, pts.ctypes.data, w.ctypes.data, v.ctypes.data, ctypes.byref(ng)) == 0
        n = ng.value
        assert np.array_equal(pts[:n], g[f"pts{nn}"]) and np.array_equal(w[:n], g[f"w{nn}"])
        assert np.array_equal(v[:n * nn * 4].reshape(n, nn, 4), g[f"vals{nn}"])
    assert lib.fs_solid2_table(8, None, None, None, None) != 0


def test_mitc3_nonlinear_vs_reference(oracle):
    """MITC3 / MITC3Composite TL tangent and internal force after update_configuration
    (elements/MITC3.py:1144-1832, MITC3_composite.py:496-621), incl. the T()-from-current-coords quirk."""
    g = golden("elem_nl3")
    props = golden_props(g)
    assert props["nonlinear"].min() == 0 and props["nonlinear"].max() == 1
    worst = [0.0, 0.0]
    for e in range(g["xyz"].shape[0]):
        KT, f, st = oracle.mitc3_NL(g["xyz"][e], props[e:e + 1], g["u"][e])
        assert st == 0
        ek = np.abs(KT - g["KT"][e]).max() / np.abs(g["KT"][e]).max()
        fs = np.abs(g["f_int"][e]).max()
        ef = np.abs(f - g["f_int"][e]).max() / fs if fs > 0 else np.abs(f).max()
        worst = [max(worst[0], ek), max(worst[1], ef)]
        assert ek <= 1e-12 and ef <= 1e-11, (e, ek, ef)
    print("oracle MITC3 KT / f_int vs reference", worst)


@pytest.mark.parametrize("nn", [4, 3])
def test_shell_geometric_stiffness_vs_reference(oracle, nn):
    """compute_geometric_stiffness(sigma) of MITC4 / MITC3 (+ composites): MITC4.py:1191-1225,
    MITC3.py:1501-1580."""
    g = golden("prestress")
    props = golden_props(g, f"prop{nn}_")
    for e in range(g[f"xyz{nn}"].shape[0]):
        KG = oracle.shell_KG(g[f"xyz{nn}"][e], props[e:e + 1], g[f"sigma{nn}"][e])
        ref = g[f"KG{nn}"][e]
        if np.abs(g[f"sigma{nn}"][e]).max() < 1e-20:      # skipped by the assembler -> zero block
            assert not KG.any()
            continue
        assert np.abs(KG - ref).max() <= 1e-13 * np.abs(ref).max(), e


def test_centrifugal_prestress_vs_reference(oracle):
    """compute_centrifugal_prestress: MITC3.py:1633-1715 and SOLID.py:268-320."""
    g = golden("prestress")
    props = golden_props(g, "prop3_")
    hit_zero = False
    for e in range(g["xyz3"].shape[0]):
        om, axis, cen = g["cfpar3"][e][0], g["cfpar3"][e][1:4], g["cfpar3"][e][4:7]
        s = oracle.centrifugal(g["xyz3"][e], float(props["rho"][e]), om, axis, cen, shell=True)
        ref = g["cf3"][e]
        hit_zero |= not ref.any()
        assert np.abs(s - ref).max() <= 1e-13 * max(np.abs(ref).max(), 1e-300), e
    assert hit_zero
    for nn in (8, 4, 6, 5):
        for e in range(g[f"sxyz{nn}"].shape[0]):
            par = g[f"scfpar{nn}"][e]
            s = oracle.centrifugal(g[f"sxyz{nn}"][e], float(g[f"srho{nn}"][e]), par[0], par[1:4], par[4:7], shell=False)
            ref = g[f"scf{nn}"][e]
            assert np.abs(s - ref).max() <= 1e-13 * max(np.abs(ref).max(), 1e-300), (nn, e)


def test_row_parallel_csr_summation_is_bit_identical(oracle):
    """bench.py's all-core CPU baseline sums rows in parallel (fso_csr_numeric_rows): same sums, same element order."""
    from fem_shell_amd import meshes
    import fem_shell_amd as fsa
    from tests import util
    mesh = meshes.jittered_plate(23, 17, n_sets=8)
    table = fsa.build_prop_table(util.standard_props())
    Ke, _ = oracle.mitc4_K_batch(mesh.coords[mesh.conn], mesh.set_id, table)
    dofs = (mesh.conn[:, :, None] * 6 + np.arange(6)).reshape(mesh.conn.shape[0], -1)
    ip, ix, kd = oracle.csr_assemble(dofs, mesh.node_count * 6, Ke)
    for nt in (1, 3):
        assert np.array_equal(oracle.csr_numeric_rows(dofs, mesh.node_count * 6, Ke, ip, ix, nthreads=nt).view(np.int64),
                              kd.view(np.int64))
