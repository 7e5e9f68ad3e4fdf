"""GPU parity of the batched stress / strain recovery (csrc/femshell_stress.hip behind
fem_shell_amd.stress_recovery.StressRecovery) against the reference's UNMODIFIED StressRecovery driven by
its own MeshAssembler (tests/golden/stress_ref.npz, tools/gen_golden_asm.py): every public product
(element / nodal / area-weighted nodal, TOP / MIDDLE / BOTTOM, TOTAL / MEMBRANE / BENDING, strains,
derived invariants) on mixed MITC3/MITC4, MITC4, MITC3, mixed solids and HEXA8 meshes.
Tolerance: 1e-12 of the field's largest magnitude for the Voigt components and the algebraic invariants;
the Cardano principal stresses of solids go through acos and are checked to 1e-10 (conditioning of the
reference's own formula near repeated eigenvalues)."""
import numpy as np
import pytest

from fem_shell_amd.stress_recovery import (StressLocation, StressRecovery, StressType, compute_max_shear_stress,
                                           compute_principal_stresses, compute_von_mises)
from tests.test_gpu_asm_ref import build
from tests.util_golden import golden

pytestmark = pytest.mark.gpu
LOCS = (StressLocation.TOP, StressLocation.MIDDLE, StressLocation.BOTTOM)
TYPES = (StressType.TOTAL, StressType.MEMBRANE, StressType.BENDING)
CARDANO = ("sigma_1", "sigma_2", "sigma_3", "tau_max")


def _cmp(res, ref, tag, solid, worst):
    d = res.to_dict()
    keys = [k[len(tag):] for k in ref.files if k.startswith(tag)]
    assert sorted(keys) == sorted(d), (tag, sorted(keys), sorted(d))
    comp = [k for k in keys if k.startswith(("sigma_x", "sigma_y", "sigma_z", "tau_y", "tau_z", "epsilon_x", "epsilon_y",
                                             "epsilon_z", "gamma_x", "gamma_y", "gamma_z"))]
    scale = max(np.abs(ref[tag + k]).max() for k in comp) or 1.0
    for k in keys:
        a, b = d[k], ref[tag + k]
        assert a.shape == b.shape, (tag, k)
        if k == "principal_angle":          # degrees; atan2 of two rounded stresses
            err = np.abs(a - b).max() / 180.0
            # where sxx == syy to rounding the angle is ill-defined: compare through the stresses instead
            sx, sy = ref[tag + "sigma_xx"], ref[tag + "sigma_yy"]
            ok = np.abs(sx - sy) > 1e-6 * scale
            err = np.abs(a - b)[ok].max() / 180.0 if ok.any() else 0.0
            tol = 1e-9
        else:
            err = np.abs(a - b).max() / scale
            tol = 1e-10 if (solid and k in CARDANO) else 1e-12
        worst[k] = max(worst.get(k, 0.0), err)
        assert err < tol, (tag, k, err)


@pytest.mark.parametrize("case", ["smix", "quad", "tri", "somix", "hex", "q2", "p13"])
def test_stress_recovery_golden(case):
    g, ref = golden("asm_ref"), golden("stress_ref")
    if case == "p13":       # HEXA20 + PYRAMID13 (tools/gen_golden_pyr13.py)
        g, ref = golden("asm_p13"), golden("stress_p13")
    mesh, asm = build(case, g)
    pre = case + "_"
    solid = case in ("somix", "hex", "q2", "p13")
    sr = StressRecovery(asm, ref[pre + "u"])
    assert sr.n_nodes == mesh.node_count and sr.n_elements == mesh.conn.shape[0]
    worst = {}
    combos = [(l, t) for l in LOCS for t in TYPES] if not solid else [(StressLocation.MIDDLE, StressType.TOTAL)]
    for loc, typ in combos:
        tag = f"{pre}{loc.value}_{typ.value}_"
        _cmp(sr.compute_element_stresses(loc, typ), ref, tag + "elem_", solid, worst)
        _cmp(sr.compute_nodal_stresses(loc, typ), ref, tag + "nodal_", solid, worst)
        _cmp(sr.compute_nodal_stresses(loc, typ, smoothing="area_weighted"), ref, tag + "nodalw_", solid, worst)
    if not solid:
        gp = (0.3, -0.2) if case == "quad" else ((0.2, 0.5) if case == "tri" else
                                                 ((0.3, -0.2) if mesh.conn[0, 3] >= 0 else (0.2, 0.5)))
        _cmp(sr.compute_element_stresses(StressLocation.TOP, StressType.TOTAL, gauss_point=gp), ref,
             pre + "elem_offcentre_", solid, worst)
    for loc in (LOCS if not solid else LOCS[1:2]):
        tag = f"{pre}{loc.value}_"
        _cmp(sr.compute_element_strains(loc), ref, tag + "elem_strain_", solid, worst)
        _cmp(sr.compute_nodal_strains(loc), ref, tag + "nodal_strain_", solid, worst)
        _cmp(sr.compute_nodal_strains(loc, smoothing="area_weighted"), ref, tag + "nodalw_strain_", solid, worst)
    print(case, {k: float(f"{v:.1e}") for k, v in sorted(worst.items())})
    # all-layers convenience forms (stress_recovery.py:871-953)
    layers = sr.compute_nodal_stresses_all_layers_dict()
    mid = "middle_total_nodal_"
    assert np.array_equal(layers["MID_von_mises"], sr.compute_nodal_stresses().von_mises)
    assert np.abs(layers["MID_sigma_xx"] - ref[pre + mid + "sigma_xx"]).max() <= 1e-12 * np.abs(ref[pre + mid + "sigma_xx"]).max()
    assert set(k.split("_", 1)[0] for k in layers) == {"TOP", "MID", "BOT"}


def test_module_helpers():
    rng = np.random.default_rng(0)
    s = rng.standard_normal((6, 257)) * 1e6
    vm2 = compute_von_mises(s[0], s[1], s[3])
    assert np.allclose(vm2, np.sqrt(s[0]**2 + s[1]**2 - s[0] * s[1] + 3 * s[3]**2), rtol=1e-15)
    s1, s2, s3 = compute_principal_stresses(s[0], s[1], s[3], s[2], s[4], s[5])
    # the reference's Cardano form takes cos(3 phi) = -q / (2 (p/3)^1.5) (stress_recovery.py:1216-1230): its
    # roots are mean - (deviatoric eigenvalue), i.e. the true principal stresses mirrored about the mean.
    # Restated here with LAPACK eigenvalues; the golden fixtures above pin the same behaviour.
    ev = np.array([np.linalg.eigvalsh(np.array([[s[0, i], s[3, i], s[5, i]], [s[3, i], s[1, i], s[4, i]],
                                                [s[5, i], s[4, i], s[2, i]]])) for i in range(s.shape[1])])
    mean = ev.mean(axis=1, keepdims=True)
    mir = np.sort(2 * mean - ev, axis=1)
    sc = np.abs(ev).max()
    assert np.abs(s1 - mir[:, 2]).max() / sc < 1e-9 and np.abs(s2 - mir[:, 1]).max() / sc < 1e-9
    assert np.abs(s3 - mir[:, 0]).max() / sc < 1e-9
    assert np.abs(compute_max_shear_stress(s[0], s[1], s[3], s[2], s[4], s[5]) - (mir[:, 2] - mir[:, 0]) / 2).max() / sc < 1e-9
    a, b, th = compute_principal_stresses(s[0], s[1], s[3])
    assert np.allclose(a + b, s[0] + s[1], rtol=1e-12)


def test_stress_vs_oracle_larger_mesh():
    """A 40 x 23 jittered laminate plate (8 property sets) and a 12 x 6 x 4 HEXA8 block: GPU StressRecovery vs the
    CPU restatement oracle/stress.py (itself pinned to the reference fixtures in tests/test_oracle_stress.py)."""
    import fem_shell_amd as fsa
    from fem_shell_amd import meshes
    from fem_shell_amd.assembler import MeshAssembler
    from fem_shell_amd.elements import ElementFamily
    from fem_shell_amd.properties import solid_C
    from oracle import stress as S
    from tests import util
    rng = np.random.default_rng(9)
    mesh = meshes.jittered_plate(40, 23, n_sets=8)
    props = util.standard_props()
    asm = MeshAssembler(mesh, {"elements": {"element_family": ElementFamily.SHELL,
                                            "properties": {f"band_{i}": props[i] for i in range(8)}}})
    u = rng.standard_normal(asm.dofs_count) * 1e-6
    sr = StressRecovery(asm, u)
    data = [(asm._prop_table["A"][p].reshape(3, 3), float(asm._prop_table["h"][p])) for p in asm._prop_id]
    for nodal, weighted in ((False, False), (True, False), (True, True)):
        ref = S.recover(mesh.coords, mesh.conn, u, "shell", data, 1, 3, False, nodal, weighted)
        got = (sr.compute_nodal_stresses(StressLocation.TOP, StressType.TOTAL, "area_weighted" if weighted else "average")
               if nodal else sr.compute_element_stresses(StressLocation.TOP, StressType.TOTAL))
        scale = np.abs(ref).max()
        assert np.abs(got.sigma_xx - ref[:, 0]).max() < 1e-12 * scale and np.abs(got.sigma_yy - ref[:, 1]).max() < 1e-12 * scale
        assert np.abs(got.von_mises - S.invariants(ref, False)["von_mises"]).max() < 1e-12 * scale
    # solids
    nx, ny, nz = 12, 6, 4
    ids = np.arange((nx + 1) * (ny + 1) * (nz + 1)).reshape(nz + 1, ny + 1, nx + 1)
    Z, Y, X = np.meshgrid(np.arange(nz + 1.0), np.arange(ny + 1.0), np.arange(nx + 1.0), indexing="ij")
    coords = np.stack([X, Y, Z], -1).reshape(-1, 3) + 0.05 * rng.standard_normal((ids.size, 3))
    n = lambda dz, dy, dx: ids[dz:nz + dz, dy:ny + dy, dx:nx + dx]
    conn = np.stack([n(0, 0, 0), n(0, 0, 1), n(0, 1, 1), n(0, 1, 0), n(1, 0, 0), n(1, 0, 1), n(1, 1, 1), n(1, 1, 0)], -1).reshape(-1, 8)
    mat = util.t300()
    asm = MeshAssembler(meshes.ArrayMesh(coords, conn), {"elements": {"element_family": ElementFamily.SOLID, "material": mat}})
    u = rng.standard_normal(asm.dofs_count) * 1e-4
    sr = StressRecovery(asm, u)
    C = [solid_C(mat, None)] * conn.shape[0]
    for nodal, weighted in ((False, False), (True, True)):
        ref = S.recover(coords, conn, u, "solid", C, 0, 3, False, nodal, weighted)
        got = sr.compute_nodal_stresses(smoothing="area_weighted") if nodal else sr.compute_element_stresses()
        scale = np.abs(ref).max()
        for k, col in (("sigma_xx", 0), ("sigma_yy", 1), ("sigma_zz", 2), ("sigma_xy", 3), ("tau_yz", 4), ("tau_zx", 5)):
            assert np.abs(getattr(got, k) - ref[:, col]).max() < 1e-12 * scale, k
        eps = S.recover(coords, conn, u, "solid", C, 0, 3, True, nodal, weighted)
        ge = sr.compute_nodal_strains(smoothing="area_weighted") if nodal else sr.compute_element_strains()
        assert np.abs(ge.gamma_yz - eps[:, 4]).max() < 1e-12 * np.abs(eps).max()
