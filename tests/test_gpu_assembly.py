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
    # the cached-blocks route (element kernel ran earlier, assembly kernel alone) gives the same bits
    c = asm.assemble_stiffness_matrix().data
    assert bool((a == c).all())


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


def test_newton_reassembly_matches_oracle_loop(oracle):
    """config 4: K_T(u) and f_int(u) re-assembled from u -- oracle = the Python loop 'for each
    element: update_configuration(u[dofs]); compute_tangent_stiffness(); compute_internal_forces()'."""
    from fem_shell_amd.assembler import MeshAssembler
    mesh = meshes.flat_patch(0.0, 10.0, 0.0, 1.0, 40, 4)
    mesh.coords[:, 2] = 0.02 * np.sin(mesh.coords[:, 0])          # slightly curved strip
    mat = fsa.IsotropicMaterial("cant", 1.2e6, 0.0, 1.0)
    model = {"elements": {"element_family": ElementFamily.SHELL, "material": mat, "thickness": 0.1,
                          "nonlinear": True}}
    asm = MeshAssembler(mesh, model)
    assert asm._prop_table["nonlinear"][0] == 1
    rng = np.random.default_rng(2)
    xs = mesh.coords[:, 0]
    u = np.zeros((mesh.node_count, 6))
    u[:, 2] = 1e-2 * xs**2
    u[:, 0] = -1e-3 * xs
    u[:, 4] = -2e-2 * xs
    u += 1e-5 * rng.standard_normal(u.shape)
    u = u.reshape(-1)
    KT, f = asm.assemble_tangent_and_forces(u)
    dofs = asm._dofs_array
    E = mesh.conn.shape[0]
    kt_e = np.empty((E, 24, 24)); f_e = np.empty((E, 24))
    for e in range(E):
        kt_e[e], f_e[e], _ = oracle.mitc4_NL(mesh.coords[mesh.conn[e]], asm._prop_table[:1], u[dofs[e]])
    ip, ix, data = oracle.csr_assemble(dofs, mesh.node_count * 6, kt_e)
    assert np.array_equal(KT.indptr.cpu().numpy(), ip) and np.array_equal(KT.indices.cpu().numpy(), ix)
    errK = np.abs(KT.data.cpu().numpy() - data).max() / np.abs(data).max()
    fo = np.zeros(mesh.node_count * 6)
    for e in range(E):
        fo[dofs[e]] += f_e[e]
    errF = np.abs(f.cpu().numpy() - fo).max() / np.abs(fo).max()
    print("newton reassembly: KT", errK, "f_int", errF)
    assert errK < 1e-12 and errF < 1e-12
    assert np.abs(asm.assemble_internal_forces(np.zeros_like(u)).cpu().numpy()).max() == 0.0
    with pytest.raises(ValueError):
        asm.assemble_internal_forces(np.zeros(5))
    # element-class route
    from fem_shell_amd.elements import MITC4
    el = MITC4(mesh.coords[mesh.conn[3]], tuple(mesh.conn[3]), mat, 0.1, nonlinear=True)
    el.update_configuration(u[dofs[3]])
    assert np.abs(el.compute_tangent_stiffness() - kt_e[3]).max() < 1e-12 * np.abs(kt_e[3]).max()
    assert np.abs(el.compute_internal_forces() - f_e[3]).max() < 1e-12 * np.abs(f_e[3]).max()
    with pytest.raises(ValueError):
        el.update_configuration(np.zeros(3))


def _oracle_mixed(oracle, mesh, table, pid, which):
    dofs_list, mats = [], []
    for e in range(mesh.conn.shape[0]):
        row = mesh.conn[e][mesh.conn[e] >= 0]
        x = mesh.coords[row][None]
        if len(row) == 3:
            K, M, st = oracle.mitc3_batch(x, pid[e:e + 1], table, want_K=True, want_M=True)
        else:
            K, st = oracle.mitc4_K_batch(x, pid[e:e + 1], table)
            M, _ = oracle.mitc4_M_batch(x, pid[e:e + 1], table)
        assert not np.any(st)
        dofs_list.append((row[:, None] * 6 + np.arange(6)).reshape(-1))
        mats.append((K if which == "K" else M)[0])
    return oracle.csr_assemble_mixed(dofs_list, mesh.node_count * 6, mats), mats


@pytest.mark.parametrize("case", ["tri_cylinder", "mixed_plate"])
def test_assembled_tri_and_mixed(case, oracle):
    """MITC3+ meshes (test_ko2017_performance.py:882-892 triangulation) and mixed MITC3+/MITC4+
    meshes (the reference's `_is_mixed_mesh` list branch, core/assembler.py:201-218,287-292)."""
    from fem_shell_amd.assembler import MeshAssembler
    props = util.standard_props()
    if case == "tri_cylinder":
        mesh = meshes.cylindrical_patch(25.0, 25.0, 40.0, 12, 10, triangular=True)
        model = {"elements": {"element_family": ElementFamily.SHELL,
                              "material": fsa.IsotropicMaterial("sc", 4.32e8, 0.0, 360.0), "thickness": 0.25}}
    else:
        base = meshes.jittered_plate(9, 7, n_sets=4)
        conn = meshes.mixed_conn(9, 7)
        cell_of = np.repeat(np.arange(63), np.where(((np.arange(63) // 9 + np.arange(63) % 9) % 2) == 1, 2, 1))
        mesh = meshes.ArrayMesh(base.coords, conn, base.set_id[cell_of], base.set_names)
        model = {"elements": {"element_family": ElementFamily.SHELL,
                              "properties": {f"band_{i}": props[3 + i] for i in range(4)}}}
    asm = MeshAssembler(mesh, model)
    assert asm._is_mixed_mesh == (case == "mixed_plate")
    for which in ("K", "M"):
        A = asm.assemble_stiffness_matrix() if which == "K" else asm.assemble_mass_matrix()
        (ip, ix, data), mats = _oracle_mixed(oracle, mesh, asm._prop_table, asm._prop_id, which)
        assert np.array_equal(A.indptr.cpu().numpy(), ip) and np.array_equal(A.indices.cpu().numpy(), ix)
        err = np.abs(A.data.cpu().numpy() - data).max() / np.abs(data).max()
        print(case, which, "nnz", A.nnz, "rel err", err)
        assert err < 1e-12
    # reference-shaped views
    if case == "mixed_plate":
        assert asm._ke_array is None and asm._dofs_array is None
        kel, dl = asm._ke_list, asm._dofs_list
        assert len(kel) == mesh.conn.shape[0] and kel[1].shape == (18, 18) and kel[0].shape == (24, 24)
        (_, _, _), Kmats = _oracle_mixed(oracle, mesh, asm._prop_table, asm._prop_id, "K")
        for e in (0, 1, 2, 5):
            assert np.abs(kel[e] - Kmats[e]).max() < 1e-12 * np.abs(Kmats[e]).max()
            assert len(dl[e]) == kel[e].shape[0]
        el = asm._element_map[1]
        assert type(el).__name__ == "MITC3Composite" and el.dofs_count == 18
        assert np.abs(el.K - Kmats[1]).max() < 1e-12 * np.abs(Kmats[1]).max()
    else:
        assert asm._ke_array.shape == (240, 18, 18) and asm._dofs_array.shape == (240, 18)


def test_mitc3_element_classes(oracle):
    from fem_shell_amd.elements import MITC3, MITC3Composite, ElementFactory
    xyz = np.array([[0, 0, 0], [1, 0, .05], [.3, 1, .1]], float)
    mat = fsa.IsotropicMaterial("steel", 2.1e11, 0.3, 7800.0)
    el = MITC3(xyz, (0, 1, 2), mat, 0.01)
    table = fsa.build_prop_table([fsa.ShellProperty(mat, 0.01)])
    Ko, Mo, _ = oracle.mitc3_batch(xyz[None], [0], table, want_K=True, want_M=True)
    assert np.abs(el.K - Ko[0]).max() / np.abs(Ko).max() < 1e-12
    assert np.abs(el.M - Mo[0]).max() / np.abs(Mo).max() < 1e-12
    assert el.K.shape == (18, 18) and el.dofs_count == 18 and abs(el.area() - 0.5 * np.linalg.norm(np.cross(xyz[1] - xyz[0], xyz[2] - xyz[0]))) < 1e-15
    lam = fsa.create_laminate_from_angles(util.t300(), 0.125e-3, [0, 45, 90])
    ec = MITC3Composite(xyz, (0, 1, 2), lam)
    Kc, _, _ = oracle.mitc3_batch(xyz[None], [0], fsa.build_prop_table([fsa.CompositeShellProperty(lam)]))
    assert ec.has_coupling and np.abs(ec.K - Kc[0]).max() / np.abs(Kc).max() < 1e-12
    with pytest.raises(ValueError):
        MITC3(np.array([[0, 0, 0], [1, 0, 0], [2, 0, 0]], float), (0, 1, 2), mat, 0.01)
    view = meshes.ArrayMesh(xyz, np.array([[0, 1, 2]])).elements[0]
    made = ElementFactory.get_element(ElementFamily.SHELL, view, shell_property=fsa.CompositeShellProperty(lam))
    assert type(made).__name__ == "MITC3Composite"


def _hex_block(nx, ny, nz, jitter=0.08, seed=4):
    ids = np.arange((nx + 1) * (ny + 1) * (nz + 1)).reshape(nz + 1, ny + 1, nx + 1)
    Z, Y, X = np.meshgrid(np.arange(nz + 1.0), np.arange(ny + 1.0), np.arange(nx + 1.0), indexing="ij")
    coords = np.stack([X, Y, Z], -1).reshape(-1, 3)
    coords += jitter * np.random.default_rng(seed).uniform(-1, 1, coords.shape)
    n = lambda dz, dy, dx: ids[dz:nz + dz, dy:ny + dy, dx:nx + dx]
    hexa = np.stack([n(0, 0, 0), n(0, 0, 1), n(0, 1, 1), n(0, 1, 0), n(1, 0, 0), n(1, 0, 1), n(1, 1, 1), n(1, 1, 0)], -1)
    return coords, hexa.reshape(-1, 8)


@pytest.mark.parametrize("case", ["hexa", "mixed_solid"])
def test_assembled_solids(case, oracle):
    """SOLID family: K, M, body-load vector and geometric stiffness; 'mixed_solid' splits some
    bricks into 2 wedges or 6 tets... (here: wedges + pyramids/tets around them) so that all four
    linear solids meet in one mesh (tests/test_mixed_solid_elements.py of the reference)."""
    from fem_shell_amd.assembler import MeshAssembler
    from fem_shell_amd.properties import solid_row
    coords, hexa = _hex_block(4, 3, 3)
    if case == "hexa":
        conn = hexa
        mat = fsa.IsotropicMaterial("steel", 2.1e11, 0.3, 7800.0)
        model = {"elements": {"element_family": ElementFamily.SOLID, "material": mat}}
    else:
        rows = []
        for k, h in enumerate(hexa):
            if k % 3 == 0:
                rows.append(list(h))
            elif k % 3 == 1:      # two wedges: triangles (0,1,2)/(4,5,6) and (0,2,3)/(4,6,7)
                rows.append([h[0], h[1], h[2], h[4], h[5], h[6], -1, -1])
                rows.append([h[0], h[2], h[3], h[4], h[6], h[7], -1, -1])
            else:                 # pyramid on the bottom face with apex 6 + two tets filling the rest... keep it
                rows.append([h[0], h[1], h[2], h[3], h[6], -1, -1, -1])      # simple: base 0123, apex 6
                rows.append([h[0], h[1], h[6], h[5], -1, -1, -1, -1])        # tets (orientation checked below)
                rows.append([h[0], h[5], h[6], h[4], -1, -1, -1, -1])
        conn = np.array(rows, dtype=np.int64)
        R = np.linalg.qr(np.random.default_rng(1).standard_normal((3, 3)))[0]
        mat = util.t300()
        model = {"elements": {"element_family": ElementFamily.SOLID, "material": mat, "orientation": R}}
    mesh = meshes.ArrayMesh(coords, conn)
    asm = MeshAssembler(mesh, model)
    assert asm.dofs_per_node == 3 and asm.dofs_count == 3 * mesh.node_count
    props = solid_row(mat, model["elements"].get("orientation"))[None]
    body = np.array([0.3, -1.0, -9.81])
    sig = np.random.default_rng(3).standard_normal((conn.shape[0], 6)) * 1e5
    sig[2] = 0.0
    dofs_list, mats = [], {"K": [], "M": [], "KG": []}
    f_ref = np.zeros(asm.dofs_count)
    for e in range(conn.shape[0]):
        row = conn[e][conn[e] >= 0]
        o = oracle.solid_batch(len(row), coords[row][None], [0], props, want_K=True, want_M=True, body=body,
                               sigma=sig[e:e + 1])
        assert o["status"][0] == 0, (e, len(row))
        d = (row[:, None] * 3 + np.arange(3)).reshape(-1)
        dofs_list.append(d)
        for k in mats:
            mats[k].append(o[k][0])
        f_ref[d] += o["f"][0]
    got = {"K": asm.assemble_stiffness_matrix(), "M": asm.assemble_mass_matrix(),
           "KG": asm.assemble_geometric_stiffness(stress_field={int(e): sig[e] for e in range(conn.shape[0])})}
    for k, A in got.items():
        ip, ix, data = oracle.csr_assemble_mixed(dofs_list, asm.dofs_count, mats[k])
        assert np.array_equal(A.indptr.cpu().numpy(), ip) and np.array_equal(A.indices.cpu().numpy(), ix)
        err = np.abs(A.data.cpu().numpy() - data).max() / np.abs(data).max()
        print(case, k, "nnz", A.nnz, "rel err", err)
        assert err < 1e-12
    f = asm.assemble_load_vector(type("LC", (), {"value": body})()).cpu().numpy()
    assert np.abs(f - f_ref).max() < 1e-12 * np.abs(f_ref).max()
    el = asm._element_map[1]
    assert np.abs(el.K - mats["K"][1]).max() < 1e-12 * np.abs(mats["K"][1]).max()
    assert np.abs(el.body_load(body) - oracle.solid_batch(el.node_count, coords[conn[1][conn[1] >= 0]][None], [0], props,
                                                          want_K=False, body=body)["f"][0]).max() < 1e-9
    with pytest.raises(ValueError):
        asm.assemble_geometric_stiffness()


def test_reference_mesh_protocol_duck_typed():
    """The mesh protocol the reference assembler touches (core/assembler.py:72,107,113-119,160-187,220): a mesh of
    plain objects with arbitrary node ids / element ids, `node_id_to_index`, `element_sets[name].elements`, one
    element the SHELL factory does not know (5 nodes: skipped like the reference) and the deprecated per-element
    thickness -- against the same mesh given as ArrayMesh."""
    import warnings
    from types import SimpleNamespace
    from fem_shell_amd.assembler import MeshAssembler
    base = meshes.cylindrical_patch(25.0, 25.0, 40.0, 5, 4)
    coords, conn = base.coords, meshes.mixed_conn(5, 4)
    nid = 100 + 3 * np.arange(coords.shape[0])                       # node ids != node indices
    perm = np.random.default_rng(1).permutation(coords.shape[0])     # index of node id k is perm-defined
    index_of = {int(nid[k]): int(perm[k]) for k in range(coords.shape[0])}
    coords_by_index = np.empty_like(coords)
    coords_by_index[perm] = coords
    elements = []
    for e, row in enumerate(conn):
        nodes = [int(k) for k in row if k >= 0]
        elements.append(SimpleNamespace(id=1000 + 7 * e, node_ids=tuple(int(nid[k]) for k in nodes), node_count=len(nodes),
                                        node_coords=coords[nodes], thickness=None))
    odd = SimpleNamespace(id=5, node_ids=tuple(int(nid[k]) for k in range(5)), node_count=5, node_coords=coords[:5],
                          thickness=None)                             # no 5-node shell: ElementFactory -> False
    elements.insert(3, odd)
    sets = {"a": SimpleNamespace(elements=elements[:10]), "b": SimpleNamespace(elements=elements[10:])}
    mesh = SimpleNamespace(elements=elements, node_id_to_index=index_of, node_count=coords.shape[0], element_sets=sets)
    props = util.standard_props()
    model = {"elements": {"element_family": ElementFamily.SHELL, "properties": {"a": props[0], "b": props[4]}}}
    asm = MeshAssembler(mesh, model)
    # the same mesh as arrays: node index = perm, elements without the odd one
    conn_idx = np.where(conn >= 0, perm[np.maximum(conn, 0)], -1)
    set_id = np.array([0 if k < 9 else 1 for k in range(conn.shape[0])], dtype=np.int32)   # "a" lost the odd element
    ref = MeshAssembler(meshes.ArrayMesh(coords_by_index, conn_idx, set_id, ["a", "b"]), model)
    K, Kr = asm.assemble_stiffness_matrix(), ref.assemble_stiffness_matrix()
    assert np.array_equal(K.indptr.cpu().numpy(), Kr.indptr.cpu().numpy())
    assert np.array_equal(K.indices.cpu().numpy(), Kr.indices.cpu().numpy())
    assert bool((K.data == Kr.data).all()) and bool((asm.assemble_mass_matrix().data == ref.assemble_mass_matrix().data).all())
    assert len(asm._element_map) == conn.shape[0] and 5 not in asm._element_map       # odd element skipped
    assert sorted(asm._element_map.keys())[0] == 1000 and asm._is_mixed_mesh
    el = asm._element_map[1000 + 7 * 4]
    assert el.node_ids == list(elements[5].node_ids) or tuple(el.node_ids) == elements[5].node_ids
    nd = asm._node_dofs_map
    k = 17
    assert nd[int(nid[k])] == tuple(range(int(perm[k]) * 6, int(perm[k]) * 6 + 6))     # keyed by node ID
    assert asm.dofs_count == coords.shape[0] * 6
    # deprecated per-element thickness (core/assembler.py:144-158)
    for el_ in elements:
        el_.thickness = 0.02
    mat = fsa.IsotropicMaterial("m", 2.1e11, 0.3, 7800.0)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        asm2 = MeshAssembler(mesh, {"elements": {"element_family": ElementFamily.SHELL, "material": mat, "thickness": 0.5}})
    assert any(issubclass(x.category, DeprecationWarning) for x in w)
    assert float(asm2._prop_table["h"][0]) == 0.02


@pytest.mark.parametrize("kind", ["tri", "mixed"])
def test_newton_reassembly_cached_linear_blocks(kind):
    """K_T / f_int re-assembly with the cached linear blocks (default) against recomputing them every call: bitwise
    equal on a MITC3+ mesh and on a mixed MITC3+/MITC4+ mesh (quads in the element-major layout recompute, triangles
    read the cache)."""
    from fem_shell_amd.assembler import MeshAssembler
    mat = fsa.IsotropicMaterial("cant", 1.2e6, 0.3, 1.0)
    if kind == "tri":
        mesh = meshes.cylindrical_patch(5.0, 4.0, 60.0, 9, 7, triangular=True)
    else:
        base = meshes.cylindrical_patch(5.0, 4.0, 60.0, 9, 7)
        mesh = meshes.ArrayMesh(base.coords, meshes.mixed_conn(9, 7))
    asm = MeshAssembler(mesh, {"elements": {"element_family": ElementFamily.SHELL, "material": mat, "thickness": 0.05,
                                            "nonlinear": True}})
    rng = np.random.default_rng(9)
    u = 1e-3 * rng.standard_normal(asm.dofs_count)
    Kc, fc = asm.assemble_tangent_and_forces(u)
    f_only = asm.assemble_internal_forces(u)
    asm.reuse_linear_blocks = False
    Kr, fr = asm.assemble_tangent_and_forces(u)
    assert bool((Kc.data == Kr.data).all()) and bool((fc == fr).all()) and bool((f_only == fr).all())
    assert float(Kr.data.abs().max()) > 0.0 and float(fr.abs().max()) > 0.0
