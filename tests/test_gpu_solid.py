"""GPU parity: linear solids (fs_solid_batch) vs the reference golden vectors
(tests/golden/elem_solid.npz) and vs the oracle on random batches."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
from tests.util_golden import golden

SOLID_REF = {
    8: np.array([[-1, -1, -1], [1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, -1, 1], [1, -1, 1], [1, 1, 1], [-1, 1, 1]], float),
    4: np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], float),
    6: np.array([[0, 0, -1], [1, 0, -1], [0, 1, -1], [0, 0, 1], [1, 0, 1], [0, 1, 1]], float),
    5: np.array([[-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0], [0, 0, 1]], float),
}


def _rel(a, b):
    scale = np.abs(b).reshape(b.shape[0], -1).max(axis=1)
    return (np.abs(a - b).reshape(b.shape[0], -1).max(axis=1) / scale).max()


def _run(nn, xyz, pid, props, **kw):
    from fem_shell_amd import batch
    dev = torch.device("cuda:0")
    sg = kw.pop("sigma", None)
    out = batch.solid_batch(nn, torch.as_tensor(xyz, device=dev), torch.as_tensor(pid, dtype=torch.int32, device=dev),
                            batch.to_device_solid_props(props), len(props), dense=True,
                            sigma=None if sg is None else torch.as_tensor(sg, device=dev), **kw)
    torch.cuda.synchronize()
    return {k: (v.cpu().numpy() if v is not None else None) for k, v in out.items()}


@pytest.mark.parametrize("nn", [8, 4, 6, 5])
def test_solids_vs_reference_goldens(nn):
    g = golden("elem_solid")
    xyz, props, red = g[f"xyz{nn}"], g[f"props{nn}"], g[f"reduced{nn}"]
    pid = np.arange(xyz.shape[0], dtype=np.int32)
    for r in (False, True):
        sel = np.flatnonzero(red == r)
        if not sel.size:
            continue
        o = _run(nn, xyz[sel], pid[sel], props, want_K=True, want_M=True, body=g["body"],
                 sigma=g[f"sigma{nn}"][sel], reduced=r)
        assert not o["status"].any()
        for k, ref in (("K", "K"), ("M", "M"), ("f", "F"), ("KG", "KG")):
            err = _rel(o[k], g[f"{ref}{nn}"][sel])
            print(nn, "reduced" if r else "full", k, err)
            assert err < 1e-12
        assert np.array_equal(o["K"], np.transpose(o["K"], (0, 2, 1)))


@pytest.mark.parametrize("nn", [8, 4, 6, 5])
def test_solids_vs_oracle_random(oracle, nn):
    import fem_shell_amd as fsa
    from fem_shell_amd.properties import solid_row
    from tests import util
    rng = np.random.default_rng(nn)
    E = 1003
    xyz = SOLID_REF[nn][None] + 0.12 * rng.standard_normal((E, nn, 3))
    xyz = xyz * rng.choice([1.0, 1e-2, 30.0], size=(E, 1, 1)) + rng.choice([0.0, 3.0], size=(E, 1, 1)) * rng.standard_normal((E, 1, 3))
    R = np.linalg.qr(rng.standard_normal((3, 3)))[0]
    props = np.stack([solid_row(fsa.IsotropicMaterial("s", 2.1e11, 0.3, 7800.0)),
                      solid_row(util.t300()), solid_row(util.t300(), R)])
    pid = rng.integers(0, 3, E).astype(np.int32)
    sg = rng.standard_normal((E, 6)) * 1e5
    body = np.array([0.0, 0.0, -9.81])
    ref = oracle.solid_batch(nn, xyz, pid, props, want_K=True, want_M=True, body=body, sigma=sg, nthreads=8)
    o = _run(nn, xyz, pid, props, want_K=True, want_M=True, body=body, sigma=sg)
    assert np.array_equal(o["status"], ref["status"])
    for k in ("K", "M", "f", "KG"):
        err = _rel(o[k], ref[k])
        print(nn, k, err)
        assert err < 1e-12


def test_solid_bad_jacobian_and_empty():
    import fem_shell_amd as fsa
    from fem_shell_amd.properties import solid_row
    props = solid_row(fsa.IsotropicMaterial("s", 2.1e11, 0.3, 7800.0))[None]
    x = np.stack([SOLID_REF[8], SOLID_REF[8][[1, 0, 3, 2, 5, 4, 7, 6]]])     # second one inverted
    o = _run(8, x, np.zeros(2, np.int32), props)
    assert o["status"][0] == 0 and o["status"][1] == 4
    o = _run(4, np.zeros((0, 4, 3)), np.zeros(0, np.int32), props)
    assert o["K"].shape == (0, 12, 12)
    with pytest.raises(NotImplementedError):
        _run(9, np.zeros((1, 9, 3)), np.zeros(1, np.int32), props)


SOLID2_REF = {
    10: np.array([(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (.5, 0, 0), (.5, .5, 0), (0, .5, 0), (0, 0, .5),
                  (0, .5, .5), (.5, 0, .5)], float),
    15: np.array([(0, 0, -1), (1, 0, -1), (0, 1, -1), (0, 0, 1), (1, 0, 1), (0, 1, 1), (.5, 0, -1), (0, .5, -1),
                  (0, 0, 0), (.5, .5, -1), (1, 0, 0), (0, 1, 0), (.5, 0, 1), (0, .5, 1), (.5, .5, 1)], float),
    20: np.array([(-1, -1, -1), (1, -1, -1), (1, 1, -1), (-1, 1, -1), (-1, -1, 1), (1, -1, 1), (1, 1, 1), (-1, 1, 1),
                  (0, -1, -1), (-1, 0, -1), (-1, -1, 0), (1, 0, -1), (1, -1, 0), (0, 1, -1), (1, 1, 0), (-1, 1, 0),
                  (0, -1, 1), (-1, 0, 1), (1, 0, 1), (0, 1, 1)], float),
    13: np.array([(-1, -1, 0), (1, -1, 0), (1, 1, 0), (-1, 1, 0), (0, 0, 1), (0, -1, 0), (-1, 0, 0), (-.5, -.5, .5),
                  (1, 0, 0), (.5, -.5, .5), (0, 1, 0), (.5, .5, .5), (-.5, .5, .5)], float),       # Gmsh type 19
}


@pytest.mark.parametrize("nn", [10, 13, 15, 20])
def test_quadratic_solids_vs_reference_goldens(nn):
    """TETRA10 / WEDGE15 / HEXA20 (tests/golden/elem_solid2.npz) incl. the HEXA20 finite-difference quirk, and
    PYRAMID13 (tests/golden/elem_pyr13.npz, elements/SOLID.py:1221-1537)."""
    g = golden("elem_pyr13" if nn == 13 else "elem_solid2")
    xyz, props = g[f"xyz{nn}"], g[f"props{nn}"]
    pid = np.arange(xyz.shape[0], dtype=np.int32)
    o = _run(nn, xyz, pid, props, want_K=True, want_M=True, body=g["body"], sigma=g[f"sigma{nn}"])
    assert not o["status"].any()
    for k, ref in (("K", "K"), ("M", "M"), ("f", "F"), ("KG", "KG")):
        err = _rel(o[k], g[f"{ref}{nn}"])
        print(nn, k, err)
        assert err < 1e-12
    assert np.array_equal(o["K"], np.transpose(o["K"], (0, 2, 1)))


@pytest.mark.parametrize("nn", [10, 13, 15, 20])
def test_quadratic_solids_vs_oracle_random(oracle, nn):
    import fem_shell_amd as fsa
    from fem_shell_amd.properties import solid_row
    from tests import util
    rng = np.random.default_rng(nn)
    E = 301
    xyz = SOLID2_REF[nn][None] + 0.04 * rng.standard_normal((E, nn, 3))
    xyz = xyz * rng.choice([1.0, 1e-2, 30.0], size=(E, 1, 1)) + rng.choice([0.0, 3.0], size=(E, 1, 1)) * rng.standard_normal((E, 1, 3))
    R = np.linalg.qr(rng.standard_normal((3, 3)))[0]
    props = np.stack([solid_row(fsa.IsotropicMaterial("s", 2.1e11, 0.3, 7800.0)),
                      solid_row(util.t300()), solid_row(util.t300(), R)])
    pid = rng.integers(0, 3, E).astype(np.int32)
    sg = rng.standard_normal((E, 6)) * 1e5
    body = np.array([0.0, 0.0, -9.81])
    ref = oracle.solid_batch(nn, xyz, pid, props, want_K=True, want_M=True, body=body, sigma=sg, nthreads=8)
    o = _run(nn, xyz, pid, props, want_K=True, want_M=True, body=body, sigma=sg)
    assert np.array_equal(o["status"], ref["status"])
    for k in ("K", "M", "f", "KG"):
        err = _rel(o[k], ref[k])
        print(nn, k, err)
        assert err < 1e-12


def test_quadratic_solid_assembly(oracle):
    """HEXA20 mesh through MeshAssembler: pattern from the block plan, values vs the oracle's element
    matrices summed in element order; element classes and factory."""
    import fem_shell_amd as fsa
    from fem_shell_amd import meshes
    from fem_shell_amd.assembler import MeshAssembler
    from fem_shell_amd.elements import HEXA20, ElementFactory, ElementFamily
    from fem_shell_amd.properties import solid_row
    rng = np.random.default_rng(5)
    # 3 x 2 x 2 bricks sharing corner and mid-edge nodes: build by deduplicating physical points
    pts, conn, key = [], [], {}
    for k in range(2):
        for j in range(2):
            for i in range(3):
                row = []
                for p in SOLID2_REF[20]:
                    q = tuple(np.round(p + 2 * np.array([i, j, k]), 6))
                    if q not in key:
                        key[q] = len(pts); pts.append(q)
                    row.append(key[q])
                conn.append(row)
    coords = np.array(pts, float) + 0.03 * rng.standard_normal((len(pts), 3))
    conn = np.array(conn, dtype=np.int64)
    mat = fsa.IsotropicMaterial("s", 2.1e11, 0.3, 7800.0)
    mesh = meshes.ArrayMesh(coords, conn)
    asm = MeshAssembler(mesh, {"elements": {"element_family": ElementFamily.SOLID, "material": mat}})
    K = asm.assemble_stiffness_matrix()
    ref = oracle.solid_batch(20, coords[conn], np.zeros(len(conn), np.int32), solid_row(mat)[None], want_K=True, want_M=True)
    dofs = (conn[:, :, None] * 3 + np.arange(3)).reshape(len(conn), -1)
    ip, ix, data = oracle.csr_assemble(dofs, len(pts) * 3, ref["K"])
    assert np.array_equal(K.indptr.cpu().numpy(), ip) and np.array_equal(K.indices.cpu().numpy(), ix)
    assert np.abs(K.data.cpu().numpy() - data).max() / np.abs(data).max() < 1e-12
    _, _, mdata = oracle.csr_assemble(dofs, len(pts) * 3, ref["M"])
    assert np.abs(asm.assemble_mass_matrix().data.cpu().numpy() - mdata).max() / np.abs(mdata).max() < 1e-12
    el = asm._element_map[4]
    assert isinstance(el, HEXA20) and el.K.shape == (60, 60)
    assert np.abs(el.K - ref["K"][4]).max() / np.abs(ref["K"][4]).max() < 1e-12
    view = mesh.elements[0]
    assert isinstance(ElementFactory.get_element(ElementFamily.SOLID, view, material=mat), HEXA20)
