#!/usr/bin/env python3
"""Assembly + linear-static goldens of SURVEY.md 8(c)-6 from the reference's UNMODIFIED MeshAssembler
(core/assembler.py:12-336, run through tools/ref_shim.py's recording PETSc stand-in), with the boundary
conditions, loads and measurement points of the reference's own Ko-Lee-Bathe benchmark cases
(tests/benchmarks/test_ko2017_performance.py).  Build container only.

    python tools/gen_golden_asm_ko.py     # writes tests/golden/asm_scordelis_32.npz, asm_hemisphere_16.npz,
                                          # asm_twisted_16x96.npz, asm_scordelis_16_floor.npz

Each fixture is DATA: mesh arrays, material numbers, what the reference assembler returns for them (dofs_count,
_row_nnz, CSR of K), the load vector and fixed dofs of the benchmark case, u = spsolve of the reduced system, the
normalised displacement -- and the reference's OWN sensitivity to its inputs: every nodal coordinate is moved by
+-1 ulp (seeded signs, 3 draws), the unmodified reference re-assembles and re-solves, and
  K_floor = max |K' - K| / max |K|,   u_floor = max |u' - u| / max |u|
are stored.  A differently rounded but equally valid implementation cannot be expected to agree with the
reference better than the reference agrees with itself under a 1-ulp input change; the parity tests assert
err <= max(1e-12, 10 * floor) and print both numbers.

  asm_scordelis_32   config C1 itself: Scordelis-Lo roof 32 x 32 (:857-893 mesh, :1031-1106 BCs / gravity load,
                     :1109-1185 case; R = L/2 = 25, 40 degrees, t = 0.25, E = 4.32e8, nu = 0, rho = 360)
  asm_hemisphere_16  hemisphere with 18 degree cut-out (:1558-1604 mesh, :1639-1657 BCs, :1731-1769 load / measure;
                     R = 10, t = 0.04, E = 6.825e7, nu = 0.3, P = 2)
  asm_twisted_16x96  MacNeal-Harder twisted beam (:1195-1229 mesh, clamped root :497-516, tip load :1360-1368;
                     L = 12, b = 1.1, 90 degrees, t = 0.32, E = 29e6, nu = 0.22, in-plane load P = 1 on dof 2)
  asm_scordelis_16_floor   the two floors for the existing asm_scordelis_16.npz mesh (tools/gen_golden.py)
"""
import contextlib
import io
import os
import sys

import numpy as np
from scipy.sparse import csr_matrix
from scipy.sparse.linalg import spsolve

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import gen_golden_asm as G  # noqa: E402  (installs the shim; RefMesh)
from fem_shell.core.assembler import MeshAssembler  # noqa: E402
from fem_shell.core.material import IsotropicMaterial  # noqa: E402
from fem_shell.elements import ElementFamily  # noqa: E402

from fem_shell_amd import meshes  # noqa: E402  (array mesh generators only; no kernels involved)

OUT = os.path.join(ROOT, "tests", "golden")


def ref_assemble(coords, conn, mat, t):
    """The reference's unmodified assembler on array data -> (asm, scipy CSR of K)."""
    with contextlib.redirect_stdout(io.StringIO()):      # the reference prints progress lines
        asm = MeshAssembler(G.RefMesh(coords, conn),
                            {"elements": {"element_family": ElementFamily.SHELL, "material": mat, "thickness": t}})
        K = asm.assemble_stiffness_matrix()
    n = asm.dofs_count
    return asm, csr_matrix((K.data, K.indices, K.indptr), shape=(n, n))


def solve(K, F, fixed):
    """_solve of the reference benchmarks (test_ko2017_performance.py:401-408)."""
    ndof = F.shape[0]
    free = np.setdiff1d(np.arange(ndof), fixed)
    u = np.zeros(ndof)
    u[free] = spsolve(K[free, :][:, free].tocsc(), F[free])
    return u


def ulp_perturbed(coords, seed):
    rng = np.random.default_rng(seed)
    up = rng.random(coords.shape) < 0.5
    return np.where(up, np.nextafter(coords, np.inf), np.nextafter(coords, -np.inf))


def floors(coords, conn, mat, t, K, u, F, fixed, seeds=(101, 102, 103)):
    kf = uf = 0.0
    for s in seeds:
        _, Kp = ref_assemble(ulp_perturbed(coords, s), conn, mat, t)
        assert np.array_equal(Kp.indices, K.indices)
        kf = max(kf, np.abs(Kp.data - K.data).max() / np.abs(K.data).max())
        uf = max(uf, np.abs(solve(Kp, F, fixed) - u).max() / np.abs(u).max())
    return kf, uf


def save(name, coords, conn, mat, t, F, fixed, node_info, wref, measure):
    asm, K = ref_assemble(coords, conn, mat, t)
    u = solve(K, F, fixed)
    kf, uf = floors(coords, conn, mat, t, K, u, F, fixed)
    norm = abs(measure(u)) / wref
    np.savez_compressed(os.path.join(OUT, name + ".npz"), coords=coords, conn=conn,
                        dofs_count=asm.dofs_count, row_nnz=np.asarray(asm._row_nnz), dofs_array=asm._dofs_array,
                        K_indptr=K.indptr.astype(np.int64), K_indices=K.indices.astype(np.int32), K_data=K.data,
                        F=F, fixed=fixed, u=u, norm_disp=norm, wref=wref, K_floor=kf, u_floor=uf,
                        E=mat.E, nu=mat.nu, rho=mat.rho, t=t, **node_info)
    print(f"{name}: nnz {K.nnz}, normalised displacement {norm:.4f}, K_floor {kf:.2e}, u_floor {uf:.2e}, "
          f"{os.path.getsize(os.path.join(OUT, name + '.npz')) // 1024} KiB")


def scordelis_case(n):
    R, L, ang, t = 25.0, 25.0, 40.0, 0.25
    mat = IsotropicMaterial("Ko2017_Scordelis", 4.32e8, 0.0, 360.0)
    mesh = meshes.cylindrical_patch(R, L, ang, n, n)
    coords, conn = mesh.coords, mesh.conn
    ndof = 6 * coords.shape[0]
    F = np.zeros(ndof)                                   # gravity in -x lumped to the element nodes (:1061-1106)
    for e in range(conn.shape[0]):
        c = coords[conn[e]]
        area = 0.5 * np.linalg.norm(np.cross(c[1] - c[0], c[2] - c[0])) + \
            0.5 * np.linalg.norm(np.cross(c[2] - c[0], c[3] - c[0]))
        for nd in conn[e]:
            F[nd * 6 + 0] -= area * t * 360.0 * 1.0 / 4
    fixed = []
    for i, (x, y, z) in enumerate(coords):               # (:1031-1058)
        if abs(z) < 1e-6:
            fixed += [i * 6 + 2, i * 6 + 3, i * 6 + 4]
        if abs(z - L) < 1e-6:
            fixed += [i * 6 + 0, i * 6 + 1, i * 6 + 5]
        if abs(y) < 1e-6 and x > 0.0:
            fixed += [i * 6 + 1, i * 6 + 3, i * 6 + 5]
    fixed = np.array(sorted(set(fixed)), dtype=np.int64)
    target = np.array([R * np.cos(np.radians(ang)), R * np.sin(np.radians(ang)), 0.0])
    node = int(np.argmin(np.linalg.norm(coords - target, axis=1)))
    return coords, conn, mat, t, F, fixed, node


def gen_scordelis_32():
    coords, conn, mat, t, F, fixed, node = scordelis_case(32)
    save("asm_scordelis_32", coords, conn, mat, t, F, fixed, dict(node=node), 3.0240e-1, lambda u: u[node * 6])


def gen_scordelis_16_floor():
    coords, conn, mat, t, F, fixed, node = scordelis_case(16)
    g = np.load(os.path.join(OUT, "asm_scordelis_16.npz"))
    assert np.array_equal(g["coords"], coords) and np.array_equal(g["fixed"], fixed) and np.array_equal(g["F"], F)
    _, K = ref_assemble(coords, conn, mat, t)
    u = solve(K, F, fixed)
    kf, uf = floors(coords, conn, mat, t, K, u, F, fixed)
    # the assembler route and the SciPy COO route of the existing fixture hold the same matrix
    same = np.abs(K.data - g["K_data"]).max() / np.abs(K.data).max()
    np.savez_compressed(os.path.join(OUT, "asm_scordelis_16_floor.npz"), K_floor=kf, u_floor=uf,
                        assembler_vs_coo_route=same, u_assembler_route=u)
    print(f"asm_scordelis_16_floor: K_floor {kf:.2e}, u_floor {uf:.2e}, assembler vs COO route {same:.2e}")


def gen_hemisphere_16():
    n, R, t, P = 16, 10.0, 0.04, 2.0
    mat = IsotropicMaterial("Ko2017_Sphere", 6.825e7, 0.3, 1.0)
    mesh = meshes.sphere_patch(R, 18.0, 90.0, 90.0, n)
    coords, conn = mesh.coords, mesh.conn
    fixed = []
    for i, (x, y, z) in enumerate(coords):               # _sphere_symmetry_fixed(fix_z_at_radius=R) (:1639-1657)
        if abs(y) < 1e-6 and x >= 0.0:
            fixed += [i * 6 + 1, i * 6 + 3, i * 6 + 5]
        if abs(x) < 1e-6 and y >= 0.0:
            fixed += [i * 6 + 0, i * 6 + 4, i * 6 + 5]
        if abs(x - R) < 1e-2 and abs(y) < 1e-2 and abs(z) < 1e-2:
            fixed.append(i * 6 + 2)
    fixed = np.array(sorted(set(fixed)), dtype=np.int64)
    i0 = int(np.argmin(np.abs(coords - [R, 0.0, 0.0]).sum(axis=1)))      # _find_node_by_xyz (:461-475)
    i90 = int(np.argmin(np.abs(coords - [0.0, R, 0.0]).sum(axis=1)))
    F = np.zeros(6 * coords.shape[0])
    F[i0 * 6 + 0] = P / 2.0                              # (:1759-1760)
    F[i90 * 6 + 1] = -P / 2.0
    save("asm_hemisphere_16", coords, conn, mat, t, F, fixed, dict(node0=i0, node90=i90), 9.3000e-2,
         lambda u: (abs(u[i0 * 6]) + abs(u[i90 * 6 + 1])) / 2.0)


def gen_twisted_16x96():
    L, b, t = 12.0, 1.1, 0.32
    mat = IsotropicMaterial("Ko2017_TwistedBeam", 29.0e6, 0.22, 1.0)
    mesh = meshes.twisted_beam(L, b, 90.0, 96, 16)
    coords, conn = mesh.coords, mesh.conn
    fixed = np.array(sorted(i * 6 + d for i in np.flatnonzero(np.abs(coords[:, 0]) < 1e-6) for d in range(6)),
                     dtype=np.int64)                     # clamped root (:1232-1236, :497-516)
    tip = int(np.argmin(np.abs(coords - [L, 0.0, 0.0]).sum(axis=1)))
    F = np.zeros(6 * coords.shape[0])
    F[tip * 6 + 2] = 1.0                                 # in-plane load case (:1321, :1360-1368)
    save("asm_twisted_16x96", coords, conn, mat, t, F, fixed, dict(node=tip), 5.4240e-3, lambda u: u[tip * 6 + 2])


if __name__ == "__main__":
    ALL = dict(scordelis_32=gen_scordelis_32, hemisphere_16=gen_hemisphere_16, twisted_16x96=gen_twisted_16x96,
               scordelis_16_floor=gen_scordelis_16_floor)
    for name in (sys.argv[1:] or list(ALL)):
        ALL[name]()
