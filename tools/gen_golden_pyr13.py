#!/usr/bin/env python3
"""PYRAMID13 fixtures from the UNMODIFIED reference (elements/SOLID.py:1221-1537, core/assembler.py,
postprocess/stress_recovery.py), through tools/ref_shim.py.  Build container only.

    python tools/gen_golden_pyr13.py      # writes tests/golden/elem_pyr13.npz, asm_p13.npz, stress_p13.npz

elem_pyr13.npz   pts13 / w13 / vals13: integration points, weights, N and dN at them (bit-for-bit pins of the host
                 table); 12 seeded elements (iso / ortho / rotated ortho): K, M, body_load, geometric stiffness
asm_p13.npz      p13_: two HEXA20 bricks, each with a PYRAMID13 on its top face (shared 8-node faces, the pyramids
                 share an edge), through the reference's MeshAssembler: dofs, _row_nnz, CSR of K, M, load vector
stress_p13.npz   every StressRecovery product of that mesh for a seeded u (pinv extrapolation, 18 / 27 Gauss points)
Fixtures are DATA (inputs and reference outputs); no reference source is stored.
"""
import contextlib
import io
import os
import sys
from types import SimpleNamespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import gen_golden_asm as GA  # noqa: E402  (installs the shim and the recording PETSc stand-in)
from gen_golden import ISO, SOLID2_REF, t300  # noqa: E402

from fem_shell.core.material import IsotropicMaterial  # noqa: E402
from fem_shell.elements import ElementFamily  # noqa: E402
from fem_shell.elements.SOLID import PYRAMID13  # noqa: E402

PYR13_REF = np.array([(-1, -1, 0), (1, -1, 0), (1, 1, 0), (-1, 1, 0), (0, 0, 1), (0, -1, 0), (-1, 0, 0), (-.5, -.5, .5),
                      (1, 0, 0), (.5, -.5, .5), (0, 1, 0), (.5, .5, .5), (-.5, .5, .5)], float)    # Gmsh type 19 order


def gen_elements():
    rng = np.random.default_rng(19)
    nn = 13
    el0 = PYRAMID13(PYR13_REF, tuple(range(nn)), IsotropicMaterial("m", 1.0, 0.3, 1.0))
    pts, w = el0.integration_points
    vals = np.empty((len(w), nn, 4))
    for q, pt in enumerate(pts):
        vals[q, :, 0] = el0.shape_functions(*pt)
        vals[q, :, 1], vals[q, :, 2], vals[q, :, 3] = el0.shape_function_derivatives(*pt)
    body = np.array([1.0, -2.0, 9.81])
    xyz, props, K, M, F, KG, sig = [], [], [], [], [], [], []
    for t in range(12):
        x = PYR13_REF.copy() if t == 0 else PYR13_REF + 0.05 * rng.standard_normal((nn, 3))
        x = x * rng.choice([1.0, 1e-2, 30.0]) + (rng.choice([0.0, 5.0]) * rng.standard_normal(3) if t else 0.0)
        if t % 3 == 2:
            mat = t300()
            ori = np.linalg.qr(rng.standard_normal((3, 3)))[0] if t % 2 else None
        else:
            E, nu, rho = ISO[t % len(ISO)]
            mat, ori = IsotropicMaterial("m", E, nu, rho), None
        el = PYRAMID13(x, tuple(range(nn)), mat, orientation=ori)
        sg = rng.standard_normal(6) * 1e6
        xyz.append(x); props.append(np.concatenate([el.C.ravel(), [mat.rho]]))
        K.append(el.K); M.append(el.M); F.append(el.body_load(body))
        KG.append(el.compute_geometric_stiffness(sg)); sig.append(sg)
    np.savez_compressed(os.path.join(GA.OUT, "elem_pyr13.npz"), body=body, pts13=pts, w13=w, vals13=vals,
                        xyz13=np.array(xyz), props13=np.array(props), K13=np.array(K), M13=np.array(M),
                        F13=np.array(F), KG13=np.array(KG), sigma13=np.array(sig))
    print("elem_pyr13.npz")


def gen_mesh():
    rng = np.random.default_rng(53)
    pts, rows, key = [], [], {}

    def add(ref, shift):
        row = []
        for p in ref:
            q = tuple(np.round(p + np.asarray(shift, float), 6))
            if q not in key:
                key[q] = len(pts); pts.append(q)
            row.append(key[q])
        rows.append(row + [-1] * (20 - len(row)))

    for i in range(2):
        add(SOLID2_REF[20], (2 * i, 0, 0))
        add(PYR13_REF, (2 * i, 0, 1))
    coords = np.array(pts, float) + 0.02 * rng.standard_normal((len(pts), 3))
    conn = np.array(rows, dtype=np.int64)
    out, sout = {}, {}
    with contextlib.redirect_stdout(io.StringIO()):
        asm = GA.MeshAssembler(GA.RefMesh(coords, conn), {"elements": {
            "element_family": ElementFamily.SOLID, "material": IsotropicMaterial("steel", 2.1e11, 0.3, 7800.0)}})
        out.update(p13_coords=coords, p13_conn=conn)
        GA.asm_common(asm, "p13_", out)
        out["p13_dofs_list"] = np.concatenate(asm._dofs_list)
        body = np.array([1.0, -2.0, -9.81])
        out["p13_body"] = body
        out["p13_F"] = asm.assemble_load_vector(SimpleNamespace(value=body)).array
        GA.stress_fields(asm, GA.seeded_u(rng, coords, 3), "p13_", sout, shell=False)
    names = sorted(type(e).__name__ for e in asm._element_map.values())
    assert names == ["HEXA20", "HEXA20", "PYRAMID13", "PYRAMID13"], names
    np.savez_compressed(os.path.join(GA.OUT, "asm_p13.npz"), **out)
    np.savez_compressed(os.path.join(GA.OUT, "stress_p13.npz"), **sout)
    print("asm_p13.npz", len(out), "arrays; stress_p13.npz", len(sout), "arrays")


if __name__ == "__main__":
    gen_elements()
    gen_mesh()
