"""ctypes front-end of the CPU oracle (oracle/libfemshell_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by fem_shell_amd (the product path).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

PROP_DTYPE = np.dtype([
    ("A", "f8", (9,)), ("B", "f8", (9,)), ("D", "f8", (9,)), ("Cs", "f8", (4,)),
    ("h", "f8"), ("m_t", "f8"), ("m_r", "f8"), ("k_drill", "f8"), ("C_raw", "f8", (9,)),
    ("rho", "f8"), ("flags", "i4"), ("nonlinear", "i4"),
])
assert PROP_DTYPE.itemsize == 368


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libfemshell_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
        _LIB.fso_csr_symbolic.restype = ctypes.c_int64
    return _LIB


def _p(a):
    return ctypes.c_void_p(a.ctypes.data) if a is not None else ctypes.c_void_p(0)


def as_props(props):
    props = np.ascontiguousarray(props)
    if props.dtype != PROP_DTYPE:
        props = np.frombuffer(props.tobytes(), dtype=PROP_DTYPE).copy()
    return props


def mitc4_geom(xyz):
    xyz = np.ascontiguousarray(xyz, dtype=np.float64).reshape(4, 3)
    e = np.zeros(9); xl = np.zeros(8); dj = np.zeros(4); ac = np.zeros(5)
    st = lib().fso_mitc4_geom(_p(xyz), _p(e), _p(xl), _p(dj), _p(ac))
    return e.reshape(3, 3), xl.reshape(4, 2), dj, ac, st


def mitc4_K_parts(xyz, prop):
    xyz = np.ascontiguousarray(xyz, dtype=np.float64).reshape(4, 3)
    prop = as_props(prop).reshape(-1)[:1]
    out = [np.zeros((24, 24)) for _ in range(4)]
    lib().fso_mitc4_K_parts(_p(xyz), _p(prop), *[_p(o) for o in out])
    return out


def mitc4_K_batch(xyz, prop_id, props, nthreads=1):
    xyz = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 4, 3)
    E = xyz.shape[0]
    prop_id = np.ascontiguousarray(prop_id, dtype=np.int32).reshape(E)
    props = as_props(props)
    Ke = np.empty((E, 24, 24)); st = np.zeros(E, dtype=np.int32)
    lib().fso_mitc4_K_batch(_p(xyz), _p(prop_id), _p(props), ctypes.c_int64(E), _p(Ke), _p(st),
                            ctypes.c_int(nthreads))
    return Ke, st


def mitc4_M_batch(xyz, prop_id, props, nthreads=1):
    xyz = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 4, 3)
    E = xyz.shape[0]
    prop_id = np.ascontiguousarray(prop_id, dtype=np.int32).reshape(E)
    props = as_props(props)
    Me = np.empty((E, 24, 24)); st = np.zeros(E, dtype=np.int32)
    lib().fso_mitc4_M_batch(_p(xyz), _p(prop_id), _p(props), ctypes.c_int64(E), _p(Me), _p(st),
                            ctypes.c_int(nthreads))
    return Me, st


def mitc3_batch(xyz, prop_id, props, want_K=True, want_M=False, nthreads=1):
    xyz = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3, 3)
    E = xyz.shape[0]
    prop_id = np.ascontiguousarray(prop_id, dtype=np.int32).reshape(E)
    props = as_props(props)
    Ke = np.empty((E, 18, 18)) if want_K else None
    Me = np.empty((E, 18, 18)) if want_M else None
    st = np.zeros(E, dtype=np.int32)
    lib().fso_mitc3_batch(_p(xyz), _p(prop_id), _p(props), ctypes.c_int64(E), _p(Ke), _p(Me), _p(st),
                          ctypes.c_int(nthreads))
    return Ke, Me, st


def solid_batch(nn, xyz, prop_id, props, want_K=True, want_M=False, body=None, sigma=None,
                reduced=False, nthreads=1):
    """Linear solids (nn = 4 TETRA4, 5 PYRAMID5, 6 WEDGE6, 8 HEXA8). props: [P,37] = C.ravel(), rho.
    Returns dict K, M, f (body load), KG (geometric stiffness from sigma [E,6]), status."""
    xyz = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, nn, 3)
    E, nd = xyz.shape[0], 3 * nn
    prop_id = np.ascontiguousarray(prop_id, dtype=np.int32).reshape(E)
    props = np.ascontiguousarray(props, dtype=np.float64).reshape(-1, 37)
    K = np.empty((E, nd, nd)) if want_K else None
    M = np.empty((E, nd, nd)) if want_M else None
    f = np.empty((E, nd)) if body is not None else None
    b = np.ascontiguousarray(body if body is not None else np.zeros(3), dtype=np.float64)
    sg = np.ascontiguousarray(sigma, dtype=np.float64).reshape(E, 6) if sigma is not None else None
    KG = np.empty((E, nd, nd)) if sigma is not None else None
    st = np.zeros(E, dtype=np.int32)
    lib().fso_solid_batch(ctypes.c_int(nn), ctypes.c_int(int(reduced)), _p(xyz), _p(prop_id), _p(props), _p(b),
                          _p(sg), ctypes.c_int64(E), _p(K), _p(M), _p(f), _p(KG), _p(st), ctypes.c_int(nthreads))
    return dict(K=K, M=M, f=f, KG=KG, status=st)


def mitc4_NL(xyz, prop, u):
    xyz = np.ascontiguousarray(xyz, dtype=np.float64).reshape(4, 3)
    prop = as_props(prop).reshape(-1)[:1]
    u = np.ascontiguousarray(u, dtype=np.float64).reshape(24)
    KT = np.zeros((24, 24)); f = np.zeros(24)
    st = lib().fso_mitc4_NL(_p(xyz), _p(prop), _p(u), _p(KT), _p(f))
    return KT, f, st


def mitc3_NL(xyz, prop, u):
    xyz = np.ascontiguousarray(xyz, dtype=np.float64).reshape(3, 3)
    prop = as_props(prop).reshape(-1)[:1]
    u = np.ascontiguousarray(u, dtype=np.float64).reshape(18)
    KT = np.zeros((18, 18)); f = np.zeros(18)
    st = lib().fso_mitc3_NL(_p(xyz), _p(prop), _p(u), _p(KT), _p(f))
    return KT, f, st


def shell_KG(xyz, prop, sigma):
    """compute_geometric_stiffness(sigma_local[3]) of a MITC3 (xyz [3,3]) or MITC4 (xyz [4,3])."""
    xyz = np.ascontiguousarray(xyz, dtype=np.float64)
    nn = xyz.shape[0]
    prop = as_props(prop).reshape(-1)[:1]
    sg = np.ascontiguousarray(sigma, dtype=np.float64).reshape(3)
    KG = np.zeros((6 * nn, 6 * nn))
    lib().fso_shell_KG(ctypes.c_int(nn), _p(xyz), _p(prop), _p(sg), _p(KG))
    return KG


def centrifugal(xyz, rho, omega, axis, center, shell=True):
    """compute_centrifugal_prestress: MITC3 (local [sxx,syy,sxy]) or a linear solid (Voigt 6)."""
    xyz = np.ascontiguousarray(xyz, dtype=np.float64)
    ax = np.ascontiguousarray(axis, dtype=np.float64); ce = np.ascontiguousarray(center, dtype=np.float64)
    if shell:
        out = np.zeros(3)
        lib().fso_mitc3_centrifugal(_p(xyz), ctypes.c_double(rho), ctypes.c_double(omega), _p(ax), _p(ce), _p(out))
    else:
        out = np.zeros(6)
        lib().fso_solid_centrifugal(ctypes.c_int(xyz.shape[0]), _p(xyz), ctypes.c_double(rho),
                                    ctypes.c_double(omega), _p(ax), _p(ce), _p(out))
    return out


def csr_assemble(dofs, ndof, vals_e):
    """Pattern + element-order summation (core/assembler.py:222-241, 269-303)."""
    dofs = np.ascontiguousarray(dofs, dtype=np.int64)
    E, nd = dofs.shape
    indptr = np.zeros(ndof + 1, dtype=np.int64)
    nnz = lib().fso_csr_symbolic(_p(dofs), ctypes.c_int64(E), ctypes.c_int(nd),
                                 ctypes.c_int64(ndof), _p(indptr), ctypes.c_void_p(0))
    indices = np.zeros(nnz, dtype=np.int32)
    lib().fso_csr_symbolic(_p(dofs), ctypes.c_int64(E), ctypes.c_int(nd), ctypes.c_int64(ndof),
                           _p(indptr), _p(indices))
    data = None
    if vals_e is not None:
        vals_e = np.ascontiguousarray(vals_e, dtype=np.float64)
        data = np.zeros(nnz)
        rc = lib().fso_csr_numeric(_p(dofs), ctypes.c_int64(E), ctypes.c_int(nd), _p(vals_e),
                                   _p(indptr), _p(indices), _p(data))
        assert rc == 0
    return indptr, indices, data


def csr_numeric_rows(dofs, ndof, vals_e, indptr, indices, nthreads=0):
    """Row-parallel element-order summation (fso_csr_numeric_rows): bit-identical to csr_assemble's values."""
    dofs = np.ascontiguousarray(dofs, dtype=np.int64)
    vals_e = np.ascontiguousarray(vals_e, dtype=np.float64)
    E, nd = dofs.shape
    data = np.zeros(indices.shape[0])
    rc = lib().fso_csr_numeric_rows(_p(dofs), ctypes.c_int64(E), ctypes.c_int(nd), _p(vals_e), ctypes.c_int64(ndof),
                                    _p(indptr), _p(indices), _p(data), ctypes.c_int(nthreads))
    assert rc == 0
    return data


def csr_assemble_mixed(dofs_list, ndof, ke_list):
    """Mixed-mesh branch of the reference assembler (core/assembler.py:232-241 pattern by set
    union, :287-292 values added element by element).  Pure NumPy/Python: small meshes only."""
    rows = [set() for _ in range(ndof)]
    for dofs in dofs_list:
        for r in dofs:
            rows[int(r)].update(int(c) for c in dofs)
    indptr = np.zeros(ndof + 1, dtype=np.int64)
    indptr[1:] = np.cumsum([len(s) for s in rows])
    indices = np.fromiter((c for s in rows for c in sorted(s)), dtype=np.int32, count=int(indptr[-1]))
    data = np.zeros(indices.shape[0])
    for dofs, ke in zip(dofs_list, ke_list):
        for a, r in enumerate(dofs):
            lo, hi = indptr[r], indptr[r + 1]
            pos = lo + np.searchsorted(indices[lo:hi], dofs)
            data[pos] += ke[a]
    return indptr, indices, data


# ---------------------------------------------------------------------------------------------
# Dirichlet reduction / lumping restated with SciPy (core/bc.py:116-290,
# solvers/fsi/linear_dynamic.py:227-234).  PARITY UNPINNED for the PETSc calls themselves (petsc4py
# is absent and no reference test pins them): Mat.zeroRows / createSubMatrix / getRowSum are restated
# from their documented semantics; the reduced solve is pinned by tests/golden/asm_scordelis_16.npz.
# ---------------------------------------------------------------------------------------------
def bc_apply(K, F, fixed: dict, M=None):
    """K.zeroRows(fixed, 1.0, U_fixed, F); M.zeroRows(fixed, 0.0) on scipy CSR copies (pattern kept)."""
    import scipy.sparse as sp
    K = sp.csr_matrix(K, copy=True); F = np.array(F, dtype=float)
    M = sp.csr_matrix(M, copy=True) if M is not None else None
    for r, v in fixed.items():
        lo, hi = K.indptr[r], K.indptr[r + 1]
        K.data[lo:hi] = np.where(K.indices[lo:hi] == r, 1.0, 0.0)
        F[r] = 1.0 * v
        if M is not None:
            M.data[M.indptr[r]:M.indptr[r + 1]] = 0.0
    return K, F, M


def bc_reduced(K, F, fixed: dict, M=None):
    """(K_red, F_red, M_red, free): createSubMatrix(free, free) and F_free - K_free,fixed u_fixed."""
    n = K.shape[0]
    fx = np.array(sorted(fixed), dtype=np.int64)
    free = np.setdiff1d(np.arange(n), fx)
    uf = np.array([fixed[d] for d in fx], dtype=float)
    K_red = K[free, :][:, free].tocsr()
    K_red.sort_indices()
    F_red = F[free] - (K[free, :][:, fx] @ uf if len(fx) else 0.0)
    M_red = None
    if M is not None:
        M_red = M[free, :][:, free].tocsr()
        M_red.sort_indices()
    return K_red, F_red, M_red, free
