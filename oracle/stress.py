"""CPU restatement (NumPy) of the reference's stress / strain recovery
(postprocess/stress_recovery.py:280-1402) for the shell and solid families.

TEST INFRASTRUCTURE ONLY: imported by tests/ -- never by fem_shell_amd (the product path).  Pinned against the
reference's own StressRecovery through tests/golden/stress_ref.npz (tests/test_oracle_golden.py).

Inputs are plain arrays: coords [N,3], conn [E,nn] (short rows padded with -1), per-element property data and the
global displacement vector.  Element loops are Python (small cases only)."""
import numpy as np

from . import oracle as O

GP = 1.0 / np.sqrt(3.0)


def _quad_local(x):
    """MITC4.py:436-476 (_compute_local_coordinates): frame from the two triangle normals, local coordinates as
    dot products on ABSOLUTE coordinates."""
    frame, xl, _, _, _ = O.mitc4_geom(x)         # the C oracle's geometry stage is pinned bit for bit
    return frame, xl


def _quad_dN(xl, r, s):
    """MITC4.py:141-155, 614-657 (_get_dH)."""
    dxi = np.array([-0.25 * (1 - s), 0.25 * (1 - s), 0.25 * (1 + s), -0.25 * (1 + s)])
    deta = np.array([-0.25 * (1 - r), -0.25 * (1 + r), 0.25 * (1 + r), 0.25 * (1 - r)])
    J = np.array([[dxi @ xl[:, 0], dxi @ xl[:, 1]], [deta @ xl[:, 0], deta @ xl[:, 1]]])
    Ji = np.linalg.inv(J)
    return Ji[0, 0] * dxi + Ji[0, 1] * deta, Ji[1, 0] * dxi + Ji[1, 1] * deta, float(np.linalg.det(J))


def _tri_local(x):
    """MITC3.py:375-418: node 1 at the origin, e1 along edge 1-2."""
    v12, v13 = x[1] - x[0], x[2] - x[0]
    e1 = v12 / np.linalg.norm(v12)
    n = np.cross(v12, v13)
    e3 = n / np.linalg.norm(n)
    e2 = np.cross(e3, e1)
    return np.array([[0.0, 0.0], [v12 @ e1, v12 @ e2], [v13 @ e1, v13 @ e2]])


def _tri_dN(xl):
    """MITC3.py:439-469, 527-558."""
    J = np.array([[xl[1, 0] - xl[0, 0], xl[1, 1] - xl[0, 1]], [xl[2, 0] - xl[0, 0], xl[2, 1] - xl[0, 1]]])
    Ji = np.linalg.inv(J)
    dr, ds = np.array([-1.0, 1.0, 0.0]), np.array([-1.0, 0.0, 1.0])
    return Ji[0, 0] * dr + Ji[0, 1] * ds, Ji[1, 0] * dr + Ji[1, 1] * ds, float(J[0, 0] * J[1, 1] - J[0, 1] * J[1, 0])


def shell_point(x, u_e, Cm, h, r, s, z, parts=3, strain=False):
    """_compute_shell_stress / _compute_shell_strain (:529-650): B_m / B_kappa of MITC4.py:756-790 /
    MITC3.py:670-739 applied to the GLOBAL-axis element vector, as the reference does."""
    nn = x.shape[0]
    if nn == 4:
        _, xl = _quad_local(x)
        dNx, dNy, _ = _quad_dN(xl, r, s)
    else:
        dNx, dNy, _ = _tri_dN(_tri_local(x))
    U = u_e.reshape(nn, 6)
    em = np.array([dNx @ U[:, 0], dNy @ U[:, 1], dNy @ U[:, 0] + dNx @ U[:, 1]])
    kp = np.array([dNx @ U[:, 4], -(dNy @ U[:, 3]), dNy @ U[:, 4] - dNx @ U[:, 3]])
    if strain:
        return em + z * kp
    C = Cm / h
    out = np.zeros(3)
    if parts & 1:
        out += C @ em
    if parts & 2:
        out += C @ (z * kp)
    return out


def shell_area(x):
    """:1372-1388: sum of det J over the element's own rule (2x2, weights 1; Hammer, weights 1/3)."""
    if x.shape[0] == 4:
        _, xl = _quad_local(x)
        return abs(sum(_quad_dN(xl, r, s)[2] for r, s in ((-GP, -GP), (GP, -GP), (GP, GP), (-GP, GP))))
    det = _tri_dN(_tri_local(x))[2]
    return abs(det / 3 + det / 3 + det / 3)


# ---- solids --------------------------------------------------------------------------------------------
def solid_table(nn, reduced=False):
    """Integration points, weights and N / dN at them, SOLID.py (the linear rules as in fso_solid; the quadratic
    elements through the same formulas as the reference incl. HEXA20's forward differences)."""
    if nn in (10, 13, 15, 20):
        return _quadratic_table(nn)
    if nn == 8:
        if reduced:
            pts, w = np.zeros((1, 3)), np.array([8.0])
        else:
            pts = np.array([[(GP if q & 1 else -GP), (GP if q & 2 else -GP), (GP if q & 4 else -GP)] for q in range(8)])
            w = np.ones(8)
    elif nn == 4:
        pts, w = np.full((1, 3), 0.25), np.array([1.0 / 6])
    elif nn == 6:
        tp = [(1 / 6, 1 / 6), (2 / 3, 1 / 6), (1 / 6, 2 / 3)]
        pts = np.array([[tp[q // 2][0], tp[q // 2][1], (GP if q & 1 else -GP)] for q in range(6)])
        w = np.full(6, (1.0 / 6) * 1.0 * 2)
    else:
        a, za, wa = 0.26318405556971359557, 0.54415184401122528880, 0.10078588207982543059
        b, zb, wb = 0.50661630334978742377, 0.12251482265544137787, 0.23254745125350790274
        sx, sy = (-1, 1, 1, -1), (-1, -1, 1, 1)
        pts = np.array([[sx[q] * a, sy[q] * a, za] for q in range(4)] + [[sx[q] * b, sy[q] * b, zb] for q in range(4)])
        w = np.array([wa] * 4 + [wb] * 4)
    vals = np.array([_linear_shape(nn, *p) for p in pts])
    return pts, w, vals


def _linear_shape(nn, xi, eta, zeta):
    v = np.zeros((nn, 4))
    if nn == 8:
        sg = [(-1, -1, -1), (1, -1, -1), (1, 1, -1), (-1, 1, -1), (-1, -1, 1), (1, -1, 1), (1, 1, 1), (-1, 1, 1)]
        for i, (a, b, c) in enumerate(sg):
            fx, fy, fz = 1 + a * xi, 1 + b * eta, 1 + c * zeta
            v[i] = [0.125 * fx * fy * fz, 0.125 * a * fy * fz, 0.125 * b * fx * fz, 0.125 * c * fx * fy]
    elif nn == 4:
        v[:, 0] = [1 - xi - eta - zeta, xi, eta, zeta]
        v[:, 1:] = [[-1, -1, -1], [1, 0, 0], [0, 1, 0], [0, 0, 1]]
    elif nn == 6:
        L, dx, de = (1 - xi - eta, xi, eta), (-1, 1, 0), (-1, 0, 1)
        for c, (Lz, dz) in enumerate(((0.5 * (1 - zeta), -0.5), (0.5 * (1 + zeta), 0.5))):
            for i in range(3):
                v[3 * c + i] = [L[i] * Lz, dx[i] * Lz, de[i] * Lz, L[i] * dz]
    else:
        r = max(1 - zeta, 1e-14)
        rmx, rpx, rme, rpe = r - xi, r + xi, r - eta, r + eta
        v[:, 0] = [0.25 * rmx * rme / r, 0.25 * rpx * rme / r, 0.25 * rpx * rpe / r, 0.25 * rmx * rpe / r, zeta]
        v[:, 1] = [-0.25 * rme / r, 0.25 * rme / r, 0.25 * rpe / r, -0.25 * rpe / r, 0]
        v[:, 2] = [-0.25 * rmx / r, -0.25 * rpx / r, 0.25 * rpx / r, 0.25 * rmx / r, 0]
        v[:, 3] = [-0.25 * (rme + rmx) / r + 0.25 * rmx * rme / r ** 2, -0.25 * (rme + rpx) / r + 0.25 * rpx * rme / r ** 2,
                   -0.25 * (rpe + rpx) / r + 0.25 * rpx * rpe / r ** 2, -0.25 * (rpe + rmx) / r + 0.25 * rmx * rpe / r ** 2, 1.0]
    return v


def _hexa20_N(xi, eta, zeta):
    c = [(-1, -1, -1), (1, -1, -1), (1, 1, -1), (-1, 1, -1), (-1, -1, 1), (1, -1, 1), (1, 1, 1), (-1, 1, 1)]
    N = np.zeros(20)
    for i, (a, b, d) in enumerate(c):
        N[i] = 0.125 * (1 + xi * a) * (1 + eta * b) * (1 + zeta * d) * (xi * a + eta * b + zeta * d - 2)
    x2, e2, z2 = 1 - xi ** 2, 1 - eta ** 2, 1 - zeta ** 2
    N[8:] = [0.25 * x2 * (1 - eta) * (1 - zeta), 0.25 * (1 - xi) * e2 * (1 - zeta), 0.25 * (1 - xi) * (1 - eta) * z2,
             0.25 * (1 + xi) * e2 * (1 - zeta), 0.25 * (1 + xi) * (1 - eta) * z2, 0.25 * x2 * (1 + eta) * (1 - zeta),
             0.25 * (1 + xi) * (1 + eta) * z2, 0.25 * (1 - xi) * (1 + eta) * z2, 0.25 * x2 * (1 - eta) * (1 + zeta),
             0.25 * (1 - xi) * e2 * (1 + zeta), 0.25 * (1 + xi) * e2 * (1 + zeta), 0.25 * x2 * (1 + eta) * (1 + zeta)]
    return N


def _pyramid13_table():
    """SOLID.py:1268-1537: 18-point Felippa rule; serendipity functions of the scaled base times vertical blends."""
    u, z = (0.35309846330877704481, 0.67969709567986745790), (0.54415184401122528880, 0.12251482265544137787)
    ww = ((0.023330065296255886709, 0.037328104474009418735, 0.059724967158415069975),
          (0.053830428530904607120, 0.086128685649447371390, 0.137805897039115794220))
    pts = np.array([[i * u[l], j * u[l], z[l]] for l in range(2) for j in (-1, 0, 1) for i in (-1, 0, 1)])
    w = np.array([ww[l][(i == 0) + (j == 0)] for l in range(2) for j in (-1, 0, 1) for i in (-1, 0, 1)])
    vals = np.zeros((18, 13, 4))
    sx, sy = (-1, 1, 1, -1), (-1, -1, 1, 1)
    for q, (xi, eta, zeta) in enumerate(pts):
        r = max(1 - zeta, 1e-14)
        xs, es = np.clip(xi / r, -1, 1), np.clip(eta / r, -1, 1)
        hb, hl, dhb, dhl = r * (1 - 2 * zeta), 4 * zeta * r, -3 + 4 * zeta, 4 * (1 - 2 * zeta)
        fn = {}                                    # node -> (f, df/dxs, df/des, h, dh)
        for i in range(4):
            a, b = 1 + sx[i] * xs, 1 + sy[i] * es
            fn[i] = (0.25 * a * b * (sx[i] * xs + sy[i] * es - 1), 0.25 * b * (2 * xs + sx[i] * sy[i] * es),
                     0.25 * a * (sx[i] * sy[i] * xs + 2 * es), hb, dhb)
            fn[(7, 9, 11, 12)[i]] = (0.25 * a * b, 0.25 * sx[i] * b, 0.25 * sy[i] * a, hl, dhl)
        fn[5] = (0.5 * (1 - xs * xs) * (1 - es), -xs * (1 - es), -0.5 * (1 - xs * xs), hb, dhb)
        fn[6] = (0.5 * (1 - xs) * (1 - es * es), -0.5 * (1 - es * es), -(1 - xs) * es, hb, dhb)
        fn[8] = (0.5 * (1 + xs) * (1 - es * es), 0.5 * (1 - es * es), -(1 + xs) * es, hb, dhb)
        fn[10] = (0.5 * (1 - xs * xs) * (1 + es), -xs * (1 + es), 0.5 * (1 - xs * xs), hb, dhb)
        for n, (f, fx, fe, h, dh) in fn.items():
            vals[q, n] = [f * h, fx * h / r, fe * h / r, (fx * xi + fe * eta) / (r * r) * h + f * dh]
        vals[q, 4] = [zeta * (2 * zeta - 1), 0.0, 0.0, 4 * zeta - 1]
    return pts, w, vals


def _quadratic_table(nn):
    if nn == 13:
        return _pyramid13_table()
    if nn == 20:
        s35 = np.sqrt(3 / 5)
        p1, w1 = np.array([-s35, 0, s35]), np.array([5 / 9, 8 / 9, 5 / 9])
        pts = np.array([[p1[i], p1[j], p1[k]] for k in range(3) for j in range(3) for i in range(3)])
        w = np.array([w1[i] * w1[j] * w1[k] for k in range(3) for j in range(3) for i in range(3)])
        h = 1e-8
        vals = np.zeros((27, 20, 4))
        for q, (xi, eta, zeta) in enumerate(pts):
            xi, eta, zeta = np.float64(xi), np.float64(eta), np.float64(zeta)
            N0 = _hexa20_N(xi, eta, zeta)
            vals[q, :, 0] = N0
            vals[q, :, 1] = (_hexa20_N(xi + h, eta, zeta) - N0) / h
            vals[q, :, 2] = (_hexa20_N(xi, eta + h, zeta) - N0) / h
            vals[q, :, 3] = (_hexa20_N(xi, eta, zeta + h) - N0) / h
        return pts, w, vals
    if nn == 10:
        a, b = (5 - np.sqrt(5)) / 20, (5 + 3 * np.sqrt(5)) / 20
        pts = np.array([[a, a, a], [b, a, a], [a, b, a], [a, a, b]])
        w = np.full(4, 1 / 24)
        vals = np.zeros((4, 10, 4))
        dL = np.array([[-1, -1, -1], [1, 0, 0], [0, 1, 0], [0, 0, 1]], float)
        edges = [(0, 1), (1, 2), (2, 0), (0, 3), (2, 3), (1, 3)]
        for q, (xi, eta, zeta) in enumerate(pts):
            L = np.array([1 - xi - eta - zeta, xi, eta, zeta])
            for i in range(4):
                vals[q, i, 0] = L[i] * (2 * L[i] - 1)
                vals[q, i, 1:] = (4 * L[i] - 1) * dL[i]
            for m, (p, r) in enumerate(edges):
                vals[q, 4 + m, 0] = 4 * L[p] * L[r]
                vals[q, 4 + m, 1:] = 4 * (L[r] * dL[p] + L[p] * dL[r])
        return pts, w, vals
    tp = np.array([[1 / 3, 1 / 3], [0.797426985353087, 0.101286507323456], [0.101286507323456, 0.797426985353087],
                   [0.101286507323456, 0.101286507323456], [0.470142064105115, 0.059715871789770],
                   [0.470142064105115, 0.470142064105115], [0.059715871789770, 0.470142064105115]])
    tw = np.array([0.225, 0.125939180544827, 0.125939180544827, 0.125939180544827, 0.132394152788506,
                   0.132394152788506, 0.132394152788506]) * 0.5
    s35 = np.sqrt(3 / 5)
    lp, lw = np.array([-s35, 0, s35]), np.array([5 / 9, 8 / 9, 5 / 9])
    pts = np.array([[tp[i, 0], tp[i, 1], lp[j]] for i in range(7) for j in range(3)])
    w = np.array([tw[i] * lw[j] * 2 for i in range(7) for j in range(3)])
    vals = np.zeros((21, 15, 4))
    dx, de = np.array([-1.0, 1.0, 0.0]), np.array([-1.0, 0.0, 1.0])
    for q, (xi, eta, zeta) in enumerate(pts):
        L = np.array([1 - xi - eta, xi, eta])
        Lz = (0.5 * zeta * (zeta - 1), 0.5 * zeta * (zeta + 1))
        dLz = (zeta - 0.5, zeta + 0.5)
        L0, dL0 = 1 - zeta ** 2, -2 * zeta
        for c in range(2):
            for i in range(3):
                vals[q, 3 * c + i] = [L[i] * (2 * L[i] - 1) * Lz[c], (4 * L[i] - 1) * dx[i] * Lz[c],
                                      (4 * L[i] - 1) * de[i] * Lz[c], L[i] * (2 * L[i] - 1) * dLz[c]]
            for n, (a, b) in zip(((6, 7, 9), (12, 13, 14))[c], ((0, 1), (0, 2), (1, 2))):
                vals[q, n] = [4 * L[a] * L[b] * Lz[c], 4 * (dx[a] * L[b] + L[a] * dx[b]) * Lz[c],
                              4 * (de[a] * L[b] + L[a] * de[b]) * Lz[c], 4 * L[a] * L[b] * dLz[c]]
        for n, i in zip((8, 10, 11), range(3)):
            vals[q, n] = [L[i] * L0, dx[i] * L0, de[i] * L0, L[i] * dL0]
    return pts, w, vals


def solid_gauss(x, u_e, C, table):
    """_compute_solid_gauss_stresses (:655-705): sigma, eps [n_gp, 6] and the element volume (:1391-1402)."""
    pts, w, vals = table
    nn = x.shape[0]
    U = u_e.reshape(nn, 3)
    sig, eps, vol = np.zeros((len(w), 6)), np.zeros((len(w), 6)), 0.0
    for q in range(len(w)):
        dN = vals[q, :, 1:].T                      # [3, nn]
        J = dN @ x
        g = np.linalg.inv(J) @ dN                  # physical gradients [3, nn]
        H = g @ U                                  # H[a][c] = d u_c / d x_a
        e = np.array([H[0, 0], H[1, 1], H[2, 2], H[1, 0] + H[0, 1], H[2, 1] + H[1, 2], H[2, 0] + H[0, 2]])
        eps[q], sig[q] = e, C @ e
        vol += abs(np.linalg.det(J)) * w[q]
    return sig, eps, vol


def extrapolation(table, nn):
    """:437-492."""
    N_gp = table[2][:, :, 0]
    return np.linalg.inv(N_gp) if N_gp.shape[0] == nn else np.linalg.pinv(N_gp)


def invariants(s, is_3d):
    """_build_stress_result (:1087-1296): von Mises, s1, s2, s3 (3-D), tau_max, angle."""
    sxx, syy, szz, txy, tyz, tzx = (s[:, k] for k in range(6))
    vm = np.sqrt(0.5 * ((sxx - syy) ** 2 + (syy - szz) ** 2 + (szz - sxx) ** 2) + 3.0 * (txy ** 2 + tyz ** 2 + tzx ** 2))
    if not is_3d:
        avg, diff = (sxx + syy) / 2, (sxx - syy) / 2
        R = np.sqrt(diff ** 2 + txy ** 2)
        return dict(von_mises=vm, sigma_1=avg + R, sigma_2=avg - R, tau_max=R, angle=0.5 * np.arctan2(2 * txy, sxx - syy))
    I1 = sxx + syy + szz
    I2 = sxx * syy + syy * szz + szz * sxx - txy ** 2 - tyz ** 2 - tzx ** 2
    I3 = sxx * syy * szz + 2 * txy * tyz * tzx - sxx * tyz ** 2 - syy * tzx ** 2 - szz * txy ** 2
    p = np.maximum(I1 ** 2 / 3.0 - I2, 0.0)
    q = 2.0 * I1 ** 3 / 27.0 - I1 * I2 / 3.0 + I3
    sqp = np.sqrt(p / 3.0)
    den = 2.0 * (p / 3.0) * sqp
    den = np.where(den == 0, 1.0, den)
    phi = np.arccos(np.clip(-q / den, -1.0, 1.0)) / 3.0
    st = np.sort(np.stack([I1 / 3.0 + 2.0 * sqp * np.cos(phi - 2.0 * np.pi * k / 3.0) for k in range(3)], -1), axis=-1)
    return dict(von_mises=vm, sigma_1=st[:, 2], sigma_2=st[:, 1], sigma_3=st[:, 0], tau_max=(st[:, 2] - st[:, 0]) / 2.0,
                angle=np.zeros(s.shape[0]))


def recover(coords, conn, u, family, elem_data, location=0, parts=3, strain=False, nodal=False, weighted=False,
            point=(0.0, 0.0), reduced=False):
    """compute_element_stresses / compute_nodal_stresses / ..._strains (:710-866, 958-1081) -> Voigt rows [n, 6].
    family 'shell': elem_data[e] = (Cm [3,3], h); 'solid': elem_data[e] = C [6,6]."""
    n_nodes, E = coords.shape[0], conn.shape[0]
    d = 6 if family == "shell" else 3
    out = np.zeros((n_nodes if nodal else E, 6))
    wsum = np.zeros(n_nodes)
    tables, extrap = {}, {}
    node_pts = {3: [(0.0, 0.0), (1.0, 0.0), (0.0, 1.0)], 4: [(-1.0, -1.0), (1.0, -1.0), (1.0, 1.0), (-1.0, 1.0)]}
    for e in range(E):
        nodes = conn[e][conn[e] >= 0]
        nn = len(nodes)
        x = coords[nodes]
        u_e = u.reshape(-1, d)[nodes].reshape(-1)
        if family == "shell":
            Cm, h = elem_data[e]
            z = {1: h / 2, -1: -h / 2}.get(location, 0.0)
            wgt = shell_area(x) if weighted else 1.0
            if nodal:
                for i, (r, s) in enumerate(node_pts[nn]):
                    out[nodes[i], :3] += wgt * shell_point(x, u_e, Cm, h, r, s, z, parts, strain)
                    wsum[nodes[i]] += wgt
            else:
                out[e, :3] = shell_point(x, u_e, Cm, h, point[0], point[1], z, parts, strain)
        else:
            if nn not in tables:
                tables[nn] = solid_table(nn, reduced)
                extrap[nn] = extrapolation(tables[nn], nn)
            sig, eps, vol = solid_gauss(x, u_e, elem_data[e], tables[nn])
            val = eps if strain else sig
            if nodal:
                wgt = vol if weighted else 1.0
                vn = extrap[nn] @ val
                for i in range(nn):
                    out[nodes[i]] += wgt * vn[i]
                    wsum[nodes[i]] += wgt
            else:
                out[e] = val.mean(axis=0)
    if nodal:
        m = wsum > 0
        out[m] = out[m] / wsum[m, None]
        out[~m] = 0.0
    return out
