// femshell_geom.h -- device geometry helpers shared by the element-matrix kernels (femshell_hip.hip)
// and the stress-recovery kernels (femshell_stress.hip).  Internal; included inside each file's
// anonymous namespace region.  Compile with -ffp-contract=off (see femshell_hip.hip header).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "femshell_hip.h"

namespace {

// ------------------------------------------------------------------------------------------
// rounding-pinned primitives (file default: contraction OFF)
// ------------------------------------------------------------------------------------------
#define FMA(a, b, c) __builtin_fma((a), (b), (c))

__device__ __forceinline__ double dot3(const double *a, const double *b) {
    // np.dot on contiguous length-3 float64 == sequential FMA chain
    double s = a[0] * b[0];
    s = FMA(a[1], b[1], s);
    s = FMA(a[2], b[2], s);
    return s;
}
__device__ __forceinline__ double dots4(double a0, double a1, double a2, double a3, double b0,
                                        double b1, double b2, double b3) {
    // np.dot(dN, xl[:, k]) (strided operand, n = 4): OpenBLAS generic-stride ddot
    double m2 = a2 * b2, m3 = a3 * b3;
    double t1 = FMA(a0, b0, m2), t2 = FMA(a1, b1, m3);
    return t1 + t2;
}
__device__ __forceinline__ void cross3(const double *a, const double *b, double *c) {
    // np.cross: rounded products, then subtraction
    double p, q;
    p = a[1] * b[2]; q = a[2] * b[1]; c[0] = p - q;
    p = a[2] * b[0]; q = a[0] * b[2]; c[1] = p - q;
    p = a[0] * b[1]; q = a[1] * b[0]; c[2] = p - q;
}
// IEEE square root without the range handling of the compiler's f64 sqrt expansion: the same v_rsq_f64 start and the same
// Newton steps (so the same bits for every operand the expansion does not rescale, i.e. above 2^-767), minus the compare /
// ldexp pairs that rescale smaller operands -- squared lengths of mesh edges are nowhere near.  +-0 and +inf return
// themselves, negative operands and NaN give NaN, as sqrt does.
__device__ __forceinline__ double fsqrt(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * 0.5;
    const double r = FMA(-h, g, 0.5);
    g = FMA(g, r, g);
    double d = FMA(-g, g, x);
    h = FMA(h, r, h);
    g = FMA(d, h, g);
    d = FMA(-g, g, x);
    g = FMA(d, h, g);
    return __builtin_amdgcn_class(x, 0x260) ? x : g;      // +-0, +inf
}
__device__ __forceinline__ double norm3(const double *a) { return fsqrt(dot3(a, a)); }

// IEEE division with a SHARED reciprocal: the same arithmetic as the compiler's f64 fdiv expansion
// (v_rcp_f64, two Newton steps, quotient, residual, correction) minus the v_div_scale / v_div_fixup
// range handling, which is the identity for the magnitudes met here (no operand near 2^+-1000).
// Several numerators over one denominator (a vector / its norm, cofactors / det) then cost one
// reciprocal instead of one per quotient; results are bit-identical (tests: geometry bitwise).
__device__ __forceinline__ double recip(double b) {
    double r = __builtin_amdgcn_rcp(b);
    double e = FMA(-b, r, 1.0);
    r = FMA(r, e, r);
    e = FMA(-b, r, 1.0);
    r = FMA(r, e, r);
    return r;
}
__device__ __forceinline__ double divr(double a, double b, double r) {
    const double q = a * r;
    const double d = FMA(-b, q, a);
    return FMA(d, r, q);
}

// property row of an element: ids outside [0, n_props) never index the table (row 0 is used; the batch entry points
// report them as FS_ST_BAD_PROP through their status array)
__device__ __forceinline__ int32_t safe_pid(int32_t p, int32_t n_props) {
    return (uint32_t)p < (uint32_t)n_props ? p : 0;
}

constexpr double kGP = 0.57735026918962584;  // 1.0/np.sqrt(3.0)  (MITC4.py:87)

struct QuadGeom {
    double e1[3], e2[3], e3[3];
    double xl[4][2];
    double a[5];  // a_A..a_E
    int status;
};

// MITC4.py:436-476 (_compute_local_coordinates) + :157-208 (characteristic vectors, coefficients)
__device__ __forceinline__ void quad_geometry(const double (&x)[4][3], QuadGeom &g) {
    g.status = 0;
    double v1[3], v2[3], c1[3], c2[3], n1[3], n2[3];
    for (int k = 0; k < 3; k++) { v1[k] = x[1][k] - x[0][k]; v2[k] = x[2][k] - x[0][k]; }
    cross3(v1, v2, c1);
    const double l1 = norm3(c1);
    const bool ok1 = l1 > 1e-12;
    const double rl1 = recip(l1);
    for (int k = 0; k < 3; k++) n1[k] = divr(c1[k], l1, rl1);
    for (int k = 0; k < 3; k++) { v1[k] = x[2][k] - x[0][k]; v2[k] = x[3][k] - x[0][k]; }
    cross3(v1, v2, c2);
    const double l2 = norm3(c2);
    const bool ok2 = l2 > 1e-12;
    const double rl2 = recip(l2);
    for (int k = 0; k < 3; k++) n2[k] = divr(c2[k], l2, rl2);
    double e3[3];
    for (int k = 0; k < 3; k++) {
        const double both = (n1[k] + n2[k]) / 2.0;
        e3[k] = (ok1 && ok2) ? both : (ok1 ? n1[k] : (ok2 ? n2[k] : (k == 2 ? 1.0 : 0.0)));
    }
    if (ok1 || ok2) {
        const double l = norm3(e3), rl = recip(l);
        for (int k = 0; k < 3; k++) e3[k] = divr(e3[k], l, rl);
    }
    double e1[3];
    for (int k = 0; k < 3; k++) e1[k] = x[1][k] - x[0][k];
    double d = dot3(e1, e3);
    for (int k = 0; k < 3; k++) { const double t = d * e3[k]; e1[k] = e1[k] - t; }
    if (norm3(e1) < 1e-12) {
        for (int k = 0; k < 3; k++) e1[k] = x[2][k] - x[0][k];
        d = dot3(e1, e3);
        for (int k = 0; k < 3; k++) { const double t = d * e3[k]; e1[k] = e1[k] - t; }
    }
    double l = norm3(e1), rl = recip(l);
    for (int k = 0; k < 3; k++) e1[k] = divr(e1[k], l, rl);
    double e2[3];
    cross3(e3, e1, e2);
    l = norm3(e2);
    rl = recip(l);
    for (int k = 0; k < 3; k++) e2[k] = divr(e2[k], l, rl);
    for (int k = 0; k < 3; k++) { g.e1[k] = e1[k]; g.e2[k] = e2[k]; g.e3[k] = e3[k]; }
    for (int i = 0; i < 4; i++) { g.xl[i][0] = dot3(x[i], e1); g.xl[i][1] = dot3(x[i], e2); }
    if (!(l == l) || !(g.xl[0][0] == g.xl[0][0])) g.status |= FS_ST_NAN_FRAME;

    // characteristic vectors (Ko, Lee, Bathe 2017, eqs 9-12)
    double xr[3], xs[3], xd[3], nv[3], nh[3];
    for (int k = 0; k < 3; k++) {
        xr[k] = 0.25 * (((-x[0][k] + x[1][k]) + x[2][k]) - x[3][k]);
        xs[k] = 0.25 * (((-x[0][k] - x[1][k]) + x[2][k]) + x[3][k]);
        xd[k] = 0.25 * (((x[0][k] - x[1][k]) + x[2][k]) - x[3][k]);
    }
    cross3(xr, xs, nv);
    double nn = norm3(nv);
    if (nn < 1e-12) {
        double a[3], b[3];
        for (int k = 0; k < 3; k++) { a[k] = x[1][k] - x[0][k]; b[k] = x[2][k] - x[0][k]; }
        cross3(a, b, nv);
        nn = norm3(nv);
    }
    const double rnn = recip(nn);
    for (int k = 0; k < 3; k++) nh[k] = divr(nv[k], nn, rnn);
    const double A00 = dot3(xr, xr), A01 = dot3(xr, xs), A02 = dot3(xr, nh);
    const double A10 = dot3(xs, xr), A11 = dot3(xs, xs), A12 = dot3(xs, nh);
    const double A20 = dot3(nh, xr), A21 = dot3(nh, xs), A22 = dot3(nh, nh);
    // dual vectors: np.linalg.solve(A, e_k) -> closed-form adjugate (eps-level difference only)
    const double c00 = A11 * A22 - A12 * A21, c01 = A12 * A20 - A10 * A22, c02 = A10 * A21 - A11 * A20;
    const double det = A00 * c00 + A01 * c01 + A02 * c02;
    double mr[3], ms[3];
    if (det != 0.0) {
        const double c10 = A02 * A21 - A01 * A22, c11 = A00 * A22 - A02 * A20, c12 = A01 * A20 - A00 * A21;
        const double rdet = recip(det);
        const double r0 = divr(c00, det, rdet), r1 = divr(c01, det, rdet), r2 = divr(c02, det, rdet);
        const double s0 = divr(c10, det, rdet), s1 = divr(c11, det, rdet), s2 = divr(c12, det, rdet);
        for (int k = 0; k < 3; k++) {
            mr[k] = r0 * xr[k] + r1 * xs[k] + r2 * nh[k];
            ms[k] = s0 * xr[k] + s1 * xs[k] + s2 * nh[k];
        }
    } else {
        for (int k = 0; k < 3; k++) { mr[k] = xr[k] / A00; ms[k] = xs[k] / A11; }
    }
    const double c_r = dot3(xd, mr), c_s = dot3(xd, ms);
    double dd = c_r * c_r + c_s * c_s - 1.0;
    if (__builtin_fabs(dd) < 1e-12) dd = (dd >= 0) ? 1e-12 : -1e-12;
    const double dd2 = 2.0 * dd, rdd2 = recip(dd2);
    g.a[0] = divr(c_r * (c_r - 1.0), dd2, rdd2);
    g.a[1] = divr(c_r * (c_r + 1.0), dd2, rdd2);
    g.a[2] = divr(c_s * (c_s - 1.0), dd2, rdd2);
    g.a[3] = divr(c_s * (c_s + 1.0), dd2, rdd2);
    g.a[4] = -2.0 * c_r * c_s / dd;
    if (!(g.a[0] == g.a[0])) g.status |= FS_ST_BAD_COEFF;
}

// MITC4.py:141-155
__device__ __forceinline__ void dshape(double xi, double eta, double *dxi, double *deta) {
    dxi[0] = -0.25 * (1.0 - eta); dxi[1] = +0.25 * (1.0 - eta);
    dxi[2] = +0.25 * (1.0 + eta); dxi[3] = -0.25 * (1.0 + eta);
    deta[0] = -0.25 * (1.0 - xi); deta[1] = -0.25 * (1.0 + xi);
    deta[2] = +0.25 * (1.0 + xi); deta[3] = +0.25 * (1.0 - xi);
}

// MITC4.py:614-632 (J) ; returns detJ, fills J^-1 = [[ia, ib],[ic, id]]
__device__ __forceinline__ double jacobian(const QuadGeom &g, const double *dxi, const double *deta,
                                           double &ia, double &ib, double &ic, double &id) {
    const double J00 = dots4(dxi[0], dxi[1], dxi[2], dxi[3], g.xl[0][0], g.xl[1][0], g.xl[2][0], g.xl[3][0]);
    const double J01 = dots4(dxi[0], dxi[1], dxi[2], dxi[3], g.xl[0][1], g.xl[1][1], g.xl[2][1], g.xl[3][1]);
    const double J10 = dots4(deta[0], deta[1], deta[2], deta[3], g.xl[0][0], g.xl[1][0], g.xl[2][0], g.xl[3][0]);
    const double J11 = dots4(deta[0], deta[1], deta[2], deta[3], g.xl[0][1], g.xl[1][1], g.xl[2][1], g.xl[3][1]);
    const double p = J00 * J11, q = J01 * J10;
    const double det = p - q;
    const double rdet = recip(det);
    ia = divr(J11, det, rdet); ib = divr(-J01, det, rdet); ic = divr(-J10, det, rdet); id = divr(J00, det, rdet);
    return det;
}


// Exchange inside a quad of lanes through DPP quad_perm (VALU rate) instead of __shfl_xor, which
// compiles to ds_bpermute_b32 pairs through the LDS crossbar.  Needs all four lanes of the quad active.
template <int CTRL>
__device__ __forceinline__ double quad_perm(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double quad_sum(double v) {
    v += quad_perm<0xB1>(v);   // quad_perm:[1,0,3,2] == lane ^ 1
    v += quad_perm<0x4E>(v);   // quad_perm:[2,3,0,1] == lane ^ 2
    return v;
}

// The same reductions with the four Gauss-point lanes of an element on the four 16-lane ROWS of the wave (lane = g * 16 + Q)
// through gfx950's row swaps: v_permlane32_swap exchanges the upper 32 lanes of its first operand with the lower 32 of its
// second, v_permlane16_swap the odd rows of the first with the even rows of the second, so that a + b after a swap holds the
// partial sums of a in one half (or row parity) and of b in the other -- a TRANSPOSING reduction: four values go in, and row r
// comes out with the sum over the rows of value kRowsPi[r] of its column (nine instructions for four values; the DPP
// all-reduce above takes six per value plus its hazard waits).  tools/exp/permlane_probe.hip checks the lane map on the GPU.
__device__ __forceinline__ void rows_swap32(double &a, double &b) {
    unsigned alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
    auto lo = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
    auto hi = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
    a = __hiloint2double(hi[0], lo[0]); b = __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ void rows_swap16(double &a, double &b) {
    unsigned alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
    auto lo = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
    auto hi = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
    a = __hiloint2double(hi[0], lo[0]); b = __hiloint2double(hi[1], lo[1]);
}
// row r receives the column sums of value rows_pi(r): rows 0..3 -> values 0, 2, 1, 3
__device__ __forceinline__ int rows_pi(int g) { return ((g & 1) << 1) | (g >> 1); }
__device__ __forceinline__ double rows_sum4(double v0, double v1, double v2, double v3) {
    rows_swap32(v0, v1); rows_swap32(v2, v3);
    double a = v0 + v1, b = v2 + v3;
    rows_swap16(a, b);
    return a + b;
}
// every row receives the column sum of v
__device__ __forceinline__ double rows_allsum(double v) {
    double a = v, b = v;
    rows_swap32(a, b);
    a = a + b; b = a;
    rows_swap16(a, b);
    return a + b;
}

// tying shape-derivative weights of node i: dN/dxi at A,B ; dN/deta at C,D ; dN/deta, dN/dxi at E
__device__ __forceinline__ void tying_weights(int i, double *w) {
    w[0] = (i == 2) ? 0.5 : ((i == 3) ? -0.5 : 0.0);    // A (0, 1)
    w[1] = (i == 0) ? -0.5 : ((i == 1) ? 0.5 : 0.0);    // B (0,-1)
    w[2] = (i == 1) ? -0.5 : ((i == 2) ? 0.5 : 0.0);    // C ( 1,0)
    w[3] = (i == 0) ? -0.5 : ((i == 3) ? 0.5 : 0.0);    // D (-1,0)
    w[4] = (i >= 2) ? 0.125 : -0.125;                   // E: 0.5 * dN/deta
    w[5] = (i == 1 || i == 2) ? 0.125 : -0.125;         // E: 0.5 * dN/dxi
}

__device__ __forceinline__ void membrane_W(const QuadGeom &geo, double xi, double eta, double detJ, double ia,
                                           double ib, double ic, double id, double (&W)[3][5]) {
#pragma clang fp contract(fast)
    const double aA = geo.a[0], aB = geo.a[1], aC = geo.a[2], aD = geo.a[3], aE = geo.a[4];
    const double r = xi, s = eta;
    double bl[3][5];
    bl[0][0] = 0.5 * (1.0 - 2.0 * aA + s + 2.0 * aA * s * s);
    bl[0][1] = 0.5 * (1.0 - 2.0 * aB - s + 2.0 * aB * s * s);
    bl[0][2] = aC * (-1.0 + s * s);
    bl[0][3] = aD * (-1.0 + s * s);
    bl[0][4] = aE * (-1.0 + s * s);
    bl[1][0] = aA * (-1.0 + r * r);
    bl[1][1] = aB * (-1.0 + r * r);
    bl[1][2] = 0.5 * (1.0 - 2.0 * aC + r + 2.0 * aC * r * r);
    bl[1][3] = 0.5 * (1.0 - 2.0 * aD - r + 2.0 * aD * r * r);
    bl[1][4] = aE * (-1.0 + r * r);
    bl[2][0] = 2.0 * (0.25 * (r + 4.0 * aA * r * s));
    bl[2][1] = 2.0 * (0.25 * (-r + 4.0 * aB * r * s));
    bl[2][2] = 2.0 * (0.25 * (s + 4.0 * aC * r * s));
    bl[2][3] = 2.0 * (0.25 * (-s + 4.0 * aD * r * s));
    bl[2][4] = 2.0 * (1.0 + aE * r * s);
    double Te[3][3];
    if (__builtin_fabs(detJ) < 1e-12) {
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) Te[a][b] = (a == b) ? 1.0 : 0.0;
    } else {
        Te[0][0] = ia * ia; Te[0][1] = ic * ic; Te[0][2] = ia * ic;
        Te[1][0] = ib * ib; Te[1][1] = id * id; Te[1][2] = ib * id;
        Te[2][0] = 2.0 * ia * ib; Te[2][1] = 2.0 * ic * id; Te[2][2] = ia * id + ib * ic;
    }
    for (int a = 0; a < 3; a++)
        for (int t = 0; t < 5; t++) W[a][t] = Te[a][0] * bl[0][t] + Te[a][1] * bl[1][t] + Te[a][2] * bl[2][t];
}

// Nonlinear part of the internal force of node g of an element, for the lane (element, Gauss point g) of a quad of
// lanes (all four active): T^T ( sum_q detJ h (B_L + B_NL)^T C e_GL  -  K_m u_loc ) restricted to node g, global axes --
// the terms mitc4_nl_tiled_kernel adds to K_e u_e (MITC4.py:1300-1341; K_m u_loc = Y^T S (Y u_loc) with the pre-contracted
// membrane core of the stiffness kernel).  nx, ny, H, detJ: shape derivatives, displacement gradient and Jacobian of
// the lane's Gauss point; R: rows e1, e2, e3.
__device__ __forceinline__ void quad_fnl_lane(const double (&x)[4][3], const double (&u)[4][6], const QuadGeom &geo,
                                              const fs_shell_prop *pr, int g, double xi, double eta, double detJ, double ia,
                                              double ib, double ic, double id, const double (&nx)[4], const double (&ny)[4],
                                              const double (&H)[3][2], const double (&R)[3][3], double (&fnl)[3]) {
#pragma clang fp contract(fast)
    const double h = pr->h;
    double C[3][3];
    for (int a = 0; a < 9; a++) (&C[0][0])[a] = pr->C_raw[a];
    const double E00 = 0.5 * (2.0 * H[0][0] + (H[0][0] * H[0][0] + H[1][0] * H[1][0] + H[2][0] * H[2][0]));
    const double E11 = 0.5 * (2.0 * H[1][1] + (H[0][1] * H[0][1] + H[1][1] * H[1][1] + H[2][1] * H[2][1]));
    const double E01 = 0.5 * ((H[0][1] + H[1][0]) + (H[0][0] * H[0][1] + H[1][0] * H[1][1] + H[2][0] * H[2][1]));
    const double eg[3] = {E00, E11, 2.0 * E01};
    double sg[3];
    for (int a = 0; a < 3; a++) sg[a] = C[a][0] * eg[0] + C[a][1] * eg[1] + C[a][2] * eg[2];
    const double wh = detJ * h;
    // membrane force of node g: every lane forms the four nodes' terms of ITS Gauss point; a transposing quad reduction
    // (lane ^ 1, then lane ^ 2: three exchanges instead of four full quad sums) leaves on lane g the sum over the four
    // Gauss points of node g's term
    const bool b0 = (g & 1) != 0, b1 = (g & 2) != 0;
    double fmg[3];
    for (int k = 0; k < 3; k++) {
        double v[4];
        for (int i = 0; i < 4; i++) {
            const double c0 = ((k == 0) ? nx[i] : 0.0) + H[k][0] * nx[i];
            const double c1 = ((k == 1) ? ny[i] : 0.0) + H[k][1] * ny[i];
            const double c2 = ((k == 0) ? ny[i] : ((k == 1) ? nx[i] : 0.0)) + (H[k][0] * ny[i] + H[k][1] * nx[i]);
            v[i] = wh * (c0 * sg[0] + c1 * sg[1] + c2 * sg[2]);
        }
        const double t0 = (b0 ? v[1] : v[0]) + quad_perm<0xB1>(b0 ? v[0] : v[1]);      // node b0 over lanes g, g ^ 1
        const double t1 = (b0 ? v[3] : v[2]) + quad_perm<0xB1>(b0 ? v[2] : v[3]);      // node 2 + b0
        fmg[k] = (b1 ? t1 : t0) + quad_perm<0x4E>(b1 ? t0 : t1);
    }
    // membrane core S = sum_q detJ W^T A W (symmetric 5 x 5: 15 entries)
    double W[3][5], S[15];
    membrane_W(geo, xi, eta, detJ, ia, ib, ic, id, W);
    {
        double cw[3][5];
        for (int t = 0; t < 5; t++)
            for (int a = 0; a < 3; a++)
                cw[a][t] = detJ * (pr->A[a * 3] * W[0][t] + pr->A[a * 3 + 1] * W[1][t] + pr->A[a * 3 + 2] * W[2][t]);
        int n = 0;
        for (int t = 0; t < 5; t++)
            for (int v = t; v < 5; v++) { S[n] = quad_sum(W[0][t] * cw[0][v] + W[1][t] * cw[1][v] + W[2][t] * cw[2][v]); n++; }
    }
    double gt[6][3];
    {
#pragma clang fp contract(off)
        double t3[3];
        for (int k = 0; k < 3; k++) t3[k] = 0.5 * x[2][k] + (-0.5 * x[3][k]);
        gt[0][0] = dot3(t3, geo.e1); gt[0][1] = dot3(t3, geo.e2); gt[0][2] = dot3(t3, geo.e3);
        for (int k = 0; k < 3; k++) t3[k] = -0.5 * x[0][k] + 0.5 * x[1][k];
        gt[1][0] = dot3(t3, geo.e1); gt[1][1] = dot3(t3, geo.e2); gt[1][2] = dot3(t3, geo.e3);
        for (int k = 0; k < 3; k++) t3[k] = -0.5 * x[1][k] + 0.5 * x[2][k];
        gt[2][0] = dot3(t3, geo.e1); gt[2][1] = dot3(t3, geo.e2); gt[2][2] = dot3(t3, geo.e3);
        for (int k = 0; k < 3; k++) t3[k] = -0.5 * x[0][k] + 0.5 * x[3][k];
        gt[3][0] = dot3(t3, geo.e1); gt[3][1] = dot3(t3, geo.e2); gt[3][2] = dot3(t3, geo.e3);
        for (int k = 0; k < 3; k++) t3[k] = ((-0.25 * x[0][k] + 0.25 * x[1][k]) + 0.25 * x[2][k]) + (-0.25 * x[3][k]);
        gt[4][0] = dot3(t3, geo.e1); gt[4][1] = dot3(t3, geo.e2); gt[4][2] = dot3(t3, geo.e3);
        for (int k = 0; k < 3; k++) t3[k] = ((-0.25 * x[0][k] + (-0.25 * x[1][k])) + 0.25 * x[2][k]) + 0.25 * x[3][k];
        gt[5][0] = dot3(t3, geo.e1); gt[5][1] = dot3(t3, geo.e2); gt[5][2] = dot3(t3, geo.e3);
    }
    // z = Y u_loc (all nodes: compile-time weights), Y of node g through the lane's own weights
    double z[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        double w[6];
        tying_weights(i, w);
        for (int k = 0; k < 3; k++) {
            const double ulk = R[k][0] * u[i][0] + R[k][1] * u[i][1] + R[k][2] * u[i][2];
            for (int t = 0; t < 4; t++) z[t] += (w[t] * gt[t][k]) * ulk;
            z[4] += (w[4] * gt[4][k] + w[5] * gt[5][k]) * ulk;
        }
    }
    double wg[6], Yg[5][3];
    tying_weights(g, wg);
    for (int k = 0; k < 3; k++) {
        for (int t = 0; t < 4; t++) Yg[t][k] = wg[t] * gt[t][k];
        Yg[4][k] = wg[4] * gt[4][k] + wg[5] * gt[5][k];
    }
    double Sz[5];
    for (int t = 0; t < 5; t++) {
        double acc = 0.0;
        for (int v = 0; v < 5; v++) {
            const int lo = t < v ? t : v, hi = t < v ? v : t;
            acc += S[lo * 5 - (lo * (lo - 1)) / 2 + (hi - lo)] * z[v];
        }
        Sz[t] = acc;
    }
    double fl[3];
    for (int k = 0; k < 3; k++) {
        double v = fmg[k];
        for (int t = 0; t < 5; t++) v -= Yg[t][k] * Sz[t];
        fl[k] = v;
    }
    for (int a = 0; a < 3; a++) fnl[a] = R[0][a] * fl[0] + R[1][a] * fl[1] + R[2][a] * fl[2];
}


struct TriGeom {
    double e1[3], e2[3], e3[3];
    double x2, x3, y3, area, J00, J10, J11;
    int status;
};

// MITC3.py:375-418 (_compute_local_coordinates), :427-469 (area, J), :527-558 (dH)
__device__ __forceinline__ void tri_geometry(const double (&x)[3][3], TriGeom &t) {
    double v12[3], v13[3], n[3];
    for (int k = 0; k < 3; k++) { v12[k] = x[1][k] - x[0][k]; v13[k] = x[2][k] - x[0][k]; }
    // (three quotients over one denominator: one shared reciprocal, the same bits as three divisions -- see recip / divr)
    const double L12 = norm3(v12), rL12 = recip(L12);
    for (int k = 0; k < 3; k++) t.e1[k] = divr(v12[k], L12, rL12);
    cross3(v12, v13, n);
    const double nl = norm3(n), rnl = recip(nl);
    t.status = (nl >= 1e-12) ? 0 : FS_ST_NAN_FRAME;  // ValueError("Degenerate triangle")
    for (int k = 0; k < 3; k++) t.e3[k] = divr(n[k], nl, rnl);
    cross3(t.e3, t.e1, t.e2);
    t.x2 = L12; t.x3 = dot3(v13, t.e1); t.y3 = dot3(v13, t.e2);
    const double det = t.x2 * t.y3;      // J = [[x2, 0], [x3, y3]]
    t.area = 0.5 * __builtin_fabs(det);
    const double rdet = recip(det);
    t.J00 = divr(t.y3, det, rdet); t.J10 = divr(-t.x3, det, rdet); t.J11 = divr(t.x2, det, rdet);   // J^-1, J01^-1 == 0
}


struct SolidRule { double pt[8][3]; double w[8]; int ng; };

template <int NN>
__device__ __forceinline__ void solid_shape(double xi, double eta, double zeta, double *N, double *dxi,
                                            double *deta, double *dzeta) {
    if constexpr (NN == 8) {  // SOLID.py:1768-1818
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const double sx = ((i & 3) == 1 || (i & 3) == 2) ? 1.0 : -1.0, sy = (i & 2) ? 1.0 : -1.0, sz = (i & 4) ? 1.0 : -1.0;
            const double fx = 1.0 + sx * xi, fy = 1.0 + sy * eta, fz = 1.0 + sz * zeta;
            N[i] = 0.125 * (fx * fy * fz);
            dxi[i] = 0.125 * (sx * fy * fz); deta[i] = 0.125 * (sy * fx * fz); dzeta[i] = 0.125 * (sz * fx * fy);
        }
    } else if constexpr (NN == 4) {  // SOLID.py:1581-1597
        N[0] = 1.0 - xi - eta - zeta; N[1] = xi; N[2] = eta; N[3] = zeta;
        dxi[0] = -1.0; dxi[1] = 1.0; dxi[2] = 0.0; dxi[3] = 0.0;
        deta[0] = -1.0; deta[1] = 0.0; deta[2] = 1.0; deta[3] = 0.0;
        dzeta[0] = -1.0; dzeta[1] = 0.0; dzeta[2] = 0.0; dzeta[3] = 1.0;
    } else if constexpr (NN == 6) {  // SOLID.py:756-829
        const double L[3] = {1.0 - xi - eta, xi, eta}, Lm = 0.5 * (1.0 - zeta), Lp = 0.5 * (1.0 + zeta);
        const double dLx[3] = {-1.0, 1.0, 0.0}, dLe[3] = {-1.0, 0.0, 1.0};
#pragma unroll
        for (int i = 0; i < 3; i++) {
            N[i] = L[i] * Lm; N[3 + i] = L[i] * Lp;
            dxi[i] = dLx[i] * Lm; dxi[3 + i] = dLx[i] * Lp;
            deta[i] = dLe[i] * Lm; deta[3 + i] = dLe[i] * Lp;
            dzeta[i] = L[i] * -0.5; dzeta[3 + i] = L[i] * 0.5;
        }
    } else {  // PYRAMID5, SOLID.py:1129-1218
        static_assert(NN == 5, "unsupported solid");
        double r = 1.0 - zeta;
        r = (r < 1e-14) ? 1e-14 : r;
        const double r2 = r * r, rmx = r - xi, rpx = r + xi, rme = r - eta, rpe = r + eta;
        N[0] = 0.25 * rmx * rme / r; N[1] = 0.25 * rpx * rme / r; N[2] = 0.25 * rpx * rpe / r; N[3] = 0.25 * rmx * rpe / r; N[4] = zeta;
        dxi[0] = -0.25 * rme / r; dxi[1] = 0.25 * rme / r; dxi[2] = 0.25 * rpe / r; dxi[3] = -0.25 * rpe / r; dxi[4] = 0.0;
        deta[0] = -0.25 * rmx / r; deta[1] = -0.25 * rpx / r; deta[2] = 0.25 * rpx / r; deta[3] = 0.25 * rmx / r; deta[4] = 0.0;
        dzeta[0] = -0.25 * (rme + rmx) / r + 0.25 * rmx * rme / r2;
        dzeta[1] = -0.25 * (rme + rpx) / r + 0.25 * rpx * rme / r2;
        dzeta[2] = -0.25 * (rpe + rpx) / r + 0.25 * rpx * rpe / r2;
        dzeta[3] = -0.25 * (rpe + rmx) / r + 0.25 * rmx * rpe / r2;
        dzeta[4] = 1.0;
    }
}


// Integration rule of a linear solid as the reference builds it (SOLID.py:731-754 WEDGE6 -- weights sum
// to 2, kept; :1099-1127 PYRAMID5; :1575-1579 TETRA4; :1744-1766 HEXA8, reduced = one point, weight 8).
inline void make_solid_rule(int nn, int reduced, SolidRule &rule) {
    memset(&rule, 0, sizeof rule);
    const double gp = 0.57735026918962584;   // 1/np.sqrt(3)
    if (nn == 8) {
        if (reduced) { rule.ng = 1; rule.w[0] = 8.0; }
        else {
            rule.ng = 8;
            for (int q = 0; q < 8; q++) {
                rule.pt[q][0] = (q & 1) ? gp : -gp; rule.pt[q][1] = (q & 2) ? gp : -gp; rule.pt[q][2] = (q & 4) ? gp : -gp;
                rule.w[q] = 1.0;
            }
        }
    } else if (nn == 4) {
        rule.ng = 1; rule.pt[0][0] = rule.pt[0][1] = rule.pt[0][2] = 0.25; rule.w[0] = 1.0 / 6;
    } else if (nn == 6) {
        const double tp[3][2] = {{1.0 / 6, 1.0 / 6}, {2.0 / 3, 1.0 / 6}, {1.0 / 6, 2.0 / 3}};
        rule.ng = 6;
        for (int q = 0; q < 6; q++) {
            rule.pt[q][0] = tp[q / 2][0]; rule.pt[q][1] = tp[q / 2][1]; rule.pt[q][2] = (q & 1) ? gp : -gp;
            rule.w[q] = (1.0 / 6) * 1.0 * 2;
        }
    } else {
        const double a = 0.26318405556971359557, za = 0.54415184401122528880, wa = 0.10078588207982543059;
        const double b = 0.50661630334978742377, zb = 0.12251482265544137787, wb = 0.23254745125350790274;
        const double sx[4] = {-1, 1, 1, -1}, sy[4] = {-1, -1, 1, 1};
        rule.ng = 8;
        for (int q = 0; q < 4; q++) {
            rule.pt[q][0] = sx[q] * a; rule.pt[q][1] = sy[q] * a; rule.pt[q][2] = za; rule.w[q] = wa;
            rule.pt[4 + q][0] = sx[q] * b; rule.pt[4 + q][1] = sy[q] * b; rule.pt[4 + q][2] = zb; rule.w[4 + q] = wb;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Quadratic solids TETRA10 / PYRAMID13 / WEDGE15 / HEXA20 (SOLID.py:832-1059, 1221-1537, 1600-1708, 1821-1946).  Their shape
// data at the integration points do not depend on the element, so they are tabulated once on the HOST
// with the reference's own operation order (this header is compiled with contraction off; `x**2` of a
// NumPy float64 scalar is libm pow(x, 2.0)) and handed to the kernels as a device table.  This matters
// for HEXA20, whose derivatives are FORWARD DIFFERENCES with h = 1e-8 (:1934-1946): the difference of
// two rounded N values divided by h carries ~1e-8 of rounding noise, which only a bit-identical N
// reproduces (tests/golden/solid2_tables.npz pins the tables bit for bit against the reference).
// ------------------------------------------------------------------------------------------
constexpr int FS_SOLID2_MAX_NG = 27, FS_SOLID2_MAX_NN = 20;
struct SolidTable {
    int nn, ng;
    double pt[FS_SOLID2_MAX_NG][3], w[FS_SOLID2_MAX_NG];
    double v[FS_SOLID2_MAX_NG][FS_SOLID2_MAX_NN][4];   // N, dN/dxi, dN/deta, dN/dzeta
};

inline void hexa20_N(double xi, double eta, double zeta, double *N) {   // SOLID.py:1879-1932
    static const double c[8][3] = {{-1, -1, -1}, {1, -1, -1}, {1, 1, -1}, {-1, 1, -1}, {-1, -1, 1}, {1, -1, 1}, {1, 1, 1}, {-1, 1, 1}};
    for (int i = 0; i < 8; i++) {
        const double a = xi * c[i][0], b = eta * c[i][1], d = zeta * c[i][2];
        N[i] = (((0.125 * (1 + a)) * (1 + b)) * (1 + d)) * (((a + b) + d) - 2);
    }
    const double x2 = 1 - pow(xi, 2.0), e2 = 1 - pow(eta, 2.0), z2 = 1 - pow(zeta, 2.0);
    N[8] = ((0.25 * x2) * (1 - eta)) * (1 - zeta);
    N[9] = ((0.25 * (1 - xi)) * e2) * (1 - zeta);
    N[10] = ((0.25 * (1 - xi)) * (1 - eta)) * z2;
    N[11] = ((0.25 * (1 + xi)) * e2) * (1 - zeta);
    N[12] = ((0.25 * (1 + xi)) * (1 - eta)) * z2;
    N[13] = ((0.25 * x2) * (1 + eta)) * (1 - zeta);
    N[14] = ((0.25 * (1 + xi)) * (1 + eta)) * z2;
    N[15] = ((0.25 * (1 - xi)) * (1 + eta)) * z2;
    N[16] = ((0.25 * x2) * (1 - eta)) * (1 + zeta);
    N[17] = ((0.25 * (1 - xi)) * e2) * (1 + zeta);
    N[18] = ((0.25 * (1 + xi)) * e2) * (1 + zeta);
    N[19] = ((0.25 * x2) * (1 + eta)) * (1 + zeta);
}

inline int make_solid_table(int nn, SolidTable &t) {
    memset(&t, 0, sizeof t);
    t.nn = nn;
    if (nn == 20) {          // 3x3x3 Gauss, order k-j-i, weights wi*wj*wk (SOLID.py:1862-1877)
        const double s35 = sqrt(3.0 / 5.0), p1[3] = {-s35, 0.0, s35}, w1[3] = {5.0 / 9, 8.0 / 9, 5.0 / 9};
        t.ng = 27;
        int n = 0;
        for (int k = 0; k < 3; k++)
            for (int j = 0; j < 3; j++)
                for (int i = 0; i < 3; i++) {
                    t.pt[n][0] = p1[i]; t.pt[n][1] = p1[j]; t.pt[n][2] = p1[k];
                    t.w[n++] = (w1[i] * w1[j]) * w1[k];
                }
        const double h = 1e-8;
        for (int q = 0; q < 27; q++) {
            double N0[20], Na[20], Nb[20], Nc[20];
            const double xi = t.pt[q][0], eta = t.pt[q][1], zeta = t.pt[q][2];
            hexa20_N(xi, eta, zeta, N0);
            hexa20_N(xi + h, eta, zeta, Na);
            hexa20_N(xi, eta + h, zeta, Nb);
            hexa20_N(xi, eta, zeta + h, Nc);
            for (int i = 0; i < 20; i++) {
                t.v[q][i][0] = N0[i]; t.v[q][i][1] = (Na[i] - N0[i]) / h;
                t.v[q][i][2] = (Nb[i] - N0[i]) / h; t.v[q][i][3] = (Nc[i] - N0[i]) / h;
            }
        }
        return 0;
    }
    if (nn == 13) {          // PYRAMID13, SOLID.py:1221-1537: 18-point Felippa rule, rational serendipity functions
        const double ua = 0.35309846330877704481, za = 0.54415184401122528880;
        const double ub = 0.67969709567986745790, zb = 0.12251482265544137787;
        const double wa[3] = {0.023330065296255886709, 0.037328104474009418735, 0.059724967158415069975};
        const double wb[3] = {0.053830428530904607120, 0.086128685649447371390, 0.137805897039115794220};
        t.ng = 18;
        for (int layer = 0; layer < 2; layer++)
            for (int j = 0; j < 3; j++)
                for (int i = 0; i < 3; i++) {
                    const int q = 9 * layer + 3 * j + i;
                    const double u = layer ? ub : ua;
                    t.pt[q][0] = (i - 1) * u; t.pt[q][1] = (j - 1) * u; t.pt[q][2] = layer ? zb : za;
                    const int kind = (i == 1) + (j == 1);          // corner, edge midpoint, centre of the 3 x 3 grid
                    t.w[q] = layer ? wb[kind] : wa[kind];
                }
        for (int q = 0; q < 18; q++) {
            const double xi = t.pt[q][0], eta = t.pt[q][1], zeta = t.pt[q][2];
            double r = 1 - zeta;
            if (r < 1e-14) r = 1e-14;
            const double r2 = r * r;
            double xs = xi / r, es = eta / r;
            xs = xs < -1.0 ? -1.0 : (xs > 1.0 ? 1.0 : xs);
            es = es < -1.0 ? -1.0 : (es > 1.0 ? 1.0 : es);
            const double xs2 = pow(xs, 2.0), es2 = pow(es, 2.0);
            const double f[4] = {((0.25 * (1 - xs)) * (1 - es)) * ((-xs - es) - 1), ((0.25 * (1 + xs)) * (1 - es)) * ((xs - es) - 1),
                                 ((0.25 * (1 + xs)) * (1 + es)) * ((xs + es) - 1), ((0.25 * (1 - xs)) * (1 + es)) * ((-xs + es) - 1)};
            const double dfx[4] = {(0.25 * (1 - es)) * (2 * xs + es), (0.25 * (1 - es)) * (2 * xs - es),
                                   (0.25 * (1 + es)) * (2 * xs + es), (0.25 * (1 + es)) * (2 * xs - es)};
            const double dfe[4] = {(0.25 * (1 - xs)) * (xs + 2 * es), (0.25 * (1 + xs)) * (-xs + 2 * es),
                                   (0.25 * (1 + xs)) * (xs + 2 * es), (0.25 * (1 - xs)) * (-xs + 2 * es)};
            // base edge midpoints 5 (0-1), 6 (0-3), 8 (1-2), 10 (2-3)
            const double gv[4] = {(0.5 * (1 - xs2)) * (1 - es), (0.5 * (1 - xs)) * (1 - es2), (0.5 * (1 + xs)) * (1 - es2),
                                  (0.5 * (1 - xs2)) * (1 + es)};
            const double dgx[4] = {-xs * (1 - es), -0.5 * (1 - es2), 0.5 * (1 - es2), -xs * (1 + es)};
            const double dge[4] = {-0.5 * (1 - xs2), -(1 - xs) * es, -(1 + xs) * es, 0.5 * (1 - xs2)};
            const double ph[4] = {(0.25 * (1 - xs)) * (1 - es), (0.25 * (1 + xs)) * (1 - es), (0.25 * (1 + xs)) * (1 + es),
                                  (0.25 * (1 - xs)) * (1 + es)};
            const double dpx[4] = {-0.25 * (1 - es), 0.25 * (1 - es), 0.25 * (1 + es), -0.25 * (1 + es)};
            const double dpe[4] = {-0.25 * (1 - xs), -0.25 * (1 + xs), 0.25 * (1 + xs), 0.25 * (1 - xs)};
            const double one2z = 1 - 2 * zeta;
            const double h_base = r * one2z, h_lat = (4 * zeta) * r, dh_base = -3 + 4 * zeta, dh_lat = 4 * one2z;
            const double lat = (4 * zeta) * (1 - zeta);
            double (*v)[4] = t.v[q];
            auto put = [&](int n, double N, double fn, double a, double b, double h, double dh) {
                v[n][0] = N; v[n][1] = (a * h) / r; v[n][2] = (b * h) / r;
                v[n][3] = ((a * xi + b * eta) / r2) * h + fn * dh;
            };
            for (int i = 0; i < 4; i++) put(i, (f[i] * r) * one2z, f[i], dfx[i], dfe[i], h_base, dh_base);
            v[4][0] = zeta * (2 * zeta - 1); v[4][1] = 0.0; v[4][2] = 0.0; v[4][3] = 4 * zeta - 1;
            const int gn[4] = {5, 6, 8, 10}, ln[4] = {7, 9, 11, 12};
            for (int i = 0; i < 4; i++) put(gn[i], (gv[i] * r) * one2z, gv[i], dgx[i], dge[i], h_base, dh_base);
            for (int i = 0; i < 4; i++) put(ln[i], ph[i] * lat, ph[i], dpx[i], dpe[i], h_lat, dh_lat);
        }
        return 0;
    }
    if (nn == 10) {          // SOLID.py:1642-1708
        const double a = (5 - sqrt(5.0)) / 20, b = (5 + 3 * sqrt(5.0)) / 20;
        const double P[4][3] = {{a, a, a}, {b, a, a}, {a, b, a}, {a, a, b}};
        t.ng = 4;
        for (int q = 0; q < 4; q++) {
            for (int k = 0; k < 3; k++) t.pt[q][k] = P[q][k];
            t.w[q] = 1.0 / 24;
            const double xi = P[q][0], eta = P[q][1], zeta = P[q][2];
            const double L[4] = {((1 - xi) - eta) - zeta, xi, eta, zeta};
            const double dL[4][3] = {{-1, -1, -1}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
            double (*v)[4] = t.v[q];
            for (int i = 0; i < 4; i++) {
                v[i][0] = L[i] * (2 * L[i] - 1);
                for (int k = 0; k < 3; k++) v[i][1 + k] = (4 * L[i] - 1) * dL[i][k];
            }
            // edges (Gmsh): 4: 0-1, 5: 1-2, 6: 0-2, 7: 0-3, 8: 2-3, 9: 1-3; N = 4 La Lb as written in the
            // reference, dN = 4 (Lp dLq + Lq dLp) with its operand order
            const int na[6] = {0, 1, 2, 0, 2, 1}, nb[6] = {1, 2, 0, 3, 3, 3};       // N = 4 * L[na] * L[nb]
            const int da[6] = {1, 2, 0, 3, 3, 3}, db[6] = {0, 1, 2, 0, 2, 1};       // dN = 4 (L[da] dL[db] + L[db] dL[da])
            for (int m = 0; m < 6; m++) {
                v[4 + m][0] = (4 * L[na[m]]) * L[nb[m]];
                for (int k = 0; k < 3; k++) v[4 + m][1 + k] = 4 * (L[da[m]] * dL[db[m]][k] + L[db[m]] * dL[da[m]][k]);
            }
        }
        return 0;
    }
    if (nn == 15) {          // SOLID.py:880-1059: 7-point triangle x 3-point Gauss, weight tri_w lin_w 2
        const double tp[7][2] = {{1.0 / 3, 1.0 / 3}, {0.797426985353087, 0.101286507323456}, {0.101286507323456, 0.797426985353087},
                                 {0.101286507323456, 0.101286507323456}, {0.470142064105115, 0.059715871789770},
                                 {0.470142064105115, 0.470142064105115}, {0.059715871789770, 0.470142064105115}};
        const double tw[7] = {0.225, 0.125939180544827, 0.125939180544827, 0.125939180544827, 0.132394152788506,
                              0.132394152788506, 0.132394152788506};
        const double s35 = sqrt(3.0 / 5.0), lp[3] = {-s35, 0.0, s35}, lw[3] = {5.0 / 9, 8.0 / 9, 5.0 / 9};
        t.ng = 21;
        int n = 0;
        for (int i = 0; i < 7; i++)
            for (int j = 0; j < 3; j++) {
                const double xi = tp[i][0], eta = tp[i][1], zeta = lp[j];
                t.pt[n][0] = xi; t.pt[n][1] = eta; t.pt[n][2] = zeta;
                t.w[n] = ((tw[i] * 0.5) * lw[j]) * 2;
                const double L1 = (1 - xi) - eta, L2 = xi, L3 = eta;
                const double Lm = (0.5 * zeta) * (zeta - 1), L0 = 1 - pow(zeta, 2.0), Lp = (0.5 * zeta) * (zeta + 1);
                const double dLm = zeta - 0.5, dL0 = -2 * zeta, dLp = zeta + 0.5;
                double (*v)[4] = t.v[n];
                const double Lz[2] = {Lm, Lp}, dLz[2] = {dLm, dLp};
                for (int c = 0; c < 2; c++) {       // bottom (c = 0) / top (c = 1) triangles
                    const int o = 3 * c;
                    v[o + 0][0] = (L1 * (2 * L1 - 1)) * Lz[c]; v[o + 1][0] = (L2 * (2 * L2 - 1)) * Lz[c]; v[o + 2][0] = (L3 * (2 * L3 - 1)) * Lz[c];
                    v[o + 0][1] = ((4 * L1 - 1) * -1) * Lz[c]; v[o + 0][2] = ((4 * L1 - 1) * -1) * Lz[c]; v[o + 0][3] = (L1 * (2 * L1 - 1)) * dLz[c];
                    v[o + 1][1] = (4 * L2 - 1) * Lz[c]; v[o + 1][2] = 0; v[o + 1][3] = (L2 * (2 * L2 - 1)) * dLz[c];
                    v[o + 2][1] = 0; v[o + 2][2] = (4 * L3 - 1) * Lz[c]; v[o + 2][3] = (L3 * (2 * L3 - 1)) * dLz[c];
                    const int e01 = c ? 12 : 6, e02 = c ? 13 : 7, e12 = c ? 14 : 9;
                    v[e01][0] = ((4 * L1) * L2) * Lz[c]; v[e01][1] = (4 * (L1 - L2)) * Lz[c]; v[e01][2] = (-4 * L2) * Lz[c]; v[e01][3] = ((4 * L1) * L2) * dLz[c];
                    v[e02][0] = ((4 * L1) * L3) * Lz[c]; v[e02][1] = (-4 * L3) * Lz[c]; v[e02][2] = (4 * (L1 - L3)) * Lz[c]; v[e02][3] = ((4 * L1) * L3) * dLz[c];
                    v[e12][0] = ((4 * L2) * L3) * Lz[c]; v[e12][1] = (4 * L3) * Lz[c]; v[e12][2] = (4 * L2) * Lz[c]; v[e12][3] = ((4 * L2) * L3) * dLz[c];
                }
                v[8][0] = L1 * L0; v[8][1] = -L0; v[8][2] = -L0; v[8][3] = L1 * dL0;
                v[10][0] = L2 * L0; v[10][1] = L0; v[10][2] = 0; v[10][3] = L2 * dL0;
                v[11][0] = L3 * L0; v[11][1] = 0; v[11][2] = L0; v[11][3] = L3 * dL0;
                n++;
            }
        return 0;
    }
    return -1;
}

}  // namespace
