// femshell_tri.h -- MITC3+ device code shared by femshell_hip.hip (element kernels) and femshell_fused_tri.hip (fused
// evaluate-and-assemble): Hammer / tying points, shape and assumed-shear data of one function (tri_fn), and the 6 x 6
// node-pair block formed from an LDS image of that data (tri_pair_block).  Citations: /root/reference/src/fem_shell/.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "femshell_hip.h"
#include "femshell_geom.h"

constexpr double kTriR[3] = {1.0 / 6.0, 2.0 / 3.0, 1.0 / 6.0};
constexpr double kTriS[3] = {1.0 / 6.0, 1.0 / 6.0, 2.0 / 3.0};
constexpr double kTyD = 1.0e-4;
// tying points A,B,C,D,E,F (MITC3.py:897-905)
constexpr double kTyR[6] = {1.0 / 6.0, 2.0 / 3.0, 1.0 / 6.0, 1.0 / 3.0 + kTyD, 1.0 / 3.0 - 2.0 * kTyD, 1.0 / 3.0 + kTyD};
constexpr double kTyS[6] = {2.0 / 3.0, 1.0 / 6.0, 1.0 / 6.0, 1.0 / 3.0 - 2.0 * kTyD, 1.0 / 3.0 + kTyD, 1.0 / 3.0 + kTyD};

struct TriFn {          // shape data of one function (node 0..2 or bubble 3) at the 3 Hammer points
    double fx[3], fy[3];    // d f / dx, d f / dy (enriched)            MITC3.py:853-887
    double gx[3][2];        // assumed shear column of theta_x          MITC3.py:796-933
    double gy[3][2];        // assumed shear column of theta_y
    double h[3];            // linear shape value (drilling row)
    double dhx, dhy;        // linear derivatives (w column of the shear, membrane)
};

__device__ __forceinline__ double tri_f(int i, double r, double s) {
    const double f4 = 27.0 * r * s * (1.0 - r - s);
    const double h = (i == 0) ? (1.0 - r - s) : ((i == 1) ? r : s);
    return (i == 3) ? f4 : h - f4 / 3.0;
}

__device__ __forceinline__ void tri_fn(const TriGeom &t, int i, TriFn &F) {
#pragma clang fp contract(fast)
    const double dr = (i == 0) ? -1.0 : ((i == 1) ? 1.0 : 0.0);
    const double ds = (i == 0) ? -1.0 : ((i == 2) ? 1.0 : 0.0);
    F.dhx = t.J00 * dr;
    F.dhy = t.J10 * dr + t.J11 * ds;
    double f[6];
#pragma unroll
    for (int p = 0; p < 6; p++) f[p] = tri_f(i, kTyR[p], kTyS[p]);
    const double fA = f[0], fB = f[1], fC = f[2], fD = f[3], fE = f[4], fF = f[5];
    // theta_y: covariant strains f(P)*(x2, x3); theta_x: f(P)*(0, -y3)   (g_r = (x2,0), g_s = (x3,y3))
    const double yr = t.x2, ys = t.x3, xs = -t.y3;
    const double ertc_y = (2.0 / 3.0) * (fB * yr - 0.5 * (fB * ys)) + (1.0 / 3.0) * (fC * yr + fC * ys);
    const double estc_y = (2.0 / 3.0) * (fA * ys - 0.5 * (fA * yr)) + (1.0 / 3.0) * (fC * yr + fC * ys);
    const double chat_y = (fF * yr - fD * yr) - (fF * ys - fE * ys);
    const double ertc_x = (2.0 / 3.0) * (-0.5 * (fB * xs)) + (1.0 / 3.0) * (fC * xs);
    const double estc_x = (2.0 / 3.0) * (fA * xs) + (1.0 / 3.0) * (fC * xs);
    const double chat_x = -(fF * xs - fE * xs);
#pragma unroll
    for (int g = 0; g < 3; g++) {
        const double r = kTriR[g], s = kTriS[g];
        const double df4r = 27.0 * s * (1.0 - 2.0 * r - s), df4s = 27.0 * r * (1.0 - r - 2.0 * s);
        const double fr = (i == 3) ? df4r : dr - df4r / 3.0;
        const double fs = (i == 3) ? df4s : ds - df4s / 3.0;
        F.fx[g] = t.J00 * fr;
        F.fy[g] = t.J10 * fr + t.J11 * fs;
        const double ls = (1.0 / 3.0) * (3.0 * s - 1.0), lr = (1.0 / 3.0) * (1.0 - 3.0 * r);
        const double ert_y = ertc_y + chat_y * ls, est_y = estc_y + chat_y * lr;
        const double ert_x = ertc_x + chat_x * ls, est_x = estc_x + chat_x * lr;
        F.gy[g][0] = t.J00 * ert_y; F.gy[g][1] = t.J10 * ert_y + t.J11 * est_y;
        F.gx[g][0] = t.J00 * ert_x; F.gx[g][1] = t.J10 * ert_x + t.J11 * est_x;
        F.h[g] = (i == 0) ? (1.0 - r - s) : ((i == 1) ? r : s);
    }
}

// C5 * S for the three generalised-strain columns (w, theta_x, theta_y) of one function at one GP.
// S = (kxx, kyy, kxy, gxz, gyz);  w: (0,0,0,dhx,dhy)  tx: (0,-fy,-fx,gx)  ty: (fx,0,fy,gy)
struct TriCS { double w[2], x[5], y[5]; };
__device__ __forceinline__ void tri_cs(const double *D, const double *Cs, const TriFn &F, int g, bool nodal, TriCS &c) {
#pragma clang fp contract(fast)
    const double fx = F.fx[g], fy = F.fy[g];
    c.x[0] = -(D[1] * fy + D[2] * fx); c.x[1] = -(D[4] * fy + D[5] * fx); c.x[2] = -(D[7] * fy + D[8] * fx);
    c.y[0] = D[0] * fx + D[2] * fy;    c.y[1] = D[3] * fx + D[5] * fy;    c.y[2] = D[6] * fx + D[8] * fy;
    c.x[3] = Cs[0] * F.gx[g][0] + Cs[1] * F.gx[g][1]; c.x[4] = Cs[2] * F.gx[g][0] + Cs[3] * F.gx[g][1];
    c.y[3] = Cs[0] * F.gy[g][0] + Cs[1] * F.gy[g][1]; c.y[4] = Cs[2] * F.gy[g][0] + Cs[3] * F.gy[g][1];
    c.w[0] = nodal ? Cs[0] * F.dhx + Cs[1] * F.dhy : 0.0;
    c.w[1] = nodal ? Cs[2] * F.dhx + Cs[3] * F.dhy : 0.0;
}
__device__ __forceinline__ double tri_dot_w(const TriFn &F, const double *c5) {
#pragma clang fp contract(fast)
    return F.dhx * c5[3] + F.dhy * c5[4];
}
__device__ __forceinline__ double tri_dot_x(const TriFn &F, int g, const double *c5) {
#pragma clang fp contract(fast)
    return -F.fy[g] * c5[1] - F.fx[g] * c5[2] + F.gx[g][0] * c5[3] + F.gx[g][1] * c5[4];
}
__device__ __forceinline__ double tri_dot_y(const TriFn &F, int g, const double *c5) {
#pragma clang fp contract(fast)
    return F.fx[g] * c5[0] + F.fy[g] * c5[2] + F.gy[g][0] * c5[3] + F.gy[g][1] * c5[4];
}

// G = blockwise R^T L R of a 6x6 node-pair block (T of MITC3.py:471-497 is block-diagonal in R)
__device__ __forceinline__ void rotate_block6(const double (&L)[6][6], const double (&R)[3][3], double (&G)[6][6]) {
#pragma clang fp contract(fast)
#pragma unroll
    for (int bi = 0; bi < 2; bi++)
#pragma unroll
        for (int bj = 0; bj < 2; bj++) {
            double Y[3][3];
#pragma unroll
            for (int a = 0; a < 3; a++)
#pragma unroll
                for (int b = 0; b < 3; b++) {
                    double s = 0.0;
#pragma unroll
                    for (int k = 0; k < 3; k++) s += R[k][a] * L[3 * bi + k][3 * bj + b];
                    Y[a][b] = s;
                }
#pragma unroll
            for (int a = 0; a < 3; a++)
#pragma unroll
                for (int b = 0; b < 3; b++) {
                    double s = 0.0;
#pragma unroll
                    for (int k = 0; k < 3; k++) s += Y[a][k] * R[k][b];
                    G[3 * bi + a][3 * bj + b] = s;
                }
        }
}


struct TriG { double fx, fy, gx0, gx1, gy0, gy1, h; };
// shape data of function qf = slot * 4 + f at Hammer point g from the LDS image fn[component < 23][FS entries]
template <int FS>
__device__ __forceinline__ TriG tri_g_load(const double *fn, int qf, int g) {
    TriG F;
    F.fx = fn[(0 + g) * FS + qf]; F.fy = fn[(3 + g) * FS + qf];
    F.gx0 = fn[(6 + 2 * g) * FS + qf]; F.gx1 = fn[(7 + 2 * g) * FS + qf];
    F.gy0 = fn[(12 + 2 * g) * FS + qf]; F.gy1 = fn[(13 + 2 * g) * FS + qf];
    F.h = fn[(18 + g) * FS + qf];
    return F;
}
struct TriCSg { double w[2], x[5], y[5]; };
__device__ __forceinline__ void tri_cs_g(const double *D, const double *Cs, const TriG &F, double dhx, double dhy,
                                         bool nodal, TriCSg &c) {
#pragma clang fp contract(fast)
    const double fx = F.fx, fy = F.fy;
    c.x[0] = -(D[1] * fy + D[2] * fx); c.x[1] = -(D[4] * fy + D[5] * fx); c.x[2] = -(D[7] * fy + D[8] * fx);
    c.y[0] = D[0] * fx + D[2] * fy;    c.y[1] = D[3] * fx + D[5] * fy;    c.y[2] = D[6] * fx + D[8] * fy;
    c.x[3] = Cs[0] * F.gx0 + Cs[1] * F.gx1; c.x[4] = Cs[2] * F.gx0 + Cs[3] * F.gx1;
    c.y[3] = Cs[0] * F.gy0 + Cs[1] * F.gy1; c.y[4] = Cs[2] * F.gy0 + Cs[3] * F.gy1;
    c.w[0] = nodal ? Cs[0] * dhx + Cs[1] * dhy : 0.0;
    c.w[1] = nodal ? Cs[2] * dhx + Cs[3] * dhy : 0.0;
}
__device__ __forceinline__ double tri_dw(double dhx, double dhy, const double *c5) {
#pragma clang fp contract(fast)
    return dhx * c5[3] + dhy * c5[4];
}
__device__ __forceinline__ double tri_dx(const TriG &F, const double *c5) {
#pragma clang fp contract(fast)
    return -F.fy * c5[1] - F.fx * c5[2] + F.gx0 * c5[3] + F.gx1 * c5[4];
}
__device__ __forceinline__ double tri_dy(const TriG &F, const double *c5) {
#pragma clang fp contract(fast)
    return F.fx * c5[0] + F.fy * c5[2] + F.gy0 * c5[3] + F.gy1 * c5[4];
}

// 6 x 6 block (ti <= tj) of MITC3+ element `qq` of an LDS image (MITC3.py:935-1103, composite coupling
// MITC3_composite.py:204-240): fn[component < 23][FS] holds the shape / tying data of the functions qq * 4 + {0, 1, 2,
// bubble} at the 3 Hammer points (tri_fn; components 21, 22 = linear derivatives), geo[component < 10][GS] the area and
// the frame rows.  Gauss point by Gauss point (only one point's data is live), bubble condensation, membrane CST,
// drilling, rotation to global axes; diagonal blocks symmetrised.  Shared by mitc3_k_lds_kernel and the fused kernel
// (femshell_fused_tri.hip), so both form the same bits.
// Function f of element qq sits at entry qq * QS + f * FST of a component (element kernel: QS = 4, FST = 1; the fused
// kernel keeps the functions apart, QS = 1, FST = slots + pad, so that lanes reading one function of neighbouring
// elements hit neighbouring banks).
// PROP: fs_shell_prop or any type with its members A, B, D, Cs, k_drill, flags (a register copy of the row).
template <int FS, int GS, int QS, int FST, class PROP>
__device__ __forceinline__ void tri_pair_block(const double *fn, const double *geo, int qq, int ti, int tj,
                                               const PROP *pr, double (&G)[6][6]) {
#pragma clang fp contract(fast)
    const int qi = qq * QS + ti * FST, qj = qq * QS + tj * FST, qb = qq * QS + 3 * FST;
    const double area = geo[qq];
    const double idhx = fn[21 * FS + qi], idhy = fn[22 * FS + qi];
    const double jdhx = fn[21 * FS + qj], jdhy = fn[22 * FS + qj];
    double D[9], Cs[4];
    for (int c = 0; c < 9; c++) D[c] = pr->D[c];
    for (int c = 0; c < 4; c++) Cs[c] = pr->Cs[c];
    const double kd = pr->k_drill;
    const bool coupled = (pr->flags & FS_PROP_COUPLED) != 0;
    double Kij[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    double Ni[3][2] = {{0, 0}, {0, 0}, {0, 0}}, Nj[3][2] = {{0, 0}, {0, 0}, {0, 0}};
    double Kbb[2][2] = {{0, 0}, {0, 0}};
    double dr[2][2] = {{0, 0}, {0, 0}}, drz[2] = {0, 0}, dzr[2] = {0, 0}, dzz = 0.0;
    double cpl_ij[2][2] = {{0, 0}, {0, 0}}, cpl_ji[2][2] = {{0, 0}, {0, 0}};
    const double bdi[2] = {-0.5 * idhy, 0.5 * idhx}, bdj[2] = {-0.5 * jdhy, 0.5 * jdhx};
    const double wA = (1.0 / 3.0) * area;
#pragma unroll
    for (int g = 0; g < 3; g++) {
        const TriG Fi = tri_g_load<FS>(fn, qi, g), Fj = tri_g_load<FS>(fn, qj, g), Fb = tri_g_load<FS>(fn, qb, g);
        TriCSg cj, cb;
        tri_cs_g(D, Cs, Fj, jdhx, jdhy, true, cj);
        tri_cs_g(D, Cs, Fb, 0.0, 0.0, false, cb);
        const double cjw[5] = {0.0, 0.0, 0.0, cj.w[0], cj.w[1]};
        Kij[0][0] += wA * tri_dw(idhx, idhy, cjw);  Kij[0][1] += wA * tri_dw(idhx, idhy, cj.x);  Kij[0][2] += wA * tri_dw(idhx, idhy, cj.y);
        Kij[1][0] += wA * (Fi.gx0 * cj.w[0] + Fi.gx1 * cj.w[1]);
        Kij[1][1] += wA * tri_dx(Fi, cj.x);  Kij[1][2] += wA * tri_dx(Fi, cj.y);
        Kij[2][0] += wA * (Fi.gy0 * cj.w[0] + Fi.gy1 * cj.w[1]);
        Kij[2][1] += wA * tri_dy(Fi, cj.x);  Kij[2][2] += wA * tri_dy(Fi, cj.y);
        Ni[0][0] += wA * tri_dw(idhx, idhy, cb.x);  Ni[0][1] += wA * tri_dw(idhx, idhy, cb.y);
        Ni[1][0] += wA * tri_dx(Fi, cb.x);  Ni[1][1] += wA * tri_dx(Fi, cb.y);
        Ni[2][0] += wA * tri_dy(Fi, cb.x);  Ni[2][1] += wA * tri_dy(Fi, cb.y);
        Nj[0][0] += wA * tri_dw(jdhx, jdhy, cb.x);  Nj[0][1] += wA * tri_dw(jdhx, jdhy, cb.y);
        Nj[1][0] += wA * tri_dx(Fj, cb.x);  Nj[1][1] += wA * tri_dx(Fj, cb.y);
        Nj[2][0] += wA * tri_dy(Fj, cb.x);  Nj[2][1] += wA * tri_dy(Fj, cb.y);
        Kbb[0][0] += wA * tri_dx(Fb, cb.x); Kbb[0][1] += wA * tri_dx(Fb, cb.y);
        Kbb[1][0] += wA * tri_dy(Fb, cb.x); Kbb[1][1] += wA * tri_dy(Fb, cb.y);
        const double wk = wA * kd;
        dr[0][0] += wk * bdi[0] * bdj[0]; dr[0][1] += wk * bdi[0] * bdj[1];
        dr[1][0] += wk * bdi[1] * bdj[0]; dr[1][1] += wk * bdi[1] * bdj[1];
        drz[0] += wk * bdi[0] * (-Fj.h); drz[1] += wk * bdi[1] * (-Fj.h);
        dzr[0] += wk * (-Fi.h) * bdj[0]; dzr[1] += wk * (-Fi.h) * bdj[1];
        dzz += wk * Fi.h * Fj.h;
        if (coupled) {  // MITC3_composite.py:204-240 (nodal-only form: the bubble block of K_qq is zero)
            const double *B = pr->B;
            {
                const double fx = Fj.fx, fy = Fj.fy;
                const double bx[3] = {-(B[1] * fy + B[2] * fx), -(B[4] * fy + B[5] * fx), -(B[7] * fy + B[8] * fx)};
                const double by[3] = {B[0] * fx + B[2] * fy, B[3] * fx + B[5] * fy, B[6] * fx + B[8] * fy};
                cpl_ij[0][0] += wA * (idhx * bx[0] + idhy * bx[2]); cpl_ij[0][1] += wA * (idhx * by[0] + idhy * by[2]);
                cpl_ij[1][0] += wA * (idhy * bx[1] + idhx * bx[2]); cpl_ij[1][1] += wA * (idhy * by[1] + idhx * by[2]);
            }
            {
                const double fx = Fi.fx, fy = Fi.fy;
                const double bx[3] = {-(B[1] * fy + B[2] * fx), -(B[4] * fy + B[5] * fx), -(B[7] * fy + B[8] * fx)};
                const double by[3] = {B[0] * fx + B[2] * fy, B[3] * fx + B[5] * fy, B[6] * fx + B[8] * fy};
                cpl_ji[0][0] += wA * (jdhx * bx[0] + jdhy * bx[2]); cpl_ji[0][1] += wA * (jdhx * by[0] + jdhy * by[2]);
                cpl_ji[1][0] += wA * (jdhy * bx[1] + jdhx * bx[2]); cpl_ji[1][1] += wA * (jdhy * by[1] + jdhx * by[2]);
            }
        }
    }
    // static condensation of the two bubble rotations (MITC3.py:958-1007)
    const double det = Kbb[0][0] * Kbb[1][1] - Kbb[0][1] * Kbb[1][0];
    // (four quotients over one reciprocal: the bits of the four IEEE divisions, femshell_geom.h)
    const double rdet = recip(det);
    const double i00 = divr(Kbb[1][1], det, rdet), i01 = divr(-Kbb[0][1], det, rdet), i10 = divr(-Kbb[1][0], det, rdet),
                 i11 = divr(Kbb[0][0], det, rdet);
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++)
            Kij[a][b] -= Ni[a][0] * (i00 * Nj[b][0] + i01 * Nj[b][1]) + Ni[a][1] * (i10 * Nj[b][0] + i11 * Nj[b][1]);
    if (ti == tj) {
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = a + 1; b < 3; b++) { const double m = 0.5 * (Kij[a][b] + Kij[b][a]); Kij[a][b] = m; Kij[b][a] = m; }
    }
    double L[6][6];
    for (int a = 0; a < 6; a++)
        for (int b = 0; b < 6; b++) L[a][b] = 0.0;
    {   // membrane CST (MITC3.py:1009-1022): area * Bm_i^T A Bm_j
        const double *A = pr->A;
        const double cu[3] = {A[0] * jdhx + A[2] * jdhy, A[3] * jdhx + A[5] * jdhy, A[6] * jdhx + A[8] * jdhy};
        const double cv[3] = {A[1] * jdhy + A[2] * jdhx, A[4] * jdhy + A[5] * jdhx, A[7] * jdhy + A[8] * jdhx};
        L[0][0] = area * (idhx * cu[0] + idhy * cu[2]) + dr[0][0]; L[0][1] = area * (idhx * cv[0] + idhy * cv[2]) + dr[0][1];
        L[1][0] = area * (idhy * cu[1] + idhx * cu[2]) + dr[1][0]; L[1][1] = area * (idhy * cv[1] + idhx * cv[2]) + dr[1][1];
    }
    L[0][5] = drz[0]; L[1][5] = drz[1]; L[5][0] = dzr[0]; L[5][1] = dzr[1]; L[5][5] = dzz;
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) L[2 + a][2 + b] = Kij[a][b];
    if (coupled) {
        L[0][3] = cpl_ij[0][0]; L[0][4] = cpl_ij[0][1]; L[1][3] = cpl_ij[1][0]; L[1][4] = cpl_ij[1][1];
        L[3][0] = cpl_ji[0][0]; L[4][0] = cpl_ji[0][1]; L[3][1] = cpl_ji[1][0]; L[4][1] = cpl_ji[1][1];
    }
    double R[3][3];
    for (int k = 0; k < 3; k++) {
        R[0][k] = geo[(1 + k) * GS + qq];
        R[1][k] = geo[(4 + k) * GS + qq];
        R[2][k] = geo[(7 + k) * GS + qq];
    }
    rotate_block6(L, R, G);
    if (ti == tj) {
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int b = a + 1; b < 6; b++) { const double m = 0.5 * (G[a][b] + G[b][a]); G[a][b] = m; G[b][a] = m; }
    }
}
