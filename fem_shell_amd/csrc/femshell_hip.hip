// femshell_hip.hip -- MI355X (gfx950) kernels + C ABI for the element-matrix / CSR-assembly hot path.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC   (csrc/Makefile)
//
// -ffp-contract=off is the file default because the GEOMETRY stage (local frame, local
// coordinates, characteristic vectors, Jacobians, tangent vectors) reproduces the reference's
// rounding sequence bit for bit (parity floor is eps*L/h, so a differently
// rounded geometry stage breaks 1e-12 parity on large meshes).  FMAs there are written explicitly
// where NumPy/OpenBLAS uses them.  Everything downstream (B^T C B, condensation, T^T K T) only has
// to agree at the eps level and opts back in to contraction with `#pragma clang fp contract(fast)`.
//
// Kernel design (DESIGN.md "mitc4_k"): fp64 VALU, 4 lanes per element (16 elements per wave).
//   stage 1: lane g of a quad = Gauss point g: geometry (replicated), strain-displacement data of
//            its Gauss point for the 4 nodes -> LDS (17 doubles per node, bank-conflict-free reads)
//   stage 2: lane q of a quad = node-pair task (i<=j): 6x6 local block summed over the 4 Gauss
//            points from LDS, bubble condensation, 3x3-block rotation to global axes, one
//            contiguous 288-byte store (FS_LAYOUT_SYMBLK).  10 tasks per element in 3 rounds.
// Citations: file:line relative to /root/reference/src/fem_shell/.

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "femshell_hip.h"
#include "femshell_internal.h"
#include "femshell_geom.h"
#include "femshell_tri.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *what, hipError_t e = hipSuccess) {
    if (e != hipSuccess)
        snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
    else
        snprintf(g_err, sizeof g_err, "%s", what);
    return code;
}

#define FS_LAUNCH_CHECK(name)                                         \
    do {                                                              \
        hipError_t e__ = hipGetLastError();                           \
        if (e__ != hipSuccess) return fail(FS_EHIP, name, e__);       \
    } while (0)

// ------------------------------------------------------------------------------------------
// LDS image of a 16-element tile, [component][element Q]: what stage 2 (node-pair blocks) needs from stage 1.  It is
// kept COMPACT -- per Gauss point only J^-1, the two edge-shear scale factors and detJ; dN/dx, dN/dy and the edge-tied
// shear rows B_gamma are rebuilt in stage 2 from these with the very expressions stage 1 uses (same bits) -- because
// the fused kernel holds the image of a whole patch (64 elements) and LDS capacity decides its occupancy.
//   IM_GP   [g < 4][7]   ia ib ic id (J^-1 at Gauss point g), two of the eight shear-core scalars (quad_stage1), detJ
//   IM_ROT  (free: the shear rotation sa ca sb cb, MITC4.py:844-850, is folded into the shear core)
//   IM_EX, IM_EY [4]     edge deltas 41, 21, 32, 34 in local axes
//   IM_KNB  [c < 6][node]  condensation rows
//   IM_S [15] IM_GT [18] IM_R [9] IM_INV [3]   membrane core, tying tangents, frame, Kbb^-1 (fused kernel only: the
//                        element kernel keeps them in registers)
//   IM_U    [c < 10][node] coupling cores (only when some element of the launch is coupled)
// ------------------------------------------------------------------------------------------
constexpr int IM_GP = 0, IM_EX = 32, IM_EY = 36, IM_KNB = 40, IM_S = 64, IM_GT = 79, IM_R = 97, IM_INV = 106;
constexpr int IM_U = 109, IM_NC = 109, IM_NC_COUPLED = 149;
constexpr int STG_SLOT = 38;               // staging slot of one 6x6 block: 36 doubles + 2 pad (304 B: the
                                           // 8-lane groups of ds_write_b128 then hit 32 distinct banks)
constexpr int LDS_WAVE = IM_NC_COUPLED * 16;            // 2384 doubles = 19072 B
constexpr int LDS_STG = LDS_WAVE;          // output staging of the tiled layout, 32 slots (half a wave per pass)
constexpr int LDS_WAVE_T16 = LDS_STG + 32 * STG_SLOT;   // 3600 doubles = 28800 B

// FS_LAYOUT_SYMBLK_T16 (quads only): tiles of 16 elements, 160 blocks per tile, stored in the order the
// stiffness kernel produces them -- round 0: pairs (g,g); round 1: (g,g+1),(0,3); round 2: (0,2),(1,3);
// within rounds 0/1 by half-tile, then task, then element.  Index of block (element e, pair p):
__host__ __device__ __forceinline__ int64_t blk_t16(int64_t e, int p) {
    // p -> round*4 + task:      p = 0  1  2  3  4  5  6  7  8  9
    const int rg = (int)((0x3629517840ull >> (4 * p)) & 15);   // nibbles (p = 0..9): 0,4,8,7,1,5,9,2,6,3
    const int round = rg >> 2, g = rg & 3, Q = (int)(e & 15);
    const int pos = (round < 2) ? (round * 64 + ((Q >> 3) << 5) + (g << 3) + (Q & 7)) : (128 + (g << 4) + Q);
    return (e >> 4) * 160 + pos;
}
__host__ __device__ __forceinline__ int64_t quad_blk(int tiled, int64_t e, int p) {
    return tiled ? blk_t16(e, p) : e * 10 + p;
}

// ------------------------------------------------------------------------------------------
// MITC4+ stiffness, FS_LAYOUT_SYMBLK output
//
// Membrane (and membrane-bending coupling) are PRE-CONTRACTED: with Y (5 x 12) the tying-row
// matrix (constant over the element) and W_g (3 x 5) = T_eps(g) * blend(g), B_m(g) = W_g Y, so
//   K_m = Y^T S Y,  S = sum_g detJ_g W_g^T C_m W_g   (5 x 5, symmetric)
//   K_c = Y^T U + U^T Y,  U_j = sum_g detJ_g W_g^T (B_lam B_kappa_j(g))   (5 x 2 per node)
// which is the same sum of products as MITC4.py:934-944 re-associated (eps-level difference) and
// removes B_m from LDS and from the Gauss-point loop of every node-pair task.
//
// Stage 1 (quad_stage1) and the node-pair block of stage 2 (quad_pair_block) are shared by mitc4_k_kernel and the
// fused evaluate-and-assemble kernel (fused_quad_k_kernel).  Through round 2 the two had to produce the SAME bits; since
// round 3 the fused kernel orders its Gauss-point sums and its transposes differently (quad_stage1<true>, fused_eval) and
// agrees to rounding.  The functions are still compiled with `fp contract(on)`: a*b + c contracts where the SOURCE expression says so (clang emits llvm.fmuladd,
// always an FMA for f64 on gfx950), never across statements, so the rounding sequence does not depend on the
// kernel the functions are inlined into (contract(fast) lets the backend pick which product of a*b + c*d to fuse).
// ------------------------------------------------------------------------------------------
struct QuadElem {       // per-element results of stage 1 that stage 2 needs beside the LDS image
    double S[15];       // upper triangle of the 5x5 membrane core, row-major (t <= u)
    double gt[6][3];    // tying tangents g_r(A), g_r(B), g_s(C), g_s(D), g_r(E), g_s(E), global axes
    double i00, i01, i11;   // Kbb^-1 (zero when singular: no condensation)
    int st;             // FS_ST_* bits
#ifdef FS_FUSED_TIMING
    unsigned long long tacc[8], tick;
#endif
};
#ifdef FS_FUSED_TIMING
#define FS_T1(ph) do { const unsigned long long t_ = __builtin_readcyclecounter(); el.tacc[ph] += t_ - el.tick; el.tick = t_; } while (0)
#else
#define FS_T1(ph) do { } while (0)
#endif

// Stage 1 for Gauss point g of the element held by the 4 lanes of a quad: bit-faithful geometry, then the data of this
// Gauss point -> LDS image `lds` (layout IM_* above, element Q of the 16-element tile), bubble blocks and membrane
// core reduced over the quad with DPP.
// All four lanes of every quad must be active.  any_coupled: uniform over the quad.
// ROWS (the fused kernel): the four Gauss-point lanes of element Q are lanes g * 16 + Q, the sums over the Gauss points are
// transposing row reductions (rows_sum4: every sum arrives on ONE lane, which writes it to the image -- the membrane core
// too, el.S is then scratch); otherwise lanes 4 Q + g and DPP all-reductions inside the quad.  The two forms add the four
// Gauss-point terms in different orders: equal to rounding.
template <bool ROWS>
__device__ __forceinline__ void quad_stage1(const double (&x)[4][3], const fs_shell_prop *pr, bool coupled,
                                            bool any_coupled, int g, int Q, double *lds, QuadGeom &geo, QuadElem &el) {
    double Bv[9];       // coupling matrix: fetched up front, in one batch, by every lane of a wave that needs it (a load at
                        // the point of use, under the lane's own condition, is waited for on the spot)
    if (any_coupled)
        for (int t = 0; t < 9; t++) Bv[t] = pr->B[t];
    FS_T1(7);
    quad_geometry(x, geo);
    FS_T1(0);
    const double xi = (g == 1 || g == 2) ? kGP : -kGP;
    const double eta = (g >= 2) ? kGP : -kGP;
    double dxi[4], deta[4];
    dshape(xi, eta, dxi, deta);
    double ia, ib, ic, id;
    const double detJ = jacobian(geo, dxi, deta, ia, ib, ic, id);
    int st = geo.status | ((detJ > 0.0) ? 0 : FS_ST_BAD_JAC);
    const int pig = rows_pi(g);            // ROWS: the value of a group of four this lane's row receives

    // tangent vectors at the 5 tying points (MITC4.py:589-612 evaluated at A,B,C,D,E; the zero
    // shape-derivative terms are dropped, which is exact), in GLOBAL axes: the membrane part of a node-pair block is
    // formed in global axes directly (the reference projects them on the frame, :287-296, and rotates K_m back).
    // gt[0..3] = g_r(A), g_r(B), g_s(C), g_s(D); gt[4], gt[5] = g_r(E), g_s(E)
    double (&gt)[6][3] = el.gt;
    {
        for (int k = 0; k < 3; k++) {
            gt[0][k] = 0.5 * x[2][k] + (-0.5 * x[3][k]);
            gt[1][k] = -0.5 * x[0][k] + 0.5 * x[1][k];
            gt[2][k] = -0.5 * x[1][k] + 0.5 * x[2][k];
            gt[3][k] = -0.5 * x[0][k] + 0.5 * x[3][k];
            gt[4][k] = ((-0.25 * x[0][k] + 0.25 * x[1][k]) + 0.25 * x[2][k]) + (-0.25 * x[3][k]);
            gt[5][k] = ((-0.25 * x[0][k] + (-0.25 * x[1][k])) + 0.25 * x[2][k]) + 0.25 * x[3][k];
        }
    }

    FS_T1(1);
    double *im = lds + Q;
    double (&S)[15] = el.S;
    double kbb00, kbb01, kbb11;
    {
#pragma clang fp contract(on)
        // shape-function derivatives in local axes (MITC4.py:642-657); stage 2 rebuilds them from J^-1 (quad_node_gp)
        double dNx[4], dNy[4];
        for (int i = 0; i < 4; i++) {
            dNx[i] = ia * dxi[i] + ib * deta[i];
            dNy[i] = ic * dxi[i] + id * deta[i];
        }
        im[(IM_GP + g * 7 + 0) << 4] = ia; im[(IM_GP + g * 7 + 1) << 4] = ib;
        im[(IM_GP + g * 7 + 2) << 4] = ic; im[(IM_GP + g * 7 + 3) << 4] = id;
        // ---- MITC4+ membrane (MITC4.py:351-392): W = T_eps * blend (3x5) ----
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
        // MITC4.py:334-347
        Te[0][0] = ia * ia; Te[0][1] = ic * ic; Te[0][2] = ia * ic;
        Te[1][0] = ib * ib; Te[1][1] = id * id; Te[1][2] = ib * id;
        Te[2][0] = 2.0 * ia * ib; Te[2][1] = 2.0 * ic * id; Te[2][2] = ia * id + ib * ic;
        const bool flat = __builtin_fabs(detJ) < 1e-12;  // MITC4.py:331-332 (a wave without such a lane skips the selects)
        if (__any(flat)) {
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++) Te[a][b] = flat ? ((a == b) ? 1.0 : 0.0) : Te[a][b];
        }
        double W[3][5];
        for (int a = 0; a < 3; a++)
            for (int t = 0; t < 5; t++) W[a][t] = Te[a][0] * bl[0][t] + Te[a][1] * bl[1][t] + Te[a][2] * bl[2][t];
        // S partial = detJ W^T Cm W
        {
            double CW[3][5];
            if constexpr (ROWS) {       // detJ on the nine constitutive numbers instead of the fifteen products
                double AJ[9];
                for (int t = 0; t < 9; t++) AJ[t] = detJ * pr->A[t];
                for (int a = 0; a < 3; a++)
                    for (int t = 0; t < 5; t++)
                        CW[a][t] = AJ[a * 3 + 0] * W[0][t] + AJ[a * 3 + 1] * W[1][t] + AJ[a * 3 + 2] * W[2][t];
            } else
            for (int a = 0; a < 3; a++)
                for (int t = 0; t < 5; t++)
                    CW[a][t] = detJ * (pr->A[a * 3 + 0] * W[0][t] + pr->A[a * 3 + 1] * W[1][t] + pr->A[a * 3 + 2] * W[2][t]);
            int n = 0;
            for (int t = 0; t < 5; t++)
                for (int u = t; u < 5; u++) S[n++] = W[0][t] * CW[0][u] + W[1][t] * CW[1][u] + W[2][t] * CW[2][u];
        }
        FS_T1(2);
        // coupling cores U_j = detJ W^T (Bl Bk_j)  (MITC4_composite.py:210-248, nodal-only form)
        if (any_coupled) {
            double Bl[3][3];
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++) Bl[a][b] = coupled ? detJ * Bv[a * 3 + b] : 0.0;
            if constexpr (ROWS) {
                // W^T Bl first (5 x 3), then the two non-zero curvature rows of every node: the products of the quad form
                // below in another order
                for (int t = 0; t < 5; t++) {
                    double WB[3];
                    for (int b = 0; b < 3; b++) WB[b] = W[0][t] * Bl[0][b] + W[1][t] * Bl[1][b] + W[2][t] * Bl[2][b];
                    double u0[4], u1[4];
                    for (int j = 0; j < 4; j++) {
                        u0[j] = -(WB[1] * dNy[j] + WB[2] * dNx[j]);
                        u1[j] = WB[0] * dNx[j] + WB[2] * dNy[j];
                    }
                    im[(IM_U + (2 * t) * 4 + pig) << 4] = rows_sum4(u0[0], u0[1], u0[2], u0[3]);
                    im[(IM_U + (2 * t + 1) * 4 + pig) << 4] = rows_sum4(u1[0], u1[1], u1[2], u1[3]);
                }
            } else
            for (int j = 0; j < 4; j++) {
                double LK[3][2];
                for (int a = 0; a < 3; a++) {
                    LK[a][0] = -(Bl[a][1] * dNy[j] + Bl[a][2] * dNx[j]);
                    LK[a][1] = Bl[a][0] * dNx[j] + Bl[a][2] * dNy[j];
                }
                for (int t = 0; t < 5; t++)
                    for (int c = 0; c < 2; c++) {
                        const double u = quad_sum(W[0][t] * LK[0][c] + W[1][t] * LK[1][c] + W[2][t] * LK[2][c]);
                        if (j == g) im[(IM_U + (2 * t + c) * 4 + j) << 4] = u;
                    }
            }
        }
        FS_T1(3);
        // ---- MITC4 edge-tied shear (MITC4.py:792-878) ----
        const auto &xl = geo.xl;
        double Ax, Bx, Cx, Ay, By, Cy;
        double ex[4], ey[4];  // edge deltas: 0: 41, 1: 21, 2: 32, 3: 34
        {
#pragma clang fp contract(off)
            ex[0] = xl[3][0] - xl[0][0]; ey[0] = xl[3][1] - xl[0][1];
            ex[1] = xl[1][0] - xl[0][0]; ey[1] = xl[1][1] - xl[0][1];
            ex[2] = xl[2][0] - xl[1][0]; ey[2] = xl[2][1] - xl[1][1];
            ex[3] = xl[2][0] - xl[3][0]; ey[3] = xl[2][1] - xl[3][1];
            Ax = ((-xl[0][0] + xl[1][0]) + xl[2][0]) - xl[3][0];
            Bx = ((xl[0][0] - xl[1][0]) + xl[2][0]) - xl[3][0];
            Cx = ((-xl[0][0] - xl[1][0]) + xl[2][0]) + xl[3][0];
            Ay = ((-xl[0][1] + xl[1][1]) + xl[2][1]) - xl[3][1];
            By = ((xl[0][1] - xl[1][1]) + xl[2][1]) - xl[3][1];
            Cy = ((-xl[0][1] - xl[1][1]) + xl[2][1]) + xl[3][1];
        }
        // Rot = [[sin b, -sin a], [-cos b, cos a]], a = atan2(Ay,Ax), b = pi/2 - atan2(Cx,Cy)
        const double nA = fsqrt(Ax * Ax + Ay * Ay), nC = fsqrt(Cx * Cx + Cy * Cy);
        const double rnA = recip(nA), rnC = recip(nC);
        const double sa = divr(Ay, nA, rnA), ca = divr(Ax, nA, rnA), sb = divr(Cy, nC, rnC), cb = divr(Cx, nC, rnC);
        double r1, r2;
        {
#pragma clang fp contract(off)
            const double p0 = Cx + xi * Bx, p1 = Cy + xi * By;
            r1 = fsqrt(FMA(p1, p1, p0 * p0));
            const double q0 = Ax + eta * Bx, q1 = Ay + eta * By;
            r2 = fsqrt(FMA(q1, q1, q0 * q0));
        }
        const double d8 = 8.0 * detJ, rd8 = recip(d8);
        const double f1 = divr(r1, d8, rd8), f2 = divr(r2, d8, rd8);
        im[(IM_GP + g * 7 + 6) << 4] = detJ;
        const double Cs00 = pr->Cs[0], Cs01 = pr->Cs[1], Cs10 = pr->Cs[2], Cs11 = pr->Cs[3];
        {
            // SHEAR CORE of the node-pair blocks.  The tied shear rows of node i at a Gauss point are
            //   Bs_i = p a_i c0_i^T + q b_i c1_i^T,  p = (sb, -cb), q = (-sa, ca),  a_i = (1 + sy_i eta) f1, b_i = (1 + sx_i xi) f2
            // (c0_i, c1_i: the node's two edge rows, quad_node_const), so that
            //   sum_g detJ Bs_i^T Cs Bs_j = k00 c0_i c0_j^T + k01 c0_i c1_j^T + k10 c1_i c0_j^T + k11 c1_i c1_j^T
            // with k.. bilinear in the signs (sx, sy) of the two nodes and EIGHT element scalars -- the same sum of products
            // as MITC4.py:986 evaluated per Gauss point, re-associated; Cs enters symmetrised (it is symmetric for every
            // property the reference builds).  They take the place of f1, f2 in the image.
            const double PP = sb * (Cs00 * sb - Cs01 * cb) - cb * (Cs10 * sb - Cs11 * cb);
            const double PQ = 0.5 * ((sb * (Cs01 * ca - Cs00 * sa) - cb * (Cs11 * ca - Cs10 * sa)) +
                                     (ca * (Cs10 * sb - Cs11 * cb) - sa * (Cs00 * sb - Cs01 * cb)));
            const double QQ = ca * (Cs11 * ca - Cs10 * sa) - sa * (Cs01 * ca - Cs00 * sa);
            const double t1 = detJ * f1 * f1, t2 = detJ * f1 * f2, t3 = detJ * f2 * f2;
            if constexpr (ROWS) {
                // scalars 0..3 and 4..7 as two groups of four: this row receives numbers pig and 4 + pig (odd rows: pig >= 2)
                const double sA = rows_sum4(t1, t1 * eta, t2, t2 * eta);
                const double sB = rows_sum4(t2 * xi, t2 * (xi * eta), t3, t3 * xi);
                const bool up = pig >= 2;
                im[(IM_GP + (pig >> 1) * 7 + 4 + (pig & 1)) << 4] = (up ? PQ : PP) * sA;
                im[(IM_GP + (2 + (pig >> 1)) * 7 + 4 + (pig & 1)) << 4] = (up ? QQ : PQ) * sB;
            } else {
            double sc[8];
            sc[0] = PP * quad_sum(t1);          sc[1] = PP * quad_sum(t1 * eta);
            sc[2] = PQ * quad_sum(t2);          sc[3] = PQ * quad_sum(t2 * eta);
            sc[4] = PQ * quad_sum(t2 * xi);     sc[5] = PQ * quad_sum(t2 * (xi * eta));
            sc[6] = QQ * quad_sum(t3);          sc[7] = QQ * quad_sum(t3 * xi);
            for (int n = 0; n < 4; n++)
                if (n == g) { im[(IM_GP + n * 7 + 4) << 4] = sc[2 * n]; im[(IM_GP + n * 7 + 5) << 4] = sc[2 * n + 1]; }
            }
        }
        if (g == 0)
            for (int t = 0; t < 4; t++) { im[(IM_EX + t) << 4] = ex[t]; im[(IM_EY + t) << 4] = ey[t]; }
        // node -> (edge of row 0 [eta-edges 21/34], weight), (edge of row 1 [xi-edges 41/32], weight)
        const double me = 1.0 - eta, pe = 1.0 + eta, mx = 1.0 - xi, px = 1.0 + xi;
        const int e0[4] = {1, 1, 3, 3}, e1[4] = {0, 2, 2, 0};
        const double w0[4] = {me, me, pe, pe}, w1[4] = {mx, px, px, mx};
        const double h0[4] = {-0.5, 0.5, 0.5, -0.5};  // G row of eta-edges: w coefficient
        const double h1[4] = {-0.5, -0.5, 0.5, 0.5};  // G row of xi-edges: w coefficient
        FS_T1(4);
        // ---- bubble enrichment (MITC4.py:214-222, 880-907): partial Knb, Kbb of this GP ----
        const double Cb00 = pr->D[0], Cb01 = pr->D[1], Cb02 = pr->D[2];
        const double Cb10 = pr->D[3], Cb11 = pr->D[4], Cb12 = pr->D[5];
        const double Cb20 = pr->D[6], Cb21 = pr->D[7], Cb22 = pr->D[8];
        const double Nb = (1.0 - xi * xi) * (1.0 - eta * eta);
        const double dNbxi = -2.0 * xi * (1.0 - eta * eta), dNbeta = -2.0 * eta * (1.0 - xi * xi);
        const double bx = ia * dNbxi + ib * dNbeta, by = ic * dNbxi + id * dNbeta;
        // Bkb = [[0,bx],[-by,0],[-bx,by]] ; Cb*Bkb (3x2)
        const double q00 = -Cb01 * by - Cb02 * bx, q01 = Cb00 * bx + Cb02 * by;
        const double q10 = -Cb11 * by - Cb12 * bx, q11 = Cb10 * bx + Cb12 * by;
        const double q20 = -Cb21 * by - Cb22 * bx, q21 = Cb20 * bx + Cb22 * by;
        // Bsb = [[0,Nb],[-Nb,0]] ; Cs*Bsb (2x2)
        const double s00 = -Cs01 * Nb, s01 = Cs00 * Nb, s10 = -Cs11 * Nb, s11 = Cs10 * Nb;
        if constexpr (ROWS) {
            // The same sum of products as the quad form below with detJ and the Gauss-point factors taken out of the
            // (row k, node i) loop: the shear part of entry (k, c) of node i is
            //   detJ (a_i c0 (sb s0c - cb s1c) + b_i c1 (ca s1c - sa s0c)),  a_i = w0_i f1, b_i = w1_i f2
            // and w0, w1 take two values each, so that four products per column c cover the four nodes; the bending part
            // carries detJ through bx, by.  Kbb uses the scaled factors too.
            const double NbJ = detJ * Nb, bxJ = detJ * bx, byJ = detJ * by;
            const double z00 = -Cs01 * NbJ, z01 = Cs00 * NbJ, z10 = -Cs11 * NbJ, z11 = Cs10 * NbJ;
            const double P0[2] = {f1 * (sb * z00 - cb * z10), f1 * (sb * z01 - cb * z11)};
            const double P1[2] = {f2 * (ca * z10 - sa * z00), f2 * (ca * z11 - sa * z01)};
            const double A0m[2] = {me * P0[0], me * P0[1]}, A0p[2] = {pe * P0[0], pe * P0[1]};
            const double A1m[2] = {mx * P1[0], mx * P1[1]}, A1p[2] = {px * P1[0], px * P1[1]};
            const double r00 = -Cb01 * byJ - Cb02 * bxJ, r01 = Cb00 * bxJ + Cb02 * byJ;
            const double r10 = -Cb11 * byJ - Cb12 * bxJ, r11 = Cb10 * bxJ + Cb12 * byJ;
            const double r20 = -Cb21 * byJ - Cb22 * bxJ, r21 = Cb20 * bxJ + Cb22 * byJ;
            const double qy[4] = {-0.25 * ey[0], -0.25 * ey[1], -0.25 * ey[2], -0.25 * ey[3]};
            const double qx[4] = {0.25 * ex[0], 0.25 * ex[1], 0.25 * ex[2], 0.25 * ex[3]};
            for (int k = 0; k < 3; k++) {
                double n0[4], n1[4];
                for (int i = 0; i < 4; i++) {
                    const double c0 = (k == 0) ? h0[i] : ((k == 1) ? qy[e0[i]] : qx[e0[i]]);
                    const double c1 = (k == 0) ? h1[i] : ((k == 1) ? qy[e1[i]] : qx[e1[i]]);
                    const double A00 = (i < 2) ? A0m[0] : A0p[0], A01 = (i < 2) ? A0m[1] : A0p[1];
                    const double A10 = (i == 0 || i == 3) ? A1m[0] : A1p[0], A11 = (i == 0 || i == 3) ? A1m[1] : A1p[1];
                    double t0 = 0.0, t1 = 0.0;
                    if (k == 1) { t0 = -dNy[i] * r10 - dNx[i] * r20; t1 = -dNy[i] * r11 - dNx[i] * r21; }
                    if (k == 2) { t0 = dNx[i] * r00 + dNy[i] * r20;  t1 = dNx[i] * r01 + dNy[i] * r21; }
                    n0[i] = (k == 0) ? (c0 * A00 + c1 * A10) : FMA(c0, A00, FMA(c1, A10, t0));
                    n1[i] = (k == 0) ? (c0 * A01 + c1 * A11) : FMA(c0, A01, FMA(c1, A11, t1));
                }
                im[(IM_KNB + (2 * k) * 4 + pig) << 4] = rows_sum4(n0[0], n0[1], n0[2], n0[3]);
                im[(IM_KNB + (2 * k + 1) * 4 + pig) << 4] = rows_sum4(n1[0], n1[1], n1[2], n1[3]);
            }
            kbb00 = (-by) * r10 + (-bx) * r20 + (-Nb) * z10;
            kbb01 = (-by) * r11 + (-bx) * r21 + (-Nb) * z11;
            kbb11 = bx * r01 + by * r21 + Nb * z01;
        } else
        for (int i = 0; i < 4; i++) {
            const double a0 = w0[i] * f1, a1 = w1[i] * f2;
            const double v0[3] = {a0 * h0[i], a0 * (-ey[e0[i]] * 0.25), a0 * (ex[e0[i]] * 0.25)};
            const double v1[3] = {a1 * h1[i], a1 * (-ey[e1[i]] * 0.25), a1 * (ex[e1[i]] * 0.25)};
            // Bk_i columns: thx = (0,-dNy,-dNx) ; thy = (dNx,0,dNy) ; w = 0
            const double kx0 = -dNy[i] * q10 - dNx[i] * q20, kx1 = -dNy[i] * q11 - dNx[i] * q21;
            const double ky0 = dNx[i] * q00 + dNy[i] * q20, ky1 = dNx[i] * q01 + dNy[i] * q21;
            for (int k = 0; k < 3; k++) {
                const double b0 = sb * v0[k] - sa * v1[k];
                const double b1 = -cb * v0[k] + ca * v1[k];
                double n0 = b0 * s00 + b1 * s10;
                double n1 = b0 * s01 + b1 * s11;
                if (k == 1) { n0 += kx0; n1 += kx1; }
                if (k == 2) { n0 += ky0; n1 += ky1; }
                // reduce over the 4 Gauss points of the quad; lane g publishes the rows of node g
                n0 = quad_sum(detJ * n0);
                n1 = quad_sum(detJ * n1);
                if (i == g) {
                    im[(IM_KNB + (2 * k) * 4 + i) << 4] = n0;
                    im[(IM_KNB + (2 * k + 1) * 4 + i) << 4] = n1;
                }
            }
        }
        // Kbb = Bkb^T Cb Bkb + Bsb^T Cs Bsb
        if constexpr (!ROWS) {
            kbb00 = detJ * ((-by) * q10 + (-bx) * q20 + (-Nb) * s10);
            kbb01 = detJ * ((-by) * q11 + (-bx) * q21 + (-Nb) * s11);
            kbb11 = detJ * (bx * q01 + by * q21 + Nb * s01);
        }
    }
    FS_T1(5);
    if constexpr (ROWS) {
        // membrane core: four groups of four (the last one padded); row g writes entries 4 k + pig
        for (int k4 = 0; k4 < 4; k4++) {
            const double v = rows_sum4(S[4 * k4], S[4 * k4 + 1], S[4 * k4 + 2], (k4 < 3) ? S[4 * k4 + 3] : 0.0);
            if (4 * k4 + pig < 15) im[(IM_S + 4 * k4 + pig) << 4] = v;
        }
        kbb00 = rows_allsum(kbb00); kbb01 = rows_allsum(kbb01); kbb11 = rows_allsum(kbb11);
        // FS_ST_BAD_JAC of any of the element's four Gauss points (the other bits are the same on its four lanes)
        unsigned long long bj = __ballot((st & FS_ST_BAD_JAC) != 0);
        bj |= bj >> 32; bj |= bj >> 16;
        if ((bj >> Q) & 1ull) st |= FS_ST_BAD_JAC;
    } else {
    for (int n = 0; n < 15; n++) S[n] = quad_sum(S[n]);
    kbb00 = quad_sum(kbb00); kbb01 = quad_sum(kbb01); kbb11 = quad_sum(kbb11);
    st |= __builtin_amdgcn_update_dpp(0, st, 0xB1, 0xF, 0xF, true);
    st |= __builtin_amdgcn_update_dpp(0, st, 0x4E, 0xF, 0xF, true);
    }
    // Kbb^-1 (MITC4.py:994-998; singular -> no condensation)
    const double p0 = kbb00 * kbb11, p1 = kbb01 * kbb01;
    const double kdet = p0 - p1;
    const bool cond_ok = (kdet != 0.0) && (kdet == kdet);
    const double rkdet = recip(kdet);
    el.i00 = cond_ok ? divr(kbb11, kdet, rkdet) : 0.0;
    el.i01 = cond_ok ? divr(-kbb01, kdet, rkdet) : 0.0;
    el.i11 = cond_ok ? divr(kbb00, kdet, rkdet) : 0.0;
    el.st = st;
    FS_T1(6);
}

// Shape-function derivatives and edge-tied shear rows of node n at Gauss point (xg, yg), rebuilt from the compact
// image with the expressions of quad_stage1 / dshape (MITC4.py:141-155, 642-657, 792-878): the factors (1 -+ eta),
// (1 -+ xi) come out of one FMA with an exact product, the signs and the 0.25 are exact scalings -- same bits.
struct QuadNodeConst { double qx, qy, sx, sy, c0[3], c1[3]; };   // 0.25 sx, 0.25 sy, signs, G rows of the node's 2 edges

__device__ __forceinline__ void quad_node_const(const double *im, int n, QuadNodeConst &c) {
#pragma clang fp contract(on)
    c.sx = (n == 1 || n == 2) ? 1.0 : -1.0;
    c.sy = (n >= 2) ? 1.0 : -1.0;
    c.qx = 0.25 * c.sx; c.qy = 0.25 * c.sy;
    const int e0 = (n < 2) ? 1 : 3, e1 = (n == 1 || n == 2) ? 2 : 0;     // eta-edge (21 / 34), xi-edge (41 / 32)
    const double ex0 = im[(IM_EX + e0) << 4], ey0 = im[(IM_EY + e0) << 4];
    const double ex1 = im[(IM_EX + e1) << 4], ey1 = im[(IM_EY + e1) << 4];
    c.c0[0] = 0.5 * c.sx; c.c0[1] = -ey0 * 0.25; c.c0[2] = ex0 * 0.25;   // h0 = -.5 .5 .5 -.5 == 0.5 sx
    c.c1[0] = 0.5 * c.sy; c.c1[1] = -ey1 * 0.25; c.c1[2] = ex1 * 0.25;   // h1 = -.5 -.5 .5 .5 == 0.5 sy
}

// Stage 2: the 6x6 block of node pair (ti <= tj) of element Q of the image `lds`, in GLOBAL axes:
//   R^T [ Y_i^T S Y_j + sum_g detJ (B_k^T C_b B_k + B_g^T C_s B_g + k_d B_d B_d^T) [+ coupling]
//         - K_nb,i Kbb^-1 K_nb,j^T ] R,   symmetrised when `diag` (ti == tj; MITC4.py:1034).
__device__ __forceinline__ void quad_pair_block(const double *lds, int Q, int ti, int tj, bool diag,
                                                const double (&R)[3][3], double i00, double i01, double i11,
                                                const double (&Cb)[3][3], double kd, bool any_coupled, double (&G)[6][6]) {
#pragma clang fp contract(on)
    const double *im = lds + Q;
    QuadNodeConst ci, cj;
    quad_node_const(im, ti, ci);
    quad_node_const(im, tj, cj);

    // L: the parts formed in the element frame (bending, shear, drilling, condensation); Gm: membrane block, GLOBAL axes
    // (global tying tangents); Ctr / Crt: membrane-bending coupling, translations global, rotations (thx, thy) local
    double L[6][6], Gm[3][3], Ctr[3][2], Crt[2][3];
    for (int a = 0; a < 6; a++)
        for (int b = 0; b < 6; b++) L[a][b] = 0.0;
    for (int a = 0; a < 3; a++)
        for (int c = 0; c < 2; c++) { Ctr[a][c] = 0.0; Crt[c][a] = 0.0; }

    // membrane: L[u,v,w][u,v,w] = Y_i^T S Y_j   (MITC4.py:934-944 pre-contracted).  A node has THREE non-zero tying rows:
    // g_r at A or B (weight sx / 2), g_s at C or D (weight sy / 2) and the centre row (sy g_r(E) + sx g_s(E)) / 8 -- the
    // rows and the core entries they meet are read from the image (IM_GT, IM_S) by index
    const int ai = (ti >= 2) ? 0 : 1, bi = (ti == 1 || ti == 2) ? 0 : 1;        // tying rows ai, 2 + bi, 4 of node ti
    const int aj = (tj >= 2) ? 0 : 1, bj = (tj == 1 || tj == 2) ? 0 : 1;
    double Yi[3][3], Yj[3][3];
    for (int k = 0; k < 3; k++) {
        const double g4 = im[(IM_GT + 12 + k) << 4], g5 = im[(IM_GT + 15 + k) << 4];
        Yi[0][k] = (0.5 * ci.sx) * im[(IM_GT + 3 * ai + k) << 4];
        Yi[1][k] = (0.5 * ci.sy) * im[(IM_GT + 6 + 3 * bi + k) << 4];
        Yi[2][k] = (0.125 * ci.sy) * g4 + (0.125 * ci.sx) * g5;
        Yj[0][k] = (0.5 * cj.sx) * im[(IM_GT + 3 * aj + k) << 4];
        Yj[1][k] = (0.5 * cj.sy) * im[(IM_GT + 6 + 3 * bj + k) << 4];
        Yj[2][k] = (0.125 * cj.sy) * g4 + (0.125 * cj.sx) * g5;
    }
    {
        // upper-triangle index (row-major, t <= u) of the core entries S(T_i[p], T_j[q]), T = (a, 2 + b, 4)
        const int n[3][3] = {{ai + aj + 3 * ai * aj, 2 + bj + 4 * ai, 4 + 4 * ai},
                             {2 + bi + 4 * aj, 9 + bi + bj + bi * bj, 11 + 2 * bi},
                             {4 + 4 * aj, 11 + 2 * bj, 14}};
        for (int b = 0; b < 3; b++) {
            double SY[3];
            for (int p = 0; p < 3; p++) {
                double v = im[(IM_S + n[p][0]) << 4] * Yj[0][b];
                v = FMA(im[(IM_S + n[p][1]) << 4], Yj[1][b], v);
                v = FMA(im[(IM_S + n[p][2]) << 4], Yj[2][b], v);
                SY[p] = v;
            }
            for (int a = 0; a < 3; a++) Gm[a][b] = FMA(Yi[2][a], SY[2], FMA(Yi[1][a], SY[1], Yi[0][a] * SY[0]));
        }
    }
    // membrane-bending coupling: L[uvw_i][th_j] += Y_i^T U_j ; L[th_i][uvw_j] += U_i^T Y_j
    if (any_coupled) {
        const int Ti[3] = {ai, 2 + bi, 4}, Tj[3] = {aj, 2 + bj, 4};
        for (int c = 0; c < 2; c++) {
            double Uj[3], Ui[3];       // cores of node tj at the rows of node ti, and the other way round
            for (int p = 0; p < 3; p++) {
                Uj[p] = im[(IM_U + (2 * Ti[p] + c) * 4 + tj) << 4];
                Ui[p] = im[(IM_U + (2 * Tj[p] + c) * 4 + ti) << 4];
            }
            for (int a = 0; a < 3; a++) {
                double v = 0.0, w2 = 0.0;
                for (int p = 0; p < 3; p++) { v = FMA(Yi[p][a], Uj[p], v); w2 = FMA(Ui[p], Yj[p][a], w2); }
                Ctr[a][c] = v; Crt[c][a] = w2;
            }
        }
    }

    // Gauss-point sums of the GEOMETRIC factors of the pair: G.. = sum_g w dN_a,i dN_b,j, H. = sum_g w dN_a,i N_j,
    // J. = sum_g w N_i dN_a,j, NN = sum_g w N_i N_j; the constitutive numbers meet them once, after the loop (the same
    // sums of products as MITC4.py:774-790, 909-928, 980, 1014 evaluated per Gauss point, re-associated)
    double Gxx = 0.0, Gxy = 0.0, Gyx = 0.0, Gyy = 0.0, Hx = 0.0, Hy = 0.0, Jx = 0.0, Jy = 0.0, NN = 0.0;
    for (int gg = 0; gg < 4; gg++) {
        const double xg = (gg == 1 || gg == 2) ? kGP : -kGP, yg = (gg >= 2) ? kGP : -kGP;
        const double *gp = im + ((IM_GP + gg * 7) << 4);
        const double ia = gp[0 << 4], ib = gp[1 << 4], ic = gp[2 << 4], id = gp[3 << 4], w = gp[6 << 4];
        const double Ei = 1.0 + ci.sy * yg, Xi = 1.0 + ci.sx * xg, Ej = 1.0 + cj.sy * yg, Xj = 1.0 + cj.sx * xg;
        const double dxi_i = ci.qx * Ei, deta_i = ci.qy * Xi, dxi_j = cj.qx * Ej, deta_j = cj.qy * Xj;
        const double dxi_ = ia * dxi_i + ib * deta_i, dyi_ = ic * dxi_i + id * deta_i;
        const double dxj = ia * dxi_j + ib * deta_j, dyj = ic * dxi_j + id * deta_j;
        const double Ni = 0.25 * Xi * Ei, Nj = 0.25 * Xj * Ej;
        const double wdxi = w * dxi_, wdyi = w * dyi_, wNi = w * Ni;
        Gxx = FMA(wdxi, dxj, Gxx); Gxy = FMA(wdxi, dyj, Gxy); Gyx = FMA(wdyi, dxj, Gyx); Gyy = FMA(wdyi, dyj, Gyy);
        Hx = FMA(wdxi, Nj, Hx); Hy = FMA(wdyi, Nj, Hy);
        Jx = FMA(wNi, dxj, Jx); Jy = FMA(wNi, dyj, Jy);
        NN = FMA(wNi, Nj, NN);
    }
    // bending: L[thx,thy][thx,thy] += sum_g w Bk_i^T Cb Bk_j, Bk = [[0, dNx], [-dNy, 0], [-dNx, dNy]]
    L[3][3] += (Cb[1][1] * Gyy + Cb[1][2] * Gyx) + (Cb[2][1] * Gxy + Cb[2][2] * Gxx);
    L[3][4] -= (Cb[1][0] * Gyx + Cb[1][2] * Gyy) + (Cb[2][0] * Gxx + Cb[2][2] * Gxy);
    L[4][3] -= (Cb[0][1] * Gxy + Cb[0][2] * Gxx) + (Cb[2][1] * Gyy + Cb[2][2] * Gyx);
    L[4][4] += (Cb[0][0] * Gxx + Cb[0][2] * Gxy) + (Cb[2][0] * Gyx + Cb[2][2] * Gyy);
    // drilling: k sum_g w outer(Bd_i, Bd_j) on (u, v, thz), Bd = (-dNy / 2, dNx / 2, -N)
    {
        const double q = 0.25 * kd, h = 0.5 * kd;
        L[0][0] += q * Gyy; L[0][1] -= q * Gyx; L[0][5] += h * Hy;
        L[1][0] -= q * Gxy; L[1][1] += q * Gxx; L[1][5] -= h * Hx;
        L[5][0] += h * Jy;  L[5][1] -= h * Jx;  L[5][5] += kd * NN;
    }
    // shear: L[w,thx,thy][w,thx,thy] += k00 c0_i c0_j^T + k01 c0_i c1_j^T + k10 c1_i c0_j^T + k11 c1_i c1_j^T (shear core of
    // quad_stage1; eta^2 = xi^2 = 1/3 at every Gauss point)
    {
        const double a0 = im[(IM_GP + 0 * 7 + 4) << 4], a1 = im[(IM_GP + 0 * 7 + 5) << 4];
        const double b0 = im[(IM_GP + 1 * 7 + 4) << 4], b1 = im[(IM_GP + 1 * 7 + 5) << 4];
        const double b2 = im[(IM_GP + 2 * 7 + 4) << 4], b3 = im[(IM_GP + 2 * 7 + 5) << 4];
        const double c0 = im[(IM_GP + 3 * 7 + 4) << 4], c1 = im[(IM_GP + 3 * 7 + 5) << 4];
        constexpr double k2 = kGP * kGP;
        const double syy = ci.sy * cj.sy, sxx = ci.sx * cj.sx;
        const double k00 = FMA(a0, 1.0 + syy * k2, a1 * (ci.sy + cj.sy));
        const double k11 = FMA(c0, 1.0 + sxx * k2, c1 * (ci.sx + cj.sx));
        const double k01 = (b0 + ci.sy * b1) + cj.sx * (b2 + ci.sy * b3);
        const double k10 = (b0 + cj.sy * b1) + ci.sx * (b2 + cj.sy * b3);
        for (int b = 0; b < 3; b++) {
            const double u0 = FMA(k00, cj.c0[b], k01 * cj.c1[b]), u1 = FMA(k10, cj.c0[b], k11 * cj.c1[b]);
            for (int a = 0; a < 3; a++) L[2 + a][2 + b] = FMA(ci.c0[a], u0, FMA(ci.c1[a], u1, L[2 + a][2 + b]));
        }
    }
    // bubble condensation: L[w,thx,thy][w,thx,thy] -= (Knb_i Kbb^-1) Knb_j^T (MITC4.py:996)
    {
        double ki[3][2], kj[3][2];
        for (int k = 0; k < 3; k++) {
            ki[k][0] = im[(IM_KNB + (2 * k) * 4 + ti) << 4]; ki[k][1] = im[(IM_KNB + (2 * k + 1) * 4 + ti) << 4];
            kj[k][0] = im[(IM_KNB + (2 * k) * 4 + tj) << 4]; kj[k][1] = im[(IM_KNB + (2 * k + 1) * 4 + tj) << 4];
        }
        for (int a = 0; a < 3; a++) {
            const double h0 = ki[a][0] * i00 + ki[a][1] * i01, h1 = ki[a][0] * i01 + ki[a][1] * i11;
            for (int b = 0; b < 3; b++) L[2 + a][2 + b] = FMA(-h0, kj[b][0], FMA(-h1, kj[b][1], L[2 + a][2 + b]));
        }
    }
    // rotation to global axes, 3x3 block-wise: G = R^T X R  (MITC4.py:634-640,1032-1033) -- of what was formed in the
    // element frame only, along its non-zero pattern: tt {[0..1][0..1], [2][2]}, tr {[2][3..4], [0..1][5]}, rt likewise,
    // rr {[3..4][3..4], [5][5]}
    {
        double Y[3][3];
        for (int p = 0; p < 3; p++) {                       // translations x translations
            Y[p][0] = FMA(R[1][p], L[1][0], R[0][p] * L[0][0]);
            Y[p][1] = FMA(R[1][p], L[1][1], R[0][p] * L[0][1]);
            Y[p][2] = R[2][p] * L[2][2];
        }
        for (int p = 0; p < 3; p++)
            for (int q = 0; q < 3; q++) G[p][q] = FMA(Y[p][2], R[2][q], FMA(Y[p][1], R[1][q], FMA(Y[p][0], R[0][q], Gm[p][q])));
        for (int p = 0; p < 3; p++) {                       // translations x rotations
            Y[p][0] = FMA(R[2][p], L[2][3], Ctr[p][0]);
            Y[p][1] = FMA(R[2][p], L[2][4], Ctr[p][1]);
            Y[p][2] = FMA(R[1][p], L[1][5], R[0][p] * L[0][5]);
        }
        for (int p = 0; p < 3; p++)
            for (int q = 0; q < 3; q++) G[p][3 + q] = FMA(Y[p][2], R[2][q], FMA(Y[p][1], R[1][q], Y[p][0] * R[0][q]));
        for (int q = 0; q < 3; q++) {                       // rotations x translations: Y = X R (+ coupling), then R^T Y
            Y[0][q] = FMA(L[3][2], R[2][q], Crt[0][q]);
            Y[1][q] = FMA(L[4][2], R[2][q], Crt[1][q]);
            Y[2][q] = FMA(L[5][1], R[1][q], L[5][0] * R[0][q]);
        }
        for (int p = 0; p < 3; p++)
            for (int q = 0; q < 3; q++) G[3 + p][q] = FMA(R[2][p], Y[2][q], FMA(R[1][p], Y[1][q], R[0][p] * Y[0][q]));
        for (int p = 0; p < 3; p++) {                       // rotations x rotations
            Y[p][0] = FMA(R[1][p], L[4][3], R[0][p] * L[3][3]);
            Y[p][1] = FMA(R[1][p], L[4][4], R[0][p] * L[3][4]);
            Y[p][2] = R[2][p] * L[5][5];
        }
        for (int p = 0; p < 3; p++)
            for (int q = 0; q < 3; q++) G[3 + p][3 + q] = FMA(Y[p][2], R[2][q], FMA(Y[p][1], R[1][q], Y[p][0] * R[0][q]));
    }
    if (diag) {  // diagonal block: K = 0.5 (K + K^T)  (MITC4.py:1034)
        for (int a = 0; a < 6; a++)
            for (int b = a + 1; b < 6; b++) {
                const double m = 0.5 * (G[a][b] + G[b][a]);
                G[a][b] = m; G[b][a] = m;
            }
    }
}

template <bool TILED>
__global__ __launch_bounds__(64) void mitc4_k_kernel(const double *__restrict__ xyz,
                                                     const int32_t *__restrict__ prop_id,
                                                     const fs_shell_prop *__restrict__ props, int32_t n_props,
                                                     int64_t E, double *__restrict__ out,
                                                     int32_t *__restrict__ status) {
    __shared__ double lds[TILED ? LDS_WAVE_T16 : LDS_WAVE];
    const int lane = threadIdx.x;
    const int Q = lane >> 2, g = lane & 3;
    const int64_t e_raw = (int64_t)blockIdx.x * 16 + Q;
    const bool valid = e_raw < E;
    const int64_t e = valid ? e_raw : (E - 1);

    double x[4][3];
    {
        const double2 *p = reinterpret_cast<const double2 *>(xyz + e * 12);
        for (int t = 0; t < 6; t++) {
            const double2 v = p[t];
            (&x[0][0])[2 * t] = v.x;
            (&x[0][0])[2 * t + 1] = v.y;
        }
    }
    const fs_shell_prop *pr = props + safe_pid(prop_id[e], n_props);
    const bool coupled = (pr->flags & FS_PROP_COUPLED) != 0;
    const bool any_coupled = __any(coupled);

    // ---------------- stage 1: Gauss point g ----------------
    QuadGeom geo;
    QuadElem el;
    quad_stage1<false>(x, pr, coupled, any_coupled, g, Q, lds, geo, el);
    if (status && valid && g == 0) status[e] = el.st;
    // membrane core and tying tangents go to the image: stage 2 reads them by tying row (quad_pair_block)
    if (g == 0) {
        for (int n = 0; n < 15; n++) lds[((IM_S + n) << 4) + Q] = el.S[n];
    } else if (g == 1) {
        for (int t = 0; t < 18; t++) lds[((IM_GT + t) << 4) + Q] = (&el.gt[0][0])[t];
    }

    __syncthreads();

    // ---------------- stage 2: node-pair tasks ----------------
    const double Cb[3][3] = {{pr->D[0], pr->D[1], pr->D[2]}, {pr->D[3], pr->D[4], pr->D[5]}, {pr->D[6], pr->D[7], pr->D[8]}};
    const double kd = pr->k_drill;
    const double R[3][3] = {{geo.e1[0], geo.e1[1], geo.e1[2]}, {geo.e2[0], geo.e2[1], geo.e2[2]}, {geo.e3[0], geo.e3[1], geo.e3[2]}};

    for (int round = 0; round < 3; round++) {
        int ti, tj;
        bool active = valid;
        if (round == 0) { ti = g; tj = g; }
        else if (round == 1) { ti = (g == 3) ? 0 : g; tj = (g == 3) ? 3 : g + 1; }
        else { ti = g & 1; tj = (g & 1) + 2; active = active && (g < 2); }
        double G[6][6];
        quad_pair_block(lds, Q, ti, tj, round == 0, R, el.i00, el.i01, el.i11, Cb, kd, any_coupled, G);
        if constexpr (TILED) {
            // ---- FS_LAYOUT_SYMBLK_T16 output.  A lane that stores its own 288-byte block issues 18
            // instructions of 64 scattered 16-byte pieces; that form costs 0.45 ms of 1.58 ms per 1.25 M
            // elements (L2 write-request rate, not bandwidth).  Here the blocks go through an LDS staging
            // area, half a wave per pass (round 2 has 32 tasks: one pass), and leave as 9 instructions of
            // 1024 contiguous, line-aligned bytes: the tile layout stores blocks in exactly this
            // production order, so the global address is linear in (instruction, lane) -- no per-store
            // index arithmetic (which cost more than the coalescing gained when tried on other layouts).
            const int npass = (round == 2) ? 1 : 2;
            for (int h = 0; h < npass; h++) {
                const bool mine = (round == 2) ? (g < 2) : ((lane >> 5) == h);
                const int slot = (round == 2) ? ((g << 4) + Q) : ((g << 3) + (Q & 7));
                if (mine) {
                    double2 *sp = reinterpret_cast<double2 *>(lds + LDS_STG + slot * STG_SLOT);
                    for (int t = 0; t < 18; t++) sp[t] = make_double2((&G[0][0])[2 * t], (&G[0][0])[2 * t + 1]);
                }
                __syncthreads();
                double2 *gp = reinterpret_cast<double2 *>(out + ((int64_t)blockIdx.x * 160 + round * 64 + h * 32) * 36) + lane;
                const double2 *sp2 = reinterpret_cast<const double2 *>(lds + LDS_STG);
                for (int i = 0; i < 9; i++) {
                    const int c = i * 64 + lane;            // 16-byte chunk of the pass's 32 blocks
                    const int bl = (c * 3641) >> 16;        // c / 18 for c < 576
                    gp[i * 64] = sp2[c + bl];               // staging slots are 19 chunks apart
                }
                __syncthreads();
            }
        } else if (active) {
            const int p = ti * 4 - (ti * (ti - 1)) / 2 + (tj - ti);
            double2 *dst = reinterpret_cast<double2 *>(out + (e * 10 + p) * 36);
            for (int t = 0; t < 18; t++) dst[t] = make_double2((&G[0][0])[2 * t], (&G[0][0])[2 * t + 1]);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Fused evaluate-and-assemble, GATHER form (fs_fused_assemble_quad): no element matrices in HBM.
//
// A workgroup owns a PATCH of nodes (fem_shell_amd/fused.py builds the plan for any quad mesh):
//   stage 1  every element that touches a patch node (<= 16 NT of them; the one-element halo is recomputed) runs
//            quad_stage1 (the row-reduction form, lane = Gauss point * 16 + element), 16 elements per wave; besides the
//            strain-displacement image
//            the per-element data that stage 2 of mitc4_k_kernel keeps in registers (membrane core S, tying
//            tangents, frame, Kbb^-1) goes to LDS, because here a lane meets elements it did not evaluate;
//   stage 2  a lane evaluates up to TWO contributions (element, local node pair) of a CSR node-pair block (I <= J) with
//            quad_pair_block (a contribution the block receives transposed: the block of the swapped pair); the
//            contributions of a block sit on adjacent lanes and are summed IN ELEMENT ORDER along DPP row_shr chains
//            ( ((x0 + x1) + x2) + x3 ...: the order of csr_numeric_desc6_lds_kernel and of the reference's ADD_VALUES
//            insertion), and the chain's last lane hands the block to the staging area, from which it is stored to
//            (I,J) and, transposed, to (J,I).
// The result equals the CSR of mitc4_k_kernel + csr_numeric_desc6_lds_kernel to rounding, is exactly symmetric and the
// same bits in every run (the mass operator: the same bits as the two-kernel route); HBM traffic per element drops from
// 2 880 (blocks out) + 2 880 (blocks in) + 2 592 (CSR out) + 432 (descriptors) to 2 592 + ~330 (plan records).
// ------------------------------------------------------------------------------------------
// operator of the fused kernel: stiffness K (image IM_*, quad_stage1 / quad_pair_block) or consistent mass M (image
// IMM_*: frame and the four Jacobian determinants; block arithmetic of mitc4_m_tiled_kernel)
constexpr int FS_OP_K = 0, FS_OP_M = 1;
constexpr int IMM_R = 0, IMM_DET = 9, IMM_NC = 13;
template <bool COUPLED, int OP = FS_OP_K>
struct FusedLayout {
    static constexpr int TILE = (OP == FS_OP_M ? IMM_NC : (COUPLED ? IM_NC_COUPLED : IM_NC)) * 16;     // doubles per 16-element tile
};

constexpr int FZ_SLOT = 21;               // staging slot of the fused kernel, in 16-byte pieces

#define FS_EV_SLOT(ev) ((int)((ev) & 255u))
#define FS_EV_TI(ev) ((int)(((ev) >> 8) & 3u))
#define FS_EV_TJ(ev) ((int)(((ev) >> 10) & 3u))
#define FS_EV_TR(ev) ((((ev) >> 12) & 1u) != 0)
#define FS_EV_ACTIVE(ev) ((((ev) >> 13) & 1u) != 0)

// lane i reads lane i - 1 of its 16-lane row (row_shr:1); lane 0 of a row reads 0
__device__ __forceinline__ double row_shr1(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x111, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x111, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

// s_pid entry of an element slot: property row | FS_SLOT_COUPLED when the row has membrane-bending coupling
constexpr int FS_SLOT_COUPLED = 1 << 30;

// any_coupled: some contribution of the calling WAVE belongs to a coupled element (wave-uniform; the coupling terms of
// the others add products with zero cores to +0.0, i.e. the same bits as skipping them)
template <bool COUPLED>
__device__ __forceinline__ void fused_eval(uint32_t ev, const double *lds, const int *s_pid,
                                           const fs_shell_prop *__restrict__ props, bool any_coupled, double (&Y)[36]) {
    using LY = FusedLayout<COUPLED>;
    // a contribution the block receives transposed is formed as the block of the swapped pair, (tj, ti): the same
    // products with the roles of the two nodes exchanged (equal to the transpose up to rounding) instead of 36 selects
    const bool tr = FS_EV_TR(ev);
    const int slot = FS_EV_SLOT(ev), ti = tr ? FS_EV_TJ(ev) : FS_EV_TI(ev), tj = tr ? FS_EV_TI(ev) : FS_EV_TJ(ev);
    const int T = slot >> 4, Q = slot & 15;
    const double *img = lds + T * LY::TILE;
    const double *ex = img + Q;
    double R[3][3];
    for (int t = 0; t < 9; t++) (&R[0][0])[t] = ex[(IM_R + t) << 4];
    const double i00 = ex[(IM_INV + 0) << 4], i01 = ex[(IM_INV + 1) << 4], i11 = ex[(IM_INV + 2) << 4];
    const fs_shell_prop *pr = props + (s_pid[slot] & (FS_SLOT_COUPLED - 1));
    const double Cb[3][3] = {{pr->D[0], pr->D[1], pr->D[2]}, {pr->D[3], pr->D[4], pr->D[5]}, {pr->D[6], pr->D[7], pr->D[8]}};
    const double kd = pr->k_drill;
    // (diagonal blocks are symmetrised once, on the sum of their contributions: see the patch loop)
    quad_pair_block(img, Q, ti, tj, false, R, i00, i01, i11, Cb, kd, COUPLED && any_coupled,
                    reinterpret_cast<double (&)[6][6]>(Y));
}

// One contribution of the consistent mass matrix: block (ti <= tj) of the element in image slot FS_EV_SLOT(ev), the
// arithmetic of mitc4_m_tiled_kernel statement for statement (MITC4.py:1038-1061), transposed when the record says so.
__device__ __forceinline__ void fused_eval_mass(uint32_t ev, const double *lds, const int *s_pid,
                                                const fs_shell_prop *__restrict__ props, double (&Y)[36]) {
    const int slot = FS_EV_SLOT(ev), ti = FS_EV_TI(ev), tj = FS_EV_TJ(ev);
    const double *ex = lds + (slot >> 4) * (IMM_NC * 16) + (slot & 15);
    double R[3][3], det[4];
    for (int t = 0; t < 9; t++) (&R[0][0])[t] = ex[(IMM_R + t) << 4];
    for (int q = 0; q < 4; q++) det[q] = ex[(IMM_DET + q) << 4];
    const fs_shell_prop *pr = props + s_pid[slot];
    const double mt = pr->m_t, mr = pr->m_r;
    const double sxi = (ti == 1 || ti == 2) ? 1.0 : -1.0, syi = (ti >= 2) ? 1.0 : -1.0;
    const double sxj = (tj == 1 || tj == 2) ? 1.0 : -1.0, syj = (tj >= 2) ? 1.0 : -1.0;
    double ct = 0.0, cr = 0.0;
    for (int q = 0; q < 4; q++) {
        const double xi = (q == 1 || q == 2) ? kGP : -kGP, eta = (q >= 2) ? kGP : -kGP;
        const double Ni = 0.25 * (1.0 + sxi * xi) * (1.0 + syi * eta);
        const double Nj = 0.25 * (1.0 + sxj * xi) * (1.0 + syj * eta);
        ct = ct + Ni * Nj * mt * 1.0 * det[q];
        cr = cr + Ni * Nj * mr * 1.0 * det[q];
    }
    double G[6][6];
    for (int a = 0; a < 6; a++)
        for (int b = 0; b < 6; b++) G[a][b] = 0.0;
    for (int p3 = 0; p3 < 3; p3++)
        for (int q = 0; q < 3; q++) {
            double st = 0.0, sr = 0.0;
            for (int k = 0; k < 3; k++) {
                st = FMA(R[k][p3] * ct, R[k][q], st);
                sr = FMA(R[k][p3] * cr, R[k][q], sr);
            }
            G[p3][q] = st;
            G[3 + p3][3 + q] = sr;
        }
    const bool tr = FS_EV_TR(ev);
    for (int k = 0; k < 6; k++)
        for (int l = 0; l < 6; l++) Y[k * 6 + l] = tr ? G[l][k] : G[k][l];
}

#ifdef FS_FUSED_TIMING
__device__ unsigned long long g_fused_t[24];
#define FS_TICK(ph) do { const unsigned long long t_ = __builtin_readcyclecounter(); \
    acc_[ph] += t_ - tick_; tick_ = t_; } while (0)
#else
#define FS_TICK(ph) do { } while (0)
#endif
template <bool COUPLED, int SP, int OP = FS_OP_K>
__global__ __launch_bounds__(256) void fused_quad_k_kernel(const double *__restrict__ xyz,
                                                           const int32_t *__restrict__ prop_id,
                                                           const fs_shell_prop *__restrict__ props, int32_t n_props,
                                                           int64_t n_pat, const int64_t *__restrict__ pat_eptr,
                                                           const int32_t *__restrict__ pat_elems,
                                                           const int64_t *__restrict__ pat_rptr,
                                                           const fs_fused_rec *__restrict__ recs,
                                                           double *__restrict__ data, int32_t *__restrict__ status) {
    using LY = FusedLayout<COUPLED, OP>;
    constexpr int NT = 4;                       // 16-element tiles of the LDS image: one per wave
    __shared__ double lds[NT * LY::TILE];
    __shared__ int s_pid[NT * 16];
    __shared__ double2 s_stage[4 * FZ_SLOT * SP];   // per wave: SP staging slots of 21 x 16 bytes
    __shared__ unsigned char s_tab[4 * 128];        // per wave: rank of a transposed copy -> staging slot of its block
    __shared__ int s_st[NT * 16], s_el[NT * 16];    // status word and element index of a slot (stored after stage 2)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // stage-1 lane map: stiffness -- Gauss point g of element Q on lane g * 16 + Q (row reductions, quad_stage1<true>);
    // mass -- lane 4 Q + g
    constexpr bool FQ_ROWS = (OP == FS_OP_K);
    const int Q = FQ_ROWS ? (lane & 15) : (lane >> 2), g = FQ_ROWS ? (lane >> 4) : (lane & 3), slot = wave * 16 + Q;
    // PERSISTENT workgroups: workgroup w lives on XCD w & 7 (round-robin dispatch) and walks through that XCD's
    // contiguous range of patches -- neighbouring patches (shared halo elements, shared CSR lines) meet in one L2,
    // the stores of a patch drain while stage 1 of the next one runs, and the next patch's inputs are fetched
    // under the tail of the current one.
    const int xcd = blockIdx.x & 7, wpx = (int)(gridDim.x >> 3);
    const int64_t per_xcd = (n_pat + 7) / 8;
    int64_t p = xcd * per_xcd + (blockIdx.x >> 3);
    const int64_t pend = ((xcd + 1) * per_xcd < n_pat) ? (xcd + 1) * per_xcd : n_pat;
    if (p >= pend) return;

    int ne = (int)(pat_eptr[p + 1] - pat_eptr[p]);
    int64_t r0 = pat_rptr[p], r1 = pat_rptr[p + 1];
    int64_t e = pat_elems[pat_eptr[p] + ((slot < ne) ? slot : ne - 1)];
    double x[4][3];
    int pid;
    {
        const double2 *xp = reinterpret_cast<const double2 *>(xyz + e * 12);
        for (int t = 0; t < 6; t++) { const double2 v = xp[t]; (&x[0][0])[2 * t] = v.x; (&x[0][0])[2 * t + 1] = v.y; }
        pid = prop_id[e];
    }
    // flags of the element's property row: needed at the very start of stage 1 (wave-uniform coupling switch), so they
    // travel one patch ahead like the coordinates
    uint32_t pflags = props[safe_pid(pid, n_props)].flags;
    // headers travel TWO patches ahead (uniform scalar loads) so that the element list of the next patch can be fetched
    // at the top of the current one without waiting for its header
    int64_t e0n = 0, r0n = 0, r1n = 0;
    int nen = 0;
    if (p + wpx < pend) {
        e0n = pat_eptr[p + wpx];
        nen = (int)(pat_eptr[p + wpx + 1] - e0n);
        r0n = pat_rptr[p + wpx]; r1n = pat_rptr[p + wpx + 1];
    }
#ifdef FS_FUSED_TIMING
    unsigned long long acc_[19] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tick_ = __builtin_readcyclecounter();
#endif
    // Vector-memory operations return IN ORDER: a load issued behind a patch's store burst (36 instructions per wave,
    // 124 KB per CU, draining at the chip's write rate for several microseconds) comes back only after it -- and the
    // compiler guards every loaded value that is still pending when the loop comes round with a conservative vmcnt wait
    // at its first use, i.e. at the top of stage 1.  Two measures keep stage 1 free of vector-memory waits: (1) an
    // explicit vmcnt(0) in front of the loop and between the staging writes and the stores of a pass, where every
    // prefetch has long arrived: the wait-count bookkeeping then carries nothing round the loop; (2) the property row of
    // a wave whose elements share one is read through a uniform pointer, i.e. SCALAR loads, which another counter
    // tracks (waves across a laminate boundary keep the per-lane loads and wait for the stores).  Strip 1.49 -> 1.42 ms.
    __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0); lgkmcnt and expcnt untouched
    for (;;) {
        // element of the next patch for this lane (header already here), header of the patch after it
        const int64_t pn = p + wpx, pnn = pn + wpx;
        const bool more = pn < pend;
        // (unconditional loads from clamped indices: a load inside a branch is waited for at the end of the branch)
        const int64_t en = pat_elems[more ? e0n + ((slot < nen) ? slot : nen - 1) : 0];
        const int64_t pq = (pnn < pend) ? pnn : p;
        const int64_t e0nn = pat_eptr[pq];
        const int nenn = (int)(pat_eptr[pq + 1] - e0nn);
        const int64_t r0nn = pat_rptr[pq], r1nn = pat_rptr[pq + 1];
        // this lane's record of the first round: fetched now, used after stage 1
        int4 pa = make_int4(-1, -1, -1, -1), pb = pa, pc4 = make_int4(0, 0, 0, 0), pd = make_int4(0, 0, 0, -1);
        if (r0 + tid < r1) {
            const int4 *rp = reinterpret_cast<const int4 *>(recs + r0 + tid);
            pa = rp[0]; pb = rp[1]; pc4 = rp[2]; pd = rp[3];
        }
        // ---------------- stage 1: the patch's elements, 16 per wave ----------------
        if (wave * 16 < ne) {
            const bool valid = slot < ne;
            int bad = 0;
            if (pid < 0 || pid >= n_props) { pid = 0; bad = FS_ST_BAD_PROP; }     // never read outside the table
            // one property row for the whole wave (3 of 4 waves of the bench plate): uniform pointer, scalar loads
            const int pid_u = __builtin_amdgcn_readfirstlane(pid);
            const bool pid_uniform = __all(pid == pid_u);
            const fs_shell_prop *pr = props + pid;
            const bool flagged = (pflags & FS_PROP_COUPLED) != 0;
            if (OP == FS_OP_K && !COUPLED && flagged) bad = FS_ST_BAD_PROP;      // caller promised a table without coupled rows
            double *img = lds + wave * LY::TILE;
            QuadGeom geo;
            if constexpr (OP == FS_OP_M) {
                // mass: frame and the Jacobian determinant of Gauss point g (mitc4_m_tiled_kernel's stage)
                quad_geometry(x, geo);
                const double xi = (g == 1 || g == 2) ? kGP : -kGP, eta = (g >= 2) ? kGP : -kGP;
                double dxi[4], deta[4], ia, ib, ic, id;
                dshape(xi, eta, dxi, deta);
                double *ex = img + Q;
                ex[(IMM_DET + g) << 4] = jacobian(geo, dxi, deta, ia, ib, ic, id);
                if (g == 0) {
                    for (int k = 0; k < 3; k++) {
                        ex[(IMM_R + k) << 4] = geo.e1[k]; ex[(IMM_R + 3 + k) << 4] = geo.e2[k]; ex[(IMM_R + 6 + k) << 4] = geo.e3[k];
                    }
                    s_pid[slot] = pid;
                    if (status && valid) status[e] = geo.status | bad;
                }
            } else {
            QuadElem el;
            // the coupling cores are formed only by waves that hold a coupled element; the others zero their part of the
            // image (lane g owns the rows of node g, as in quad_stage1)
            const bool wave_coupled = COUPLED && __any(flagged);
#ifdef FS_FUSED_TIMING
            for (int t = 0; t < 8; t++) el.tacc[t] = 0;
            el.tick = __builtin_readcyclecounter();
#endif
#ifdef FS_FUSED_TIMING
            acc_[18] += el.tick - tick_;       // loop top .. call of quad_stage1 (prefetch issue, waits for the inputs)
#endif
            if (pid_uniform) quad_stage1<FQ_ROWS>(x, props + pid_u, COUPLED && flagged, wave_coupled, g, Q, img, geo, el);
            else quad_stage1<FQ_ROWS>(x, pr, COUPLED && flagged, wave_coupled, g, Q, img, geo, el);
#ifdef FS_FUSED_TIMING
            for (int t = 0; t < 7; t++) acc_[7 + t] += el.tacc[t];
            acc_[17] += el.tacc[7];
#endif
            double *ex = img + Q;
            if (COUPLED && !wave_coupled)
                for (int t = 0; t < 10; t++) ex[(IM_U + t * 4 + g) << 4] = 0.0;
            if (g == 0) {
                if (!FQ_ROWS)
                    for (int n = 0; n < 15; n++) ex[(IM_S + n) << 4] = el.S[n];
                s_pid[slot] = pid | ((COUPLED && flagged) ? FS_SLOT_COUPLED : 0);
                // the status word waits in LDS for the end of the patch: stored from here, it would be one more vector-memory
                // operation the stage-1 barrier's vmcnt(0) has to see acknowledged (1.5 % of a step)
                s_st[slot] = valid ? (el.st | bad) : -1; s_el[slot] = (int)e;
            } else if (g == 1) {
                for (int t = 0; t < 18; t++) ex[(IM_GT + t) << 4] = (&el.gt[0][0])[t];
            } else if (g == 2) {
                for (int k = 0; k < 3; k++) {
                    ex[(IM_R + k) << 4] = geo.e1[k]; ex[(IM_R + 3 + k) << 4] = geo.e2[k]; ex[(IM_R + 6 + k) << 4] = geo.e3[k];
                }
            } else {
                ex[(IM_INV + 0) << 4] = el.i00; ex[(IM_INV + 1) << 4] = el.i01; ex[(IM_INV + 2) << 4] = el.i11;
            }
            }
        }
        // property id of the next patch's element: `en` has arrived under stage 1 (the empty asm keeps the compiler from
        // forming the address, and waiting for `en`, at the top of the loop); its flags follow under the evaluations, its
        // coordinates under the last round's stores
        int64_t en_use = en;
        asm volatile("" : "+v"(en_use));
        int pidn = prop_id[en_use];
        FS_TICK(0);
        __syncthreads();
        FS_TICK(1);

        // property id of the next patch's element (its flags follow under the evaluations, its coordinates under the
        // last round's stores)
        double xn[12];
        uint32_t pflagsn = 0;

        // ---------------- stage 2: two contributions per lane, chained lanes for blocks with more ----------------
        for (int64_t rb = r0; rb < r1; rb += 256) {
            const int64_t r = rb + tid;
            int4 ra = pa, rb4 = pb, rc = pc4, rd = pd;
            if (rb != r0) {          // further rounds (patches of more than 256 lanes): fetched on the spot
                ra = make_int4(-1, -1, -1, -1); rb4 = ra; rc = make_int4(0, 0, 0, 0); rd = make_int4(0, 0, 0, -1);
                if (r < r1) {
                    const int4 *rp = reinterpret_cast<const int4 *>(recs + r);
                    ra = rp[0]; rb4 = rp[1]; rc = rp[2]; rd = rp[3];
                }
            }
            const uint32_t ev0 = (uint32_t)rd.x, ev1 = (uint32_t)rd.y, fl = (uint32_t)rd.z, slots = (uint32_t)rd.w;
            // X = first contribution, Y = second one (+0.0 when absent: adds nothing)
            double X[36], Y[36];
            const bool last_round = more && rb + 256 >= r1;
            const bool a0 = FS_EV_ACTIVE(ev0), a1 = FS_EV_ACTIVE(ev1);
            const bool c0 = COUPLED && __any(a0 && (s_pid[FS_EV_SLOT(ev0)] & FS_SLOT_COUPLED));
            const bool c1 = COUPLED && __any(a1 && (s_pid[FS_EV_SLOT(ev1)] & FS_SLOT_COUPLED));
            if (a0) {
                if constexpr (OP == FS_OP_M) fused_eval_mass(ev0, lds, s_pid, props, X);
                else fused_eval<COUPLED>(ev0, lds, s_pid, props, c0, X);
            }
            else
                for (int t = 0; t < 36; t++) X[t] = 0.0;
            asm volatile("" : "+v"(pidn));                     // (wait for pidn here, not right after its load)
            pflagsn = props[safe_pid(pidn, n_props)].flags;     // (every round: see above)
            FS_TICK(2);
            if (a1) {
                if constexpr (OP == FS_OP_M) fused_eval_mass(ev1, lds, s_pid, props, Y);
                else fused_eval<COUPLED>(ev1, lds, s_pid, props, c1, Y);
            }
            else
                for (int t = 0; t < 36; t++) Y[t] = 0.0;
            FS_TICK(3);
            asm volatile("" :: "v"(pflagsn));      // the flags load stays above this point (it would be sunk to its use)
            // sums in element order: one lane x0 + x1; a lane chained to its left neighbour (R_left + x2) + x3, ...
            // Lanes that carry two single-contribution blocks keep X and Y apart.  No selects: with m = 0 for lanes with two
            // blocks (else 1) and c = 1 for chained lanes (else 0), R = fma(m, Y, fma(c, R_left, X)) is x0 + x1 on the first lane
            // of a block, X alone on a lane with two blocks and (R_left + x2) + x3 on a chained lane -- fma(1, a, b) is a + b
            // to the bit.  A chained lane at position pos is final from iteration pos on (its left neighbour was final one
            // iteration earlier), what it holds before is never read.
            // last round: the next patch's coordinates are requested HERE, in front of the sums -- the vmcnt(0) between the
            // staging writes and the stores waits for them, and requested behind the sums they were still on their way
            // (strip 1.289 -> 1.260 ms; in front of the second evaluation the 24 registers they hold spill: 1.53 ms)
            if (last_round) {
                const double2 *xp = reinterpret_cast<const double2 *>(xyz + en_use * 12);
                for (int t = 0; t < 6; t++) { const double2 v = xp[t]; xn[2 * t] = v.x; xn[2 * t + 1] = v.y; }
            }
            const bool two = (fl & FS_FL_OUT1) != 0, chain = (fl & FS_FL_CHAIN) != 0;
            const int maxc = __builtin_amdgcn_readfirstlane((int)((fl >> 8) & 255u));
            const double mY = two ? 0.0 : 1.0, cT = chain ? 1.0 : 0.0;
            double R[36];                                       // output block 0; Y: output block 1 (lanes with two blocks)
            for (int t = 0; t < 36; t++) R[t] = FMA(mY, Y[t], X[t]);
            for (int it = 1; it < maxc; it++)
                for (int t = 0; t < 36; t++) R[t] = FMA(mY, Y[t], FMA(cT, row_shr1(R[t]), X[t]));
            // diagonal blocks: K = (K + K^T) / 2 of every element (MITC4.py:1034) taken on the SUM of the block's contributions
            // (the same number up to rounding, 15 means per block instead of per contribution; a diagonal block has
            // diagonal contributions only).  The mass blocks are symmetric as they are formed.
            if constexpr (OP == FS_OP_K) {
                if (a0 && FS_EV_TI(ev0) == FS_EV_TJ(ev0)) {
                    for (int a = 0; a < 6; a++)
                        for (int b = a + 1; b < 6; b++) {
                            const double m = 0.5 * (R[a * 6 + b] + R[b * 6 + a]);
                            R[a * 6 + b] = m; R[b * 6 + a] = m;
                        }
                }
                const bool dg1 = two && a1 && FS_EV_TI(ev1) == FS_EV_TJ(ev1);       // (a corner node's block on a two-block lane)
                if (__any(dg1)) {
                    if (dg1)
                        for (int a = 0; a < 6; a++)
                            for (int b = a + 1; b < 6; b++) {
                                const double m = 0.5 * (Y[a * 6 + b] + Y[b * 6 + a]);
                                Y[a * 6 + b] = m; Y[b * 6 + a] = m;
                            }
                }
            }
            FS_TICK(4);
            // ---- stores: through a wave-private LDS staging area.  Every block a lane emits goes to a slot (numbered by the
            // plan in address order of the DIRECT copy) and is read back twice: as 48-byte row pieces of block (I,J), one
            // store instruction per row k covering row k of 21 blocks with neighbouring lanes on neighbouring pieces, and
            // -- through a rank table in address order of the TRANSPOSED copies -- as the row pieces of block (J,I), taken
            // from the same slot with transposed indexing.  A lane that stored its own block from registers would touch
            // 36 different lines with 36 instructions.
            const bool out0 = (fl & FS_FL_OUT0) != 0;
            const int64_t dst0 = ((int64_t)(uint32_t)ra.x) | ((int64_t)ra.y << 32), dst1 = ((int64_t)(uint32_t)ra.z) | ((int64_t)ra.w << 32);
            const int64_t dst0t = ((int64_t)(uint32_t)rb4.x) | ((int64_t)rb4.y << 32), dst1t = ((int64_t)(uint32_t)rb4.z) | ((int64_t)rb4.w << 32);
            const int s0d = out0 ? (int)(slots & 255u) : 255;
            const int s0t = (out0 && dst0t >= 0) ? (int)((slots >> 8) & 255u) : 255;
            const int s1d = two ? (int)((slots >> 16) & 255u) : 255;
            const int s1t = (two && dst1t >= 0) ? (int)((slots >> 24) & 255u) : 255;
            const int nd = __popcll(__ballot(s0d != 255)) + __popcll(__ballot(s1d != 255));
            const int ntr = __popcll(__ballot(s0t != 255)) + __popcll(__ballot(s1t != 255));
            // staging slot = 21 x 16 bytes: the block's 18 pieces (row k, piece c at 3 k + c), (dst, stride), (dst_t,
            // stride_t), one spare; the odd slot size keeps the 8-lane groups of ds_write_b128 on distinct banks
            double2 *stg = s_stage + wave * (FZ_SLOT * SP);
            unsigned char *tab = s_tab + wave * 128;
            FS_TICK(14);
            for (int s0 = 0; s0 < nd; s0 += SP) {
                if (nd > SP) {      // several passes: ranks whose block is not staged in this pass must read as "none"
                    tab[lane] = 255; tab[lane + 64] = 255;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
                const int q0d = s0d - s0, q1d = s1d - s0;
                if (s0d != 255 && q0d >= 0 && q0d < SP) {
                    double2 *sp = stg + q0d * FZ_SLOT;
                    for (int t = 0; t < 18; t++) sp[t] = make_double2(R[2 * t], R[2 * t + 1]);
                    sp[18] = make_double2(__longlong_as_double(dst0), __longlong_as_double((long long)rc.x));
                    sp[19] = make_double2(__longlong_as_double(dst0t), __longlong_as_double((long long)rc.z));
                    if (s0t != 255) tab[s0t] = (unsigned char)q0d;
                }
                if (s1d != 255 && q1d >= 0 && q1d < SP) {
                    double2 *sp = stg + q1d * FZ_SLOT;
                    for (int t = 0; t < 18; t++) sp[t] = make_double2(Y[2 * t], Y[2 * t + 1]);
                    sp[18] = make_double2(__longlong_as_double(dst1), __longlong_as_double((long long)rc.y));
                    sp[19] = make_double2(__longlong_as_double(dst1t), __longlong_as_double((long long)rc.w));
                    if (s1t != 255) tab[s1t] = (unsigned char)q1d;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): every prefetch is in before the stores go out (see the loop top)
                FS_TICK(15);
                // direct copies: piece pc -> (block pc / 3, piece pc % 3)
                const int nhere = (nd - s0 < SP) ? nd - s0 : SP;
                for (int q = 0; q < (3 * SP + 63) / 64; q++) {
                    const int pc = q * 64 + lane;
                    const int bk = (pc * 21846) >> 16, cc = pc - 3 * bk;
                    if (pc < 3 * nhere) {
                        const double2 *sp = stg + bk * FZ_SLOT;
                        const double2 t = sp[18];
                        double *d = data + (int64_t)__double_as_longlong(t.x) + 2 * cc;
                        const int64_t str = (int64_t)__double_as_longlong(t.y);
                        double2 v[6];
                        for (int k = 0; k < 6; k++) v[k] = sp[k * 3 + cc];
                        for (int k = 0; k < 6; k++) *reinterpret_cast<double2 *>(d + k * str) = v[k];
                    }
                }
                // transposed copies, in address order of the copies: rank o -> slot tab[o]; piece (row k, columns 2 cc,
                // 2 cc + 1) of the transpose = entries (2 cc, k), (2 cc + 1, k) of the staged block
                int tslot[(3 * 128 + 63) / 64];
                for (int q = 0; q < (3 * 128 + 63) / 64; q++) {
                    const int pc = q * 64 + lane;
                    const int o = (pc * 21846) >> 16;
                    tslot[q] = (pc < 3 * ntr) ? (int)tab[o] : 255;
                }
                for (int q = 0; q < (3 * 128 + 63) / 64; q++) {
                    if (q * 64 >= 3 * ntr) break;
                    const int pc = q * 64 + lane;
                    const int o = (pc * 21846) >> 16, cc = pc - 3 * o;
                    if (tslot[q] != 255) {
                        const double2 *sp = stg + tslot[q] * FZ_SLOT;
                        const double2 t = sp[19];
                        double *d = data + (int64_t)__double_as_longlong(t.x) + 2 * cc;
                        const int64_t str = (int64_t)__double_as_longlong(t.y);
                        const double *sd = reinterpret_cast<const double *>(sp) + 12 * cc;
                        double2 v[6];
                        for (int k = 0; k < 6; k++) v[k] = make_double2(sd[k], sd[6 + k]);
                        for (int k = 0; k < 6; k++) *reinterpret_cast<double2 *>(d + k * str) = v[k];
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                FS_TICK(16);
            }
        }
        // status words of the patch's elements, one store instruction behind the copy-out (halo elements: every patch
        // writes the same value)
        if (OP == FS_OP_K && status && wave == 0) {
            const int st_ = s_st[lane];
            if (lane < ne && st_ >= 0) status[s_el[lane]] = st_;
        }
        FS_TICK(5);
#ifdef FS_FUSED_TIMING
        if (!more && lane == 0)
            for (int ph = 0; ph < 19; ph++) atomicAdd(&g_fused_t[ph], acc_[ph]);
#endif
        if (!more) break;
        __syncthreads();        // every wave is done with the image: stage 1 of the next patch may overwrite it
        p = pn; ne = nen; e = en_use; pid = pidn; pflags = pflagsn; r0 = r0n; r1 = r1n;
        e0n = e0nn; nen = nenn; r0n = r0nn; r1n = r1nn;
        for (int t = 0; t < 12; t++) (&x[0][0])[t] = xn[t];
        FS_TICK(6);
    }
}

// ------------------------------------------------------------------------------------------
// MITC4 total-Lagrangian tangent and internal force (MITC4.py:1067-1341, composite overrides
// MITC4_composite.py:499-621).  One lane per element; runs AFTER mitc4_k_kernel has written the
// linear blocks K into `kt` (FS_LAYOUT_SYMBLK):
//   f_int = K u + T^T( sum_g detJ h (B_L+B_NL)^T C_raw e_GL  -  K_m u_loc )     [nonlinear rows]
//         = K u                                                                 [linear rows]
//   K_T   = K + T^T sym( K_L + K_sigma ) T   -- only the translational 3x3 sub-blocks change.
// (K_bs + K_drill + K_c) u_loc of the reference is evaluated as K0 u_loc - K_m u_loc, with
// K_m = Y^T S Y from the same pre-contracted core as the stiffness kernel.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void mitc4_nl_kernel(const double *__restrict__ xyz,
                                                      const int32_t *__restrict__ prop_id,
                                                      const fs_shell_prop *__restrict__ props, int32_t n_props,
                                                      const double *__restrict__ ue, int64_t E,
                                                      double *__restrict__ kt, int want_kt,
                                                      double *__restrict__ fe, int tiled,
                                                      const double *__restrict__ sig_in) {
    // sig_in: optional [E][3] local membrane stress replacing the one computed from u (the `sigma` argument of
    // compute_tangent_stiffness, MITC4.py:1246-1262)
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    double x[4][3], u[4][6];
    for (int c = 0; c < 12; c++) (&x[0][0])[c] = xyz[e * 12 + c];
    for (int c = 0; c < 24; c++) (&u[0][0])[c] = ue[e * 24 + c];
    const fs_shell_prop *pr = props + safe_pid(prop_id[e], n_props);
    QuadGeom geo;
    quad_geometry(x, geo);
    const double R[3][3] = {{geo.e1[0], geo.e1[1], geo.e1[2]}, {geo.e2[0], geo.e2[1], geo.e2[2]}, {geo.e3[0], geo.e3[1], geo.e3[2]}};
    double *blkp[10];   // the element's 10 node-pair blocks (layout-dependent)
    for (int p = 0; p < 10; p++) blkp[p] = kt + quad_blk(tiled, e, p) * 36;
    const int pidx[4][4] = {{0, 1, 2, 3}, {1, 4, 5, 6}, {2, 5, 7, 8}, {3, 6, 8, 9}};

    {
#pragma clang fp contract(fast)
    // ---- f = K u with the linear blocks (global axes) ----
    double f[4][6];
    for (int c = 0; c < 24; c++) (&f[0][0])[c] = 0.0;
    if (fe) {
        for (int i = 0; i < 4; i++)
            for (int j = i; j < 4; j++) {
                const double *B = blkp[pidx[i][j]];
                for (int a = 0; a < 6; a++)
                    for (int b = 0; b < 6; b++) {
                        const double v = B[a * 6 + b];
                        f[i][a] += v * u[j][b];
                        if (i != j) f[j][b] += v * u[i][a];
                    }
            }
    }
    if (pr->nonlinear) {
        const double h = pr->h;
        double C[3][3];
        for (int a = 0; a < 9; a++) (&C[0][0])[a] = pr->C_raw[a];
        // local displacements
        double ul[4][3], rl[4][3];
        for (int i = 0; i < 4; i++)
            for (int a = 0; a < 3; a++) {
                ul[i][a] = R[a][0] * u[i][0] + R[a][1] * u[i][1] + R[a][2] * u[i][2];
                rl[i][a] = R[a][0] * u[i][3] + R[a][1] * u[i][4] + R[a][2] * u[i][5];
            }
        // membrane stress at the centre (MITC4.py:1227-1240; laminate MITC4_composite.py:499-508)
        double Sm00, Sm01, Sm11;
        {
            double dxi[4], deta[4], ia, ib, ic, id;
            dshape(0.0, 0.0, dxi, deta);
            jacobian(geo, dxi, deta, ia, ib, ic, id);
            double eps[3] = {0, 0, 0}, kap[3] = {0, 0, 0};
            for (int i = 0; i < 4; i++) {
                const double nx = ia * dxi[i] + ib * deta[i], ny = ic * dxi[i] + id * deta[i];
                eps[0] += nx * ul[i][0]; eps[1] += ny * ul[i][1]; eps[2] += ny * ul[i][0] + nx * ul[i][1];
                kap[0] += nx * rl[i][1]; kap[1] += -ny * rl[i][0]; kap[2] += ny * rl[i][1] - nx * rl[i][0];
            }
            double sig[3];
            if (pr->flags & FS_PROP_LAMINATE) {
                for (int a = 0; a < 3; a++)
                    sig[a] = ((pr->A[a * 3] * eps[0] + pr->A[a * 3 + 1] * eps[1] + pr->A[a * 3 + 2] * eps[2]) +
                              (pr->B[a * 3] * kap[0] + pr->B[a * 3 + 1] * kap[1] + pr->B[a * 3 + 2] * kap[2])) / h;
            } else {
                for (int a = 0; a < 3; a++) sig[a] = C[a][0] * eps[0] + C[a][1] * eps[1] + C[a][2] * eps[2];
            }
            if (sig_in) { sig[0] = sig_in[e * 3]; sig[1] = sig_in[e * 3 + 1]; sig[2] = sig_in[e * 3 + 2]; }
            Sm00 = sig[0] * h; Sm11 = sig[1] * h; Sm01 = sig[2] * h;
        }
        // tying vectors (as in mitc4_k_kernel) for K_m u_loc
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
        double Sf[5][5];
        for (int t = 0; t < 25; t++) (&Sf[0][0])[t] = 0.0;
        double dL[10][3][3];   // K_L + K_sigma, translational sub-blocks of pairs i<=j (local axes)
        for (int t = 0; t < 90; t++) (&dL[0][0][0])[t] = 0.0;
        double fm[4][3];       // membrane internal force (local axes)
        for (int t = 0; t < 12; t++) (&fm[0][0])[t] = 0.0;
        for (int g = 0; g < 4; g++) {
            const double xi = (g == 1 || g == 2) ? kGP : -kGP, eta = (g >= 2) ? kGP : -kGP;
            double dxi[4], deta[4], ia, ib, ic, id;
            dshape(xi, eta, dxi, deta);
            const double detJ = jacobian(geo, dxi, deta, ia, ib, ic, id);
            double nx[4], ny[4];
            for (int i = 0; i < 4; i++) { nx[i] = ia * dxi[i] + ib * deta[i]; ny[i] = ic * dxi[i] + id * deta[i]; }
            double W[3][5];
            membrane_W(geo, xi, eta, detJ, ia, ib, ic, id, W);
            for (int t = 0; t < 5; t++)
                for (int a = 0; a < 3; a++) {
                    const double cw = detJ * (pr->A[a * 3] * W[0][t] + pr->A[a * 3 + 1] * W[1][t] + pr->A[a * 3 + 2] * W[2][t]);
                    for (int v = 0; v < 5; v++) Sf[v][t] += W[a][v] * cw;
                }
            // displacement gradient: GLOBAL components, local derivatives (MITC4.py:1092-1107)
            double H[3][2];
            for (int c = 0; c < 3; c++) {
                H[c][0] = dots4(u[0][c], u[1][c], u[2][c], u[3][c], nx[0], nx[1], nx[2], nx[3]);
                H[c][1] = dots4(u[0][c], u[1][c], u[2][c], u[3][c], ny[0], ny[1], ny[2], ny[3]);
            }
            // Green-Lagrange membrane strain (third column of H is zero)
            const double E00 = 0.5 * (2.0 * H[0][0] + (H[0][0] * H[0][0] + H[1][0] * H[1][0] + H[2][0] * H[2][0]));
            const double E11 = 0.5 * (2.0 * H[1][1] + (H[0][1] * H[0][1] + H[1][1] * H[1][1] + H[2][1] * H[2][1]));
            const double E01 = 0.5 * ((H[0][1] + H[1][0]) + (H[0][0] * H[0][1] + H[1][0] * H[1][1] + H[2][0] * H[2][1]));
            const double eg[3] = {E00, E11, 2.0 * E01};
            double sg[3];
            for (int a = 0; a < 3; a++) sg[a] = C[a][0] * eg[0] + C[a][1] * eg[1] + C[a][2] * eg[2];
            const double wh = detJ * h;
            double Bt[4][3][3], BL[4][3][3], CBt[4][3][3], CBL[4][3][3];
            for (int i = 0; i < 4; i++) {
                for (int k = 0; k < 3; k++) {
                    BL[i][0][k] = (k == 0) ? nx[i] : 0.0;
                    BL[i][1][k] = (k == 1) ? ny[i] : 0.0;
                    BL[i][2][k] = (k == 0) ? ny[i] : ((k == 1) ? nx[i] : 0.0);
                    Bt[i][0][k] = BL[i][0][k] + H[k][0] * nx[i];
                    Bt[i][1][k] = BL[i][1][k] + H[k][1] * ny[i];
                    Bt[i][2][k] = BL[i][2][k] + (H[k][0] * ny[i] + H[k][1] * nx[i]);
                }
                for (int a = 0; a < 3; a++)
                    for (int k = 0; k < 3; k++) {
                        CBt[i][a][k] = wh * (C[a][0] * Bt[i][0][k] + C[a][1] * Bt[i][1][k] + C[a][2] * Bt[i][2][k]);
                        CBL[i][a][k] = wh * (C[a][0] * BL[i][0][k] + C[a][1] * BL[i][1][k] + C[a][2] * BL[i][2][k]);
                    }
                for (int k = 0; k < 3; k++)
                    fm[i][k] += wh * (Bt[i][0][k] * sg[0] + Bt[i][1][k] * sg[1] + Bt[i][2][k] * sg[2]);
            }
            if (want_kt) {
                for (int i = 0; i < 4; i++)
                    for (int j = i; j < 4; j++) {
                        const int p = pidx[i][j];
                        const double ks = detJ * (nx[i] * (Sm00 * nx[j] + Sm01 * ny[j]) + ny[i] * (Sm01 * nx[j] + Sm11 * ny[j]));
                        for (int k = 0; k < 3; k++)
                            for (int l = 0; l < 3; l++) {
                                double v = Bt[i][0][k] * CBt[j][0][l] + Bt[i][1][k] * CBt[j][1][l] + Bt[i][2][k] * CBt[j][2][l];
                                v -= BL[i][0][k] * CBL[j][0][l] + BL[i][1][k] * CBL[j][1][l] + BL[i][2][k] * CBL[j][2][l];
                                if (k == l) v += ks;
                                dL[p][k][l] += v;
                            }
                    }
            }
        }
        // f_loc = f_memb - K_m u_loc ;  K_m u_loc = Y^T S (Y u_loc)
        if (fe) {
            double Y[4][5][3];
            for (int i = 0; i < 4; i++) {
                double w[6];
                tying_weights(i, w);
                for (int k = 0; k < 3; k++) {
                    for (int t = 0; t < 4; t++) Y[i][t][k] = w[t] * gt[t][k];
                    Y[i][4][k] = w[4] * gt[4][k] + w[5] * gt[5][k];
                }
            }
            double z[5], Sz[5];
            for (int t = 0; t < 5; t++) {
                double v = 0.0;
                for (int i = 0; i < 4; i++)
                    for (int k = 0; k < 3; k++) v += Y[i][t][k] * ul[i][k];
                z[t] = v;
            }
            for (int t = 0; t < 5; t++) Sz[t] = Sf[t][0] * z[0] + Sf[t][1] * z[1] + Sf[t][2] * z[2] + Sf[t][3] * z[3] + Sf[t][4] * z[4];
            for (int i = 0; i < 4; i++) {
                double fl[3];
                for (int k = 0; k < 3; k++) {
                    double v = fm[i][k];
                    for (int t = 0; t < 5; t++) v -= Y[i][t][k] * Sz[t];
                    fl[k] = v;
                }
                for (int a = 0; a < 3; a++) f[i][a] += R[0][a] * fl[0] + R[1][a] * fl[1] + R[2][a] * fl[2];
            }
        }
        // K_T = K + R^T sym(dL) R on the translational sub-blocks
        if (want_kt) {
            for (int i = 0; i < 4; i++)
                for (int j = i; j < 4; j++) {
                    const int p = pidx[i][j];
                    double Xs[3][3];
                    for (int k = 0; k < 3; k++)
                        for (int l = 0; l < 3; l++) Xs[k][l] = (i == j) ? 0.5 * (dL[p][k][l] + dL[p][l][k]) : dL[p][k][l];
                    double Yt[3][3];
                    for (int a = 0; a < 3; a++)
                        for (int l = 0; l < 3; l++) Yt[a][l] = R[0][a] * Xs[0][l] + R[1][a] * Xs[1][l] + R[2][a] * Xs[2][l];
                    for (int a = 0; a < 3; a++)
                        for (int b = 0; b < 3; b++) {
                            const double v = Yt[a][0] * R[0][b] + Yt[a][1] * R[1][b] + Yt[a][2] * R[2][b];
                            blkp[p][a * 6 + b] += v;
                        }
                    if (i == j) {  // keep the diagonal block exactly symmetric
                        for (int a = 0; a < 3; a++)
                            for (int b = a + 1; b < 3; b++) {
                                const double m = 0.5 * (blkp[p][a * 6 + b] + blkp[p][b * 6 + a]);
                                blkp[p][a * 6 + b] = m; blkp[p][b * 6 + a] = m;
                            }
                    }
                }
        }
    }
    if (fe)
        for (int c = 0; c < 24; c++) fe[e * 24 + c] = (&f[0][0])[c];
    }
}

// ------------------------------------------------------------------------------------------
// Tiled-layout form of mitc4_nl_kernel (FS_LAYOUT_SYMBLK_T16).  The per-element form above walks its
// ten 288-byte blocks with per-lane strided 8-byte accesses (5.1 ms per 1.25 M elements for K_T + f_int);
// here a wave owns a 16-element tile like mitc4_k_kernel:
//   * lane = (element, Gauss point): the Gauss-point work (H, Green-Lagrange strain, B_L + B_NL, K_L + K_sigma
//     contributions, membrane force, membrane core S) is split over the four lanes of an element and summed
//     with DPP quad reductions; the 10 x 3 x 3 tangent corrections go to LDS;
//   * the tile's 160 blocks stream through the LDS staging area in the order mitc4_k_kernel wrote them (whole
//     128-byte lines in, whole lines out); the lane owning a node-pair task adds R^T sym(dL) R to the
//     translational sub-block and accumulates K u for its two nodes in LDS (one writer per node and step).
// Same formulas as mitc4_nl_kernel; sums over Gauss points and over blocks associate differently (eps level).
// ------------------------------------------------------------------------------------------
constexpr int NLT_STG = 0;                          // staging, 32 slots
constexpr int NLT_DL = NLT_STG + 32 * STG_SLOT;     // dL[p][k][l][Q]: 90 x 16
constexpr int NLT_F = NLT_DL + 90 * 16;             // f[node][a][Q]: 24 x 16
constexpr int NLT_LDS = NLT_F + 24 * 16;

template <bool STREAM>
__global__ __launch_bounds__(64) void mitc4_nl_tiled_kernel(const double *__restrict__ xyz,
                                                            const int32_t *__restrict__ prop_id,
                                                            const fs_shell_prop *__restrict__ props, int32_t n_props,
                                                            const double *__restrict__ ue, int64_t E,
                                                            const double *k0, double *kt, int want_kt,
                                                            double *__restrict__ fe,
                                                            const int64_t *__restrict__ conn,
                                                            const double *__restrict__ u_nodes) {
    // k0: the linear blocks K (read); kt: K_T (written when want_kt).  They may be the same buffer (in place).
    // STREAM == false (fs_shell_quad_fint_nl): no blocks are streamed and no LDS is held (more waves per CU), fe receives
    // the NONLINEAR part of the internal forces only (the caller adds K u from the assembled matrix).  conn / u_nodes:
    // gather the nodal displacements through the connectivity instead of reading ue.
    __shared__ double lds[STREAM ? NLT_LDS : 16];
    const int lane = threadIdx.x;
    const int Q = lane >> 2, g = lane & 3;
    const int64_t e_raw = (int64_t)blockIdx.x * 16 + Q;
    const bool valid = e_raw < E;
    const int64_t e = valid ? e_raw : (E - 1);
    double x[4][3], u[4][6];
    {
        const double2 *p = reinterpret_cast<const double2 *>(xyz + e * 12);
        for (int t = 0; t < 6; t++) { const double2 v = p[t]; (&x[0][0])[2 * t] = v.x; (&x[0][0])[2 * t + 1] = v.y; }
        if (conn) {
            const longlong2 *cp = reinterpret_cast<const longlong2 *>(conn + e * 4);
            const longlong2 c01 = cp[0], c23 = cp[1];
            const int64_t nid[4] = {c01.x, c01.y, c23.x, c23.y};
            for (int i = 0; i < 4; i++) {
                const double2 *q = reinterpret_cast<const double2 *>(u_nodes + nid[i] * 6);
                for (int t = 0; t < 3; t++) { const double2 v = q[t]; u[i][2 * t] = v.x; u[i][2 * t + 1] = v.y; }
            }
        } else {
            const double2 *q = reinterpret_cast<const double2 *>(ue + e * 24);
            for (int t = 0; t < 12; t++) { const double2 v = q[t]; (&u[0][0])[2 * t] = v.x; (&u[0][0])[2 * t + 1] = v.y; }
        }
    }
    const fs_shell_prop *pr = props + safe_pid(prop_id[e], n_props);
    QuadGeom geo;
    quad_geometry(x, geo);
    const double R[3][3] = {{geo.e1[0], geo.e1[1], geo.e1[2]}, {geo.e2[0], geo.e2[1], geo.e2[2]}, {geo.e3[0], geo.e3[1], geo.e3[2]}};
    const bool nonlinear = pr->nonlinear != 0;
    const int pidx[4][4] = {{0, 1, 2, 3}, {1, 4, 5, 6}, {2, 5, 7, 8}, {3, 6, 8, 9}};
    double fnl[3] = {0.0, 0.0, 0.0};    // nonlinear part of f for node g (global axes)
    {
#pragma clang fp contract(fast)
        if (STREAM && want_kt)
            for (int t = 0; t < 90; t++) if ((t & 3) == g) lds[NLT_DL + (t << 4) + Q] = 0.0;
        if constexpr (STREAM)
            for (int t = 0; t < 6; t++) lds[NLT_F + ((g * 6 + t) << 4) + Q] = 0.0;
        if (nonlinear) {
            const double h = pr->h;
            double C[3][3];
            for (int a = 0; a < 9; a++) (&C[0][0])[a] = pr->C_raw[a];
            double ul[4][3], rl[4][3];
            for (int i = 0; i < 4; i++)
                for (int a = 0; a < 3; a++) {
                    ul[i][a] = R[a][0] * u[i][0] + R[a][1] * u[i][1] + R[a][2] * u[i][2];
                    rl[i][a] = R[a][0] * u[i][3] + R[a][1] * u[i][4] + R[a][2] * u[i][5];
                }
            double Sm00, Sm01, Sm11;   // membrane stress at the centre (MITC4.py:1227-1240; MITC4_composite.py:499-508)
            {
                double dxi[4], deta[4], ia, ib, ic, id;
                dshape(0.0, 0.0, dxi, deta);
                jacobian(geo, dxi, deta, ia, ib, ic, id);
                double eps[3] = {0, 0, 0}, kap[3] = {0, 0, 0};
                for (int i = 0; i < 4; i++) {
                    const double nx = ia * dxi[i] + ib * deta[i], ny = ic * dxi[i] + id * deta[i];
                    eps[0] += nx * ul[i][0]; eps[1] += ny * ul[i][1]; eps[2] += ny * ul[i][0] + nx * ul[i][1];
                    kap[0] += nx * rl[i][1]; kap[1] += -ny * rl[i][0]; kap[2] += ny * rl[i][1] - nx * rl[i][0];
                }
                double sig[3];
                if (pr->flags & FS_PROP_LAMINATE) {
                    for (int a = 0; a < 3; a++)
                        sig[a] = ((pr->A[a * 3] * eps[0] + pr->A[a * 3 + 1] * eps[1] + pr->A[a * 3 + 2] * eps[2]) +
                                  (pr->B[a * 3] * kap[0] + pr->B[a * 3 + 1] * kap[1] + pr->B[a * 3 + 2] * kap[2])) / h;
                } else {
                    for (int a = 0; a < 3; a++) sig[a] = C[a][0] * eps[0] + C[a][1] * eps[1] + C[a][2] * eps[2];
                }
                Sm00 = sig[0] * h; Sm11 = sig[1] * h; Sm01 = sig[2] * h;
            }
            // ---- Gauss point g of this lane ----
            const double xi = (g == 1 || g == 2) ? kGP : -kGP, eta = (g >= 2) ? kGP : -kGP;
            double dxi[4], deta[4], ia, ib, ic, id;
            dshape(xi, eta, dxi, deta);
            const double detJ = jacobian(geo, dxi, deta, ia, ib, ic, id);
            double nx[4], ny[4];
            for (int i = 0; i < 4; i++) { nx[i] = ia * dxi[i] + ib * deta[i]; ny[i] = ic * dxi[i] + id * deta[i]; }
            double H[3][2];   // GLOBAL displacement components, local derivatives (MITC4.py:1092-1107)
            for (int c = 0; c < 3; c++) {
                H[c][0] = dots4(u[0][c], u[1][c], u[2][c], u[3][c], nx[0], nx[1], nx[2], nx[3]);
                H[c][1] = dots4(u[0][c], u[1][c], u[2][c], u[3][c], ny[0], ny[1], ny[2], ny[3]);
            }
            const double E00 = 0.5 * (2.0 * H[0][0] + (H[0][0] * H[0][0] + H[1][0] * H[1][0] + H[2][0] * H[2][0]));
            const double E11 = 0.5 * (2.0 * H[1][1] + (H[0][1] * H[0][1] + H[1][1] * H[1][1] + H[2][1] * H[2][1]));
            const double E01 = 0.5 * ((H[0][1] + H[1][0]) + (H[0][0] * H[0][1] + H[1][0] * H[1][1] + H[2][0] * H[2][1]));
            const double eg[3] = {E00, E11, 2.0 * E01};
            double sg[3];
            for (int a = 0; a < 3; a++) sg[a] = C[a][0] * eg[0] + C[a][1] * eg[1] + C[a][2] * eg[2];
            const double wh = detJ * h;
            double Bt[4][3][3], BL[4][3][3], CBt[4][3][3], CBL[4][3][3], fm[4][3];
            for (int i = 0; i < 4; i++) {
                for (int k = 0; k < 3; k++) {
                    BL[i][0][k] = (k == 0) ? nx[i] : 0.0;
                    BL[i][1][k] = (k == 1) ? ny[i] : 0.0;
                    BL[i][2][k] = (k == 0) ? ny[i] : ((k == 1) ? nx[i] : 0.0);
                    Bt[i][0][k] = BL[i][0][k] + H[k][0] * nx[i];
                    Bt[i][1][k] = BL[i][1][k] + H[k][1] * ny[i];
                    Bt[i][2][k] = BL[i][2][k] + (H[k][0] * ny[i] + H[k][1] * nx[i]);
                }
                for (int a = 0; a < 3; a++)
                    for (int k = 0; k < 3; k++) {
                        CBt[i][a][k] = wh * (C[a][0] * Bt[i][0][k] + C[a][1] * Bt[i][1][k] + C[a][2] * Bt[i][2][k]);
                        CBL[i][a][k] = wh * (C[a][0] * BL[i][0][k] + C[a][1] * BL[i][1][k] + C[a][2] * BL[i][2][k]);
                    }
                for (int k = 0; k < 3; k++)
                    fm[i][k] = quad_sum(wh * (Bt[i][0][k] * sg[0] + Bt[i][1][k] * sg[1] + Bt[i][2][k] * sg[2]));
            }
            if (STREAM && want_kt) {   // K_L + K_sigma, translational sub-blocks of the pairs i <= j (local axes) -> LDS
                for (int i = 0; i < 4; i++)
                    for (int j = i; j < 4; j++) {
                        const int p = pidx[i][j];
                        const double ks = detJ * (nx[i] * (Sm00 * nx[j] + Sm01 * ny[j]) + ny[i] * (Sm01 * nx[j] + Sm11 * ny[j]));
                        for (int k = 0; k < 3; k++)
                            for (int l = 0; l < 3; l++) {
                                double v = Bt[i][0][k] * CBt[j][0][l] + Bt[i][1][k] * CBt[j][1][l] + Bt[i][2][k] * CBt[j][2][l];
                                v -= BL[i][0][k] * CBL[j][0][l] + BL[i][1][k] * CBL[j][1][l] + BL[i][2][k] * CBL[j][2][l];
                                if (k == l) v += ks;
                                v = quad_sum(v);
                                const int t = p * 9 + k * 3 + l;
                                if ((t & 3) == g) lds[NLT_DL + (t << 4) + Q] = v;
                            }
                    }
            }
            if (fe) {   // f_loc = f_memb - K_m u_loc, K_m u_loc = Y^T S (Y u_loc); lane g keeps node g
                double W[3][5], Sf[5][5];
                membrane_W(geo, xi, eta, detJ, ia, ib, ic, id, W);
                for (int t = 0; t < 25; t++) (&Sf[0][0])[t] = 0.0;
                for (int t = 0; t < 5; t++)
                    for (int a = 0; a < 3; a++) {
                        const double cw = detJ * (pr->A[a * 3] * W[0][t] + pr->A[a * 3 + 1] * W[1][t] + pr->A[a * 3 + 2] * W[2][t]);
                        for (int v = 0; v < 5; v++) Sf[v][t] += W[a][v] * cw;
                    }
                for (int t = 0; t < 25; t++) (&Sf[0][0])[t] = quad_sum((&Sf[0][0])[t]);
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
                double z[5] = {0, 0, 0, 0, 0}, Yg[5][3];
                for (int i = 0; i < 4; i++) {
                    double w[6];
                    tying_weights(i, w);
                    for (int k = 0; k < 3; k++) {
                        double y[5];
                        for (int t = 0; t < 4; t++) y[t] = w[t] * gt[t][k];
                        y[4] = w[4] * gt[4][k] + w[5] * gt[5][k];
                        for (int t = 0; t < 5; t++) {
                            z[t] += y[t] * ul[i][k];
                            if (i == g) Yg[t][k] = y[t];
                        }
                    }
                }
                double Sz[5];
                for (int t = 0; t < 5; t++) Sz[t] = Sf[t][0] * z[0] + Sf[t][1] * z[1] + Sf[t][2] * z[2] + Sf[t][3] * z[3] + Sf[t][4] * z[4];
                double fl[3];
                for (int k = 0; k < 3; k++) {
                    double v = 0.0;
                    for (int i = 0; i < 4; i++) v = (i == g) ? fm[i][k] : v;
                    for (int t = 0; t < 5; t++) v -= Yg[t][k] * Sz[t];
                    fl[k] = v;
                }
                for (int a = 0; a < 3; a++) fnl[a] = R[0][a] * fl[0] + R[1][a] * fl[1] + R[2][a] * fl[2];
            }
        }
    }
    if constexpr (STREAM) __syncthreads();
    // ---- stream the tile: K u (linear blocks) and K_T = K + R^T sym(dL) R ----
    for (int round = 0; round < (STREAM ? 3 : 0); round++) {
#pragma clang fp contract(fast)
        int ti, tj;
        if (round == 0) { ti = g; tj = g; }
        else if (round == 1) { ti = (g == 3) ? 0 : g; tj = (g == 3) ? 3 : g + 1; }
        else { ti = g & 1; tj = (g & 1) + 2; }
        const int p = ti * 4 - (ti * (ti - 1)) / 2 + (tj - ti);
        const int npass = (round == 2) ? 1 : 2;
        for (int h = 0; h < npass; h++) {
            const int64_t toff = ((int64_t)blockIdx.x * 160 + round * 64 + h * 32) * 36;
            const double2 *gin = reinterpret_cast<const double2 *>(k0 + toff) + lane;
            double2 *gp = reinterpret_cast<double2 *>(kt + toff) + lane;
            double2 *st2 = reinterpret_cast<double2 *>(lds + NLT_STG);
            for (int i = 0; i < 9; i++) {
                const int c = i * 64 + lane;
                const int bl = (c * 3641) >> 16;        // c / 18 for c < 576
                st2[c + bl] = gin[i * 64];
            }
            __syncthreads();
            const bool mine = (round == 2) ? (g < 2) : ((lane >> 5) == h);
            const int slot = (round == 2) ? ((g << 4) + Q) : ((g << 3) + (Q & 7));
            double G[6][6];
            double *sp = lds + NLT_STG + slot * STG_SLOT;
            if (mine) {
                for (int t = 0; t < 36; t++) (&G[0][0])[t] = sp[t];
                if (want_kt && nonlinear) {
                    double Xs[3][3];
                    for (int k = 0; k < 3; k++)
                        for (int l = 0; l < 3; l++) Xs[k][l] = lds[NLT_DL + ((p * 9 + k * 3 + l) << 4) + Q];
                    if (ti == tj)
                        for (int k = 0; k < 3; k++)
                            for (int l = k + 1; l < 3; l++) { const double m = 0.5 * (Xs[k][l] + Xs[l][k]); Xs[k][l] = m; Xs[l][k] = m; }
                    double Yt[3][3];
                    for (int a = 0; a < 3; a++)
                        for (int l = 0; l < 3; l++) Yt[a][l] = R[0][a] * Xs[0][l] + R[1][a] * Xs[1][l] + R[2][a] * Xs[2][l];
                    double T[3][3];
                    for (int a = 0; a < 3; a++)
                        for (int b = 0; b < 3; b++) T[a][b] = G[a][b] + (Yt[a][0] * R[0][b] + Yt[a][1] * R[1][b] + Yt[a][2] * R[2][b]);
                    if (ti == tj)
                        for (int a = 0; a < 3; a++)
                            for (int b = a + 1; b < 3; b++) { const double m = 0.5 * (T[a][b] + T[b][a]); T[a][b] = m; T[b][a] = m; }
                    for (int a = 0; a < 3; a++)
                        for (int b = 0; b < 3; b++) sp[a * 6 + b] = T[a][b];
                }
            }
            if (fe) {   // f_i += G u_j (step 0), f_j += G^T u_i (step 1, off-diagonal tasks); one writer per node
                for (int step = 0; step < 2; step++) {
                    // round 1: lanes 0..2 give node ti first, lane 3 (pair 0-3) gives node tj = 3 first
                    const bool first = (round == 1 && g == 3) ? (step == 1) : (step == 0);
                    if (mine && !(ti == tj && step == 1)) {
                        const int node = first ? ti : tj, other = first ? tj : ti;
                        double uo[6];
                        for (int b = 0; b < 6; b++) {
                            double v = 0.0;
                            for (int n = 0; n < 4; n++) v = (n == other) ? u[n][b] : v;
                            uo[b] = v;
                        }
                        for (int a = 0; a < 6; a++) {
                            double v = lds[NLT_F + ((node * 6 + a) << 4) + Q];
                            for (int b = 0; b < 6; b++) v += (first ? G[a][b] : G[b][a]) * uo[b];
                            lds[NLT_F + ((node * 6 + a) << 4) + Q] = v;
                        }
                    }
                    __syncthreads();
                }
            } else {
                __syncthreads();
            }
            if (want_kt)
                for (int i = 0; i < 9; i++) {
                    const int c = i * 64 + lane;
                    const int bl = (c * 3641) >> 16;
                    gp[i * 64] = st2[c + bl];
                }
            __syncthreads();
        }
    }
    if (fe && valid) {
        for (int a = 0; a < 6; a++) {
            const double v = (STREAM ? lds[NLT_F + ((g * 6 + a) << 4) + Q] : 0.0) + ((a < 3) ? fnl[a] : 0.0);
            fe[e * 24 + 6 * g + a] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------
// MITC4 consistent mass (MITC4.py:1038-1061), FS_LAYOUT_SYMBLK: one thread per (element, pair)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mitc4_m_kernel(const double *__restrict__ xyz,
                                                      const int32_t *__restrict__ prop_id,
                                                      const fs_shell_prop *__restrict__ props, int32_t n_props,
                                                      int64_t E, double *__restrict__ out, int tiled) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= E * 10) return;
    const int64_t e = t / 10;
    const int p = (int)(t - e * 10);
    const int pi[10] = {0, 0, 0, 0, 1, 1, 1, 2, 2, 3}, pj[10] = {0, 1, 2, 3, 1, 2, 3, 2, 3, 3};
    const int ti = pi[p], tj = pj[p];
    double x[4][3];
    for (int c = 0; c < 12; c++) (&x[0][0])[c] = xyz[e * 12 + c];
    QuadGeom geo;
    quad_geometry(x, geo);
    const fs_shell_prop *pr = props + safe_pid(prop_id[e], n_props);
    const double mt = pr->m_t, mr = pr->m_r;
    const double sxi = (ti == 1 || ti == 2) ? 1.0 : -1.0, syi = (ti >= 2) ? 1.0 : -1.0;
    const double sxj = (tj == 1 || tj == 2) ? 1.0 : -1.0, syj = (tj >= 2) ? 1.0 : -1.0;
    double ct = 0.0, cr = 0.0;
    for (int g = 0; g < 4; g++) {
        const double xi = (g == 1 || g == 2) ? kGP : -kGP, eta = (g >= 2) ? kGP : -kGP;
        double dxi[4], deta[4], ia, ib, ic, id;
        dshape(xi, eta, dxi, deta);
        const double detJ = jacobian(geo, dxi, deta, ia, ib, ic, id);
        const double Ni = 0.25 * (1.0 + sxi * xi) * (1.0 + syi * eta);
        const double Nj = 0.25 * (1.0 + sxj * xi) * (1.0 + syj * eta);
        ct = ct + Ni * Nj * mt * 1.0 * detJ;  // val_t = N[i]*N[j]*m_trans*w*detJ
        cr = cr + Ni * Nj * mr * 1.0 * detJ;
    }
    const double R[3][3] = {{geo.e1[0], geo.e1[1], geo.e1[2]}, {geo.e2[0], geo.e2[1], geo.e2[2]}, {geo.e3[0], geo.e3[1], geo.e3[2]}};
    double G[6][6];
    for (int a = 0; a < 6; a++)
        for (int b = 0; b < 6; b++) G[a][b] = 0.0;
    for (int p3 = 0; p3 < 3; p3++)
        for (int q = 0; q < 3; q++) {
            // T^T (c I) T evaluated as ((R^T c) R): sum_k (R[k][p] c) R[k][q]
            double st = 0.0, sr = 0.0;
            for (int k = 0; k < 3; k++) {
                st = FMA(R[k][p3] * ct, R[k][q], st);
                sr = FMA(R[k][p3] * cr, R[k][q], sr);
            }
            G[p3][q] = st;
            G[3 + p3][3 + q] = sr;
        }
    double2 *dst = reinterpret_cast<double2 *>(out + quad_blk(tiled, e, p) * 36);
    for (int c = 0; c < 18; c++) dst[c] = make_double2((&G[0][0])[2 * c], (&G[0][0])[2 * c + 1]);
}

// Tiled-layout mass kernel (FS_LAYOUT_SYMBLK_T16): one wave per 16-element tile, lane = (element, task lane)
// as in mitc4_k_kernel, the same three task rounds and the same LDS-staged whole-line stores -- the mass
// blocks are cheap (8 kflop per element), the kernel is bound by the 2 880 B per element it writes, and the
// thread-per-pair form above spends most of its time on scattered 16-byte pieces.  Same arithmetic per
// block as mitc4_m_kernel (bitwise identical values).
__global__ __launch_bounds__(64) void mitc4_m_tiled_kernel(const double *__restrict__ xyz,
                                                           const int32_t *__restrict__ prop_id,
                                                           const fs_shell_prop *__restrict__ props, int32_t n_props,
                                                           int64_t E, double *__restrict__ out) {
    __shared__ double lds[32 * STG_SLOT];
    const int lane = threadIdx.x;
    const int Q = lane >> 2, g = lane & 3;
    const int64_t e_raw = (int64_t)blockIdx.x * 16 + Q;
    const int64_t e = (e_raw < E) ? e_raw : (E - 1);
    double x[4][3];
    {
        const double2 *p = reinterpret_cast<const double2 *>(xyz + e * 12);
        for (int t = 0; t < 6; t++) {
            const double2 v = p[t];
            (&x[0][0])[2 * t] = v.x;
            (&x[0][0])[2 * t + 1] = v.y;
        }
    }
    QuadGeom geo;
    quad_geometry(x, geo);
    const fs_shell_prop *pr = props + safe_pid(prop_id[e], n_props);
    const double mt = pr->m_t, mr = pr->m_r;
    double det[4];
    for (int q = 0; q < 4; q++) {
        const double xi = (q == 1 || q == 2) ? kGP : -kGP, eta = (q >= 2) ? kGP : -kGP;
        double dxi[4], deta[4], ia, ib, ic, id;
        dshape(xi, eta, dxi, deta);
        det[q] = jacobian(geo, dxi, deta, ia, ib, ic, id);
    }
    const double R[3][3] = {{geo.e1[0], geo.e1[1], geo.e1[2]}, {geo.e2[0], geo.e2[1], geo.e2[2]}, {geo.e3[0], geo.e3[1], geo.e3[2]}};
    for (int round = 0; round < 3; round++) {
        int ti, tj;
        if (round == 0) { ti = g; tj = g; }
        else if (round == 1) { ti = (g == 3) ? 0 : g; tj = (g == 3) ? 3 : g + 1; }
        else { ti = g & 1; tj = (g & 1) + 2; }
        const double sxi = (ti == 1 || ti == 2) ? 1.0 : -1.0, syi = (ti >= 2) ? 1.0 : -1.0;
        const double sxj = (tj == 1 || tj == 2) ? 1.0 : -1.0, syj = (tj >= 2) ? 1.0 : -1.0;
        double ct = 0.0, cr = 0.0;
        for (int q = 0; q < 4; q++) {
            const double xi = (q == 1 || q == 2) ? kGP : -kGP, eta = (q >= 2) ? kGP : -kGP;
            const double Ni = 0.25 * (1.0 + sxi * xi) * (1.0 + syi * eta);
            const double Nj = 0.25 * (1.0 + sxj * xi) * (1.0 + syj * eta);
            ct = ct + Ni * Nj * mt * 1.0 * det[q];
            cr = cr + Ni * Nj * mr * 1.0 * det[q];
        }
        double G[6][6];
        for (int a = 0; a < 6; a++)
            for (int b = 0; b < 6; b++) G[a][b] = 0.0;
        for (int p3 = 0; p3 < 3; p3++)
            for (int q = 0; q < 3; q++) {
                double st = 0.0, sr = 0.0;
                for (int k = 0; k < 3; k++) {
                    st = FMA(R[k][p3] * ct, R[k][q], st);
                    sr = FMA(R[k][p3] * cr, R[k][q], sr);
                }
                G[p3][q] = st;
                G[3 + p3][3 + q] = sr;
            }
        const int npass = (round == 2) ? 1 : 2;
        for (int h = 0; h < npass; h++) {
            const bool mine = (round == 2) ? (g < 2) : ((lane >> 5) == h);
            const int slot = (round == 2) ? ((g << 4) + Q) : ((g << 3) + (Q & 7));
            if (mine) {
                double2 *sp = reinterpret_cast<double2 *>(lds + slot * STG_SLOT);
                for (int t = 0; t < 18; t++) sp[t] = make_double2((&G[0][0])[2 * t], (&G[0][0])[2 * t + 1]);
            }
            __syncthreads();
            double2 *gp = reinterpret_cast<double2 *>(out + ((int64_t)blockIdx.x * 160 + round * 64 + h * 32) * 36) + lane;
            const double2 *sp2 = reinterpret_cast<const double2 *>(lds);
            for (int i = 0; i < 9; i++) {
                const int c = i * 64 + lane;
                const int bl = (c * 3641) >> 16;        // c / 18 for c < 576
                gp[i * 64] = sp2[c + bl];
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------
// MITC3+ (elements/MITC3.py, MITC3_composite.py).  One thread per (element, node pair i<=j):
// each thread rebuilds the (cheap) triangle geometry, evaluates the shape data of functions
// i, j and the cubic bubble at the 3 Hammer points, forms its 6x6 block including the bubble
// condensation term, rotates it and writes 288 contiguous bytes (FS_LAYOUT_SYMBLK, 6 pairs).
// ------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------
// MITC3+ stiffness, wave-per-10-elements form (the thread-per-pair statement of the same arithmetic it replaced is
// parked in tools/exp/parked_kernels_r02.inc.txt).
//   stage 1  lane = (element q, function f = node 0..2 or bubble 3), 40 lanes: triangle geometry and the shape /
//            tying data of f at the 3 Hammer points -> LDS, 23 doubles per (q, f); frame and area per element
//   stage 2  lane = (element q, node pair p), 60 lanes: the pair's 6 x 6 block from the LDS image, Gauss point by
//            Gauss point (only one point's data is live, so the wave fits two per SIMD), rotation, then the 60
//            blocks of the wave -- 17 280 contiguous bytes of FS_LAYOUT_SYMBLK -- leave through an LDS staging area
//            as whole-line stores (30 blocks per pass), like the tiled MITC4 kernel.
// ------------------------------------------------------------------------------------------
constexpr int M3_EPW = 10;                       // elements per wave
constexpr int M3_FN = 0;                         // fn[c < 23][q][f]
constexpr int M3_GEO = M3_FN + 23 * 40;          // geo[c < 10][q]: area, R (9)
constexpr int M3_STG = M3_GEO + 10 * M3_EPW;     // staging, 30 slots
constexpr int M3_LDS = M3_STG + 30 * STG_SLOT;   // 2160 doubles = 17 280 B

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void mitc3_k_lds_kernel(const double *__restrict__ xyz,
                                                         const int32_t *__restrict__ prop_id,
                                                         const fs_shell_prop *__restrict__ props, int32_t n_props,
                                                         int64_t E, double *__restrict__ out,
                                                         int32_t *__restrict__ status) {
    __shared__ double lds[M3_LDS];
    const int lane = threadIdx.x;
    const int64_t e0 = (int64_t)blockIdx.x * M3_EPW;
    // ---------------- stage 1 ----------------
    if (lane < 4 * M3_EPW) {
        const int q = lane >> 2, f = lane & 3;
        const int64_t e_raw = e0 + q;
        const int64_t e = e_raw < E ? e_raw : E - 1;
        double x[3][3];
        for (int c = 0; c < 9; c++) (&x[0][0])[c] = xyz[e * 9 + c];
        TriGeom geo;
        tri_geometry(x, geo);
        TriFn F;
        tri_fn(geo, f, F);
        const int qf = lane;                                   // q * 4 + f
        for (int g = 0; g < 3; g++) {
            lds[M3_FN + (0 + g) * 40 + qf] = F.fx[g];
            lds[M3_FN + (3 + g) * 40 + qf] = F.fy[g];
            lds[M3_FN + (6 + 2 * g) * 40 + qf] = F.gx[g][0];
            lds[M3_FN + (7 + 2 * g) * 40 + qf] = F.gx[g][1];
            lds[M3_FN + (12 + 2 * g) * 40 + qf] = F.gy[g][0];
            lds[M3_FN + (13 + 2 * g) * 40 + qf] = F.gy[g][1];
            lds[M3_FN + (18 + g) * 40 + qf] = F.h[g];
        }
        lds[M3_FN + 21 * 40 + qf] = F.dhx;
        lds[M3_FN + 22 * 40 + qf] = F.dhy;
        if (f == 0) {
            lds[M3_GEO + 0 * M3_EPW + q] = geo.area;
            for (int k = 0; k < 3; k++) {
                lds[M3_GEO + (1 + k) * M3_EPW + q] = geo.e1[k];
                lds[M3_GEO + (4 + k) * M3_EPW + q] = geo.e2[k];
                lds[M3_GEO + (7 + k) * M3_EPW + q] = geo.e3[k];
            }
            if (status && e_raw < E) status[e] = geo.status;
        }
    }
    __syncthreads();
    // ---------------- stage 2 ----------------
    const int q = (lane * 43) >> 8;                            // lane / 6 for lane < 64
    const int p = lane - q * 6;
    const bool worker = lane < 6 * M3_EPW;
    const int ti = (p < 3) ? 0 : ((p < 5) ? 1 : 2);
    const int tj = (p < 3) ? p : ((p < 5) ? p - 2 : 2);
    const int qq = worker ? q : 0;
    const int64_t e_raw = e0 + qq;
    const int64_t e = e_raw < E ? e_raw : E - 1;
    double G[6][6];
    tri_pair_block<40, M3_EPW, 4, 1>(lds + M3_FN, lds + M3_GEO, qq, ti, tj, props + safe_pid(prop_id[e], n_props), G);
    // ---------------- output: 60 blocks = 1080 contiguous 16-byte chunks, 30 blocks per pass ----------------
    const int64_t nblk_total = E * 6;
    const int64_t blk0 = e0 * 6;
    for (int h = 0; h < 2; h++) {
        if (worker && lane / 30 == h) {
            double2 *sp = reinterpret_cast<double2 *>(lds + M3_STG + (lane - 30 * h) * STG_SLOT);
            for (int t = 0; t < 18; t++) sp[t] = make_double2((&G[0][0])[2 * t], (&G[0][0])[2 * t + 1]);
        }
        __syncthreads();
        const int64_t pass_blk0 = blk0 + 30 * h;
        const int64_t left = nblk_total - pass_blk0;             // blocks that exist from here on
        const int nvalid = left >= 30 ? 540 : (left > 0 ? (int)left * 18 : 0);
        double2 *gp = reinterpret_cast<double2 *>(out + pass_blk0 * 36);
        const double2 *sp2 = reinterpret_cast<const double2 *>(lds + M3_STG);
        for (int i = 0; i < 9; i++) {
            const int c = i * 64 + lane;
            if (c < nvalid) {
                const int bl = (c * 3641) >> 16;                 // c / 18 for c < 576
                gp[c] = sp2[c + bl];                             // staging slots are 19 chunks apart
            }
        }
        __syncthreads();
    }
}

// MITC3 consistent mass (MITC3.py:1105-1138): one thread per (element, pair)
__global__ __launch_bounds__(256) void mitc3_m_kernel(const double *__restrict__ xyz,
                                                      const int32_t *__restrict__ prop_id,
                                                      const fs_shell_prop *__restrict__ props, int32_t n_props,
                                                      int64_t E, double *__restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= E * 6) return;
    const int64_t e = t / 6;
    const int p = (int)(t - e * 6);
    const int ti = (p < 3) ? 0 : ((p < 5) ? 1 : 2);
    const int tj = (p < 3) ? p : ((p < 5) ? p - 2 : 2);
    double x[3][3];
    for (int c = 0; c < 9; c++) (&x[0][0])[c] = xyz[e * 9 + c];
    TriGeom geo;
    tri_geometry(x, geo);
    const fs_shell_prop *pr = props + safe_pid(prop_id[e], n_props);
    const double mt = pr->m_t, mr = pr->m_r;
    double ct = 0.0, cr = 0.0;
    for (int g = 0; g < 3; g++) {
        const double r = kTriR[g], s = kTriS[g];
        const double hi = (ti == 0) ? (1.0 - r - s) : ((ti == 1) ? r : s);
        const double hj = (tj == 0) ? (1.0 - r - s) : ((tj == 1) ? r : s);
        ct = ct + hi * hj * mt * (1.0 / 3.0) * geo.area;
        cr = cr + hi * hj * mr * (1.0 / 3.0) * geo.area;
    }
    const double R[3][3] = {{geo.e1[0], geo.e1[1], geo.e1[2]}, {geo.e2[0], geo.e2[1], geo.e2[2]}, {geo.e3[0], geo.e3[1], geo.e3[2]}};
    double G[6][6];
    for (int a = 0; a < 6; a++)
        for (int b = 0; b < 6; b++) G[a][b] = 0.0;
    for (int p3 = 0; p3 < 3; p3++)
        for (int q = 0; q < 3; q++) {
            double st = 0.0, sr = 0.0;
            for (int k = 0; k < 3; k++) {
                st = FMA(R[k][p3] * ct, R[k][q], st);
                sr = FMA(R[k][p3] * cr, R[k][q], sr);
            }
            G[p3][q] = st;
            G[3 + p3][3 + q] = sr;
        }
    double2 *dst = reinterpret_cast<double2 *>(out + t * 36);
    for (int c = 0; c < 18; c++) dst[c] = make_double2((&G[0][0])[2 * c], (&G[0][0])[2 * c + 1]);
}

// ------------------------------------------------------------------------------------------
// MITC3 total-Lagrangian tangent and internal force (MITC3.py:1144-1832, composite overrides
// MITC3_composite.py:246-265, 496-621).  One lane per element; runs AFTER mitc3_k_lds_kernel has
// written the linear blocks K (global axes, INITIAL frame R0) into `kt` (FS_LAYOUT_SYMBLK, 6 pairs).
// Reference quirk kept: T() is rebuilt from the CURRENT coordinates x + u (MITC3.py:471-494) while
// K0, the derivatives and the area come from the initial geometry.  With Rc the current frame and
// P = R0^T Rc (3x3):  K0 in current-frame global axes = P^T K P block-wise, so
//   K_T   = P^T K P + Rc^T sym(K_L + K_sigma) Rc          (translational sub-blocks only)
//   f_int = P^T (K (P u)) + Rc^T ( f_membrane(GL) - K_m,CST u_loc )        [nonlinear rows]
//         = P^T (K (P u))                                                  [linear rows]
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void mitc3_nl_kernel(const double *__restrict__ xyz,
                                                      const int32_t *__restrict__ prop_id,
                                                      const fs_shell_prop *__restrict__ props, int32_t n_props,
                                                      const double *__restrict__ ue, int64_t E,
                                                      const double *k0, double *kt, int want_kt,
                                                      double *__restrict__ fe,
                                                      const double *__restrict__ sig_in) {
    // sig_in: optional [E][3] local membrane stress replacing the computed one (MITC3.py:1717-1745)
    // k0: the linear blocks K (read); kt: K_T (written when want_kt); they may be the same buffer (in place)
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    double x[3][3], u[3][6];
    for (int c = 0; c < 9; c++) (&x[0][0])[c] = xyz[e * 9 + c];
    for (int c = 0; c < 18; c++) (&u[0][0])[c] = ue[e * 18 + c];
    const fs_shell_prop *pr = props + safe_pid(prop_id[e], n_props);
    TriGeom geo;
    tri_geometry(x, geo);
    const double R0[3][3] = {{geo.e1[0], geo.e1[1], geo.e1[2]}, {geo.e2[0], geo.e2[1], geo.e2[2]}, {geo.e3[0], geo.e3[1], geo.e3[2]}};
    double Rc[3][3];
    {   // MITC3.py:480-488 on x + u
        double v12[3], v13[3], n[3];
        for (int k = 0; k < 3; k++) {
            const double a0 = x[0][k] + u[0][k], a1 = x[1][k] + u[1][k], a2 = x[2][k] + u[2][k];
            v12[k] = a1 - a0; v13[k] = a2 - a0;
        }
        double l = norm3(v12);
        for (int k = 0; k < 3; k++) Rc[0][k] = v12[k] / l;
        cross3(v12, v13, n);
        l = norm3(n);
        for (int k = 0; k < 3; k++) Rc[2][k] = n[k] / l;
        cross3(Rc[2], Rc[0], Rc[1]);
    }
    const double *bin = k0 + e * 216;
    double *blk = kt + e * 216;
    const int pidx[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
    {
#pragma clang fp contract(fast)
    double P[3][3];
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) P[a][b] = R0[0][a] * Rc[0][b] + R0[1][a] * Rc[1][b] + R0[2][a] * Rc[2][b];
    const double dr[3] = {-1.0, 1.0, 0.0}, ds[3] = {-1.0, 0.0, 1.0};
    double nx[3], ny[3];
    for (int i = 0; i < 3; i++) { nx[i] = geo.J00 * dr[i]; ny[i] = geo.J10 * dr[i] + geo.J11 * ds[i]; }
    const double h = pr->h, ar = geo.area;
    double C[3][3];
    for (int a = 0; a < 9; a++) (&C[0][0])[a] = pr->C_raw[a];
    // local displacements in the CURRENT frame
    double ul[3][3], rl[3][3];
    for (int i = 0; i < 3; i++)
        for (int a = 0; a < 3; a++) {
            ul[i][a] = Rc[a][0] * u[i][0] + Rc[a][1] * u[i][1] + Rc[a][2] * u[i][2];
            rl[i][a] = Rc[a][0] * u[i][3] + Rc[a][1] * u[i][4] + Rc[a][2] * u[i][5];
        }
    double f[3][6];
    for (int c = 0; c < 18; c++) (&f[0][0])[c] = 0.0;
    if (fe) {   // P^T (K (P u))
        double q[3][6], w[3][6];
        for (int i = 0; i < 3; i++)
            for (int a = 0; a < 3; a++) {
                q[i][a] = P[a][0] * u[i][0] + P[a][1] * u[i][1] + P[a][2] * u[i][2];
                q[i][3 + a] = P[a][0] * u[i][3] + P[a][1] * u[i][4] + P[a][2] * u[i][5];
            }
        for (int c = 0; c < 18; c++) (&w[0][0])[c] = 0.0;
        for (int i = 0; i < 3; i++)
            for (int j = i; j < 3; j++) {
                const double *B = bin + pidx[i][j] * 36;
                for (int a = 0; a < 6; a++)
                    for (int b = 0; b < 6; b++) {
                        const double v = B[a * 6 + b];
                        w[i][a] += v * q[j][b];
                        if (i != j) w[j][b] += v * q[i][a];
                    }
            }
        for (int i = 0; i < 3; i++)
            for (int a = 0; a < 3; a++) {
                f[i][a] = P[0][a] * w[i][0] + P[1][a] * w[i][1] + P[2][a] * w[i][2];
                f[i][3 + a] = P[0][a] * w[i][3] + P[1][a] * w[i][4] + P[2][a] * w[i][5];
            }
    }
    if (want_kt) {  // K <- P^T K P, every 3x3 sub-block
        for (int p = 0; p < 6; p++)
            for (int ba = 0; ba < 2; ba++)
                for (int bb = 0; bb < 2; bb++) {
                    double X[3][3], Y[3][3];
                    for (int a = 0; a < 3; a++)
                        for (int b = 0; b < 3; b++) X[a][b] = bin[p * 36 + (3 * ba + a) * 6 + 3 * bb + b];
                    for (int a = 0; a < 3; a++)
                        for (int b = 0; b < 3; b++) Y[a][b] = P[0][a] * X[0][b] + P[1][a] * X[1][b] + P[2][a] * X[2][b];
                    for (int a = 0; a < 3; a++)
                        for (int b = 0; b < 3; b++)
                            blk[p * 36 + (3 * ba + a) * 6 + 3 * bb + b] = Y[a][0] * P[0][b] + Y[a][1] * P[1][b] + Y[a][2] * P[2][b];
                }
    }
    if (pr->nonlinear) {
        // membrane stress at the centroid (MITC3.py:1582-1631; laminate MITC3_composite.py:496-505)
        double Sm00, Sm01, Sm11;
        {
            double eps[3] = {0, 0, 0}, kap[3] = {0, 0, 0};
            for (int i = 0; i < 3; i++) {
                eps[0] += nx[i] * ul[i][0]; eps[1] += ny[i] * ul[i][1]; eps[2] += ny[i] * ul[i][0] + nx[i] * ul[i][1];
                kap[0] += nx[i] * rl[i][1]; kap[1] += -ny[i] * rl[i][0]; kap[2] += ny[i] * rl[i][1] - nx[i] * rl[i][0];
            }
            double sig[3];
            if (pr->flags & FS_PROP_LAMINATE) {
                for (int a = 0; a < 3; a++)
                    sig[a] = ((pr->A[a * 3] * eps[0] + pr->A[a * 3 + 1] * eps[1] + pr->A[a * 3 + 2] * eps[2]) +
                              (pr->B[a * 3] * kap[0] + pr->B[a * 3 + 1] * kap[1] + pr->B[a * 3 + 2] * kap[2])) / h;
            } else {
                for (int a = 0; a < 3; a++) sig[a] = C[a][0] * eps[0] + C[a][1] * eps[1] + C[a][2] * eps[2];
            }
            if (sig_in) { sig[0] = sig_in[e * 3]; sig[1] = sig_in[e * 3 + 1]; sig[2] = sig_in[e * 3 + 2]; }
            Sm00 = sig[0] * h; Sm11 = sig[1] * h; Sm01 = sig[2] * h;
            // K_m,CST u_loc (MITC3.py:1018-1024), subtracted from f: A eps * area
            if (fe) {
                double N[3];
                for (int a = 0; a < 3; a++) N[a] = ar * (pr->A[a * 3] * eps[0] + pr->A[a * 3 + 1] * eps[1] + pr->A[a * 3 + 2] * eps[2]);
                for (int i = 0; i < 3; i++) {
                    const double fl[3] = {-(nx[i] * N[0] + ny[i] * N[2]), -(ny[i] * N[1] + nx[i] * N[2]), 0.0};
                    for (int a = 0; a < 3; a++) f[i][a] += Rc[0][a] * fl[0] + Rc[1][a] * fl[1] + Rc[2][a] * fl[2];
                }
            }
        }
        // displacement gradient: GLOBAL components, local derivatives (MITC3.py:1198-1225), constant
        double H[3][2];
        for (int c = 0; c < 3; c++) {
            H[c][0] = u[0][c] * nx[0] + u[1][c] * nx[1] + u[2][c] * nx[2];
            H[c][1] = u[0][c] * ny[0] + u[1][c] * ny[1] + u[2][c] * ny[2];
        }
        const double E00 = 0.5 * (2.0 * H[0][0] + (H[0][0] * H[0][0] + H[1][0] * H[1][0] + H[2][0] * H[2][0]));
        const double E11 = 0.5 * (2.0 * H[1][1] + (H[0][1] * H[0][1] + H[1][1] * H[1][1] + H[2][1] * H[2][1]));
        const double E01 = 0.5 * ((H[0][1] + H[1][0]) + (H[0][0] * H[0][1] + H[1][0] * H[1][1] + H[2][0] * H[2][1]));
        const double eg[3] = {E00, E11, 2.0 * E01};
        double sg[3];
        for (int a = 0; a < 3; a++) sg[a] = C[a][0] * eg[0] + C[a][1] * eg[1] + C[a][2] * eg[2];
        const double wh = ar * h;
        double Bt[3][3][3], BL[3][3][3], CBt[3][3][3], CBL[3][3][3];
        for (int i = 0; i < 3; i++) {
            for (int k = 0; k < 3; k++) {
                BL[i][0][k] = (k == 0) ? nx[i] : 0.0;
                BL[i][1][k] = (k == 1) ? ny[i] : 0.0;
                BL[i][2][k] = (k == 0) ? ny[i] : ((k == 1) ? nx[i] : 0.0);
                Bt[i][0][k] = BL[i][0][k] + H[k][0] * nx[i];
                Bt[i][1][k] = BL[i][1][k] + H[k][1] * ny[i];
                Bt[i][2][k] = BL[i][2][k] + (H[k][0] * ny[i] + H[k][1] * nx[i]);
            }
            for (int a = 0; a < 3; a++)
                for (int k = 0; k < 3; k++) {
                    CBt[i][a][k] = wh * (C[a][0] * Bt[i][0][k] + C[a][1] * Bt[i][1][k] + C[a][2] * Bt[i][2][k]);
                    CBL[i][a][k] = wh * (C[a][0] * BL[i][0][k] + C[a][1] * BL[i][1][k] + C[a][2] * BL[i][2][k]);
                }
            if (fe) {
                double fl[3];
                for (int k = 0; k < 3; k++) fl[k] = wh * (Bt[i][0][k] * sg[0] + Bt[i][1][k] * sg[1] + Bt[i][2][k] * sg[2]);
                for (int a = 0; a < 3; a++) f[i][a] += Rc[0][a] * fl[0] + Rc[1][a] * fl[1] + Rc[2][a] * fl[2];
            }
        }
        if (want_kt) {
            for (int i = 0; i < 3; i++)
                for (int j = i; j < 3; j++) {
                    const int p = pidx[i][j];
                    const double ks = ar * (nx[i] * (Sm00 * nx[j] + Sm01 * ny[j]) + ny[i] * (Sm01 * nx[j] + Sm11 * ny[j]));
                    double Xs[3][3];
                    for (int k = 0; k < 3; k++)
                        for (int l = 0; l < 3; l++) {
                            double v = Bt[i][0][k] * CBt[j][0][l] + Bt[i][1][k] * CBt[j][1][l] + Bt[i][2][k] * CBt[j][2][l];
                            v -= BL[i][0][k] * CBL[j][0][l] + BL[i][1][k] * CBL[j][1][l] + BL[i][2][k] * CBL[j][2][l];
                            if (k == l) v += ks;
                            Xs[k][l] = v;
                        }
                    if (i == j)
                        for (int k = 0; k < 3; k++)
                            for (int l = k + 1; l < 3; l++) { const double m = 0.5 * (Xs[k][l] + Xs[l][k]); Xs[k][l] = m; Xs[l][k] = m; }
                    double Yt[3][3];
                    for (int a = 0; a < 3; a++)
                        for (int l = 0; l < 3; l++) Yt[a][l] = Rc[0][a] * Xs[0][l] + Rc[1][a] * Xs[1][l] + Rc[2][a] * Xs[2][l];
                    for (int a = 0; a < 3; a++)
                        for (int b = 0; b < 3; b++)
                            blk[p * 36 + a * 6 + b] += Yt[a][0] * Rc[0][b] + Yt[a][1] * Rc[1][b] + Yt[a][2] * Rc[2][b];
                }
        }
    }
    if (want_kt)   // keep the diagonal blocks exactly symmetric (MITC3.py:1771 symmetrises K_T)
        for (int i = 0; i < 3; i++) {
            const int p = pidx[i][i];
            for (int a = 0; a < 6; a++)
                for (int b = a + 1; b < 6; b++) {
                    const double m = 0.5 * (blk[p * 36 + a * 6 + b] + blk[p * 36 + b * 6 + a]);
                    blk[p * 36 + a * 6 + b] = m; blk[p * 36 + b * 6 + a] = m;
                }
        }
    if (fe)
        for (int c = 0; c < 18; c++) fe[e * 18 + c] = (&f[0][0])[c];
    }
}

// ------------------------------------------------------------------------------------------
// Shell geometric stiffness from a given LOCAL membrane stress [sxx, syy, sxy]
// (MITC4.py:1191-1225, MITC3.py:1501-1580): K_sigma couples only the translations,
// block (i,j) = g_ij R^T R with g_ij = h sum_gp w detJ grad(N_i)^T S grad(N_j).
// One thread per (element, node pair i<=j); elements with max|sigma| < 1e-20 give zero blocks
// (the assembler skips them, core/assembler.py:498-500).
// ------------------------------------------------------------------------------------------
template <int NN>
__global__ __launch_bounds__(256) void shell_kg_kernel(const double *__restrict__ xyz,
                                                       const int32_t *__restrict__ prop_id,
                                                       const fs_shell_prop *__restrict__ props, int32_t n_props,
                                                       const double *__restrict__ sigma, int64_t E,
                                                       double *__restrict__ out, int tiled) {
    constexpr int NP = NN * (NN + 1) / 2;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= E * NP) return;
    const int64_t e = t / NP;
    const int p = (int)(t - e * NP);
    int ti = 0, rem = p;
    while (rem >= NN - ti) { rem -= NN - ti; ti++; }
    const int tj = ti + rem;
    const double s0 = sigma[e * 3], s1 = sigma[e * 3 + 1], s2 = sigma[e * 3 + 2];
    const double h = props[safe_pid(prop_id[e], n_props)].h;
    const bool skip = __builtin_fmax(__builtin_fabs(s0), __builtin_fmax(__builtin_fabs(s1), __builtin_fabs(s2))) < 1e-20;
    double g = 0.0, R[3][3];
    if constexpr (NN == 3) {
        double x[3][3];
        for (int c = 0; c < 9; c++) (&x[0][0])[c] = xyz[e * 9 + c];
        TriGeom geo;
        tri_geometry(x, geo);
        for (int k = 0; k < 3; k++) { R[0][k] = geo.e1[k]; R[1][k] = geo.e2[k]; R[2][k] = geo.e3[k]; }
        const double dri = (ti == 0) ? -1.0 : ((ti == 1) ? 1.0 : 0.0), dsi = (ti == 0) ? -1.0 : ((ti == 2) ? 1.0 : 0.0);
        const double drj = (tj == 0) ? -1.0 : ((tj == 1) ? 1.0 : 0.0), dsj = (tj == 0) ? -1.0 : ((tj == 2) ? 1.0 : 0.0);
        const double xi_ = geo.J00 * dri, yi_ = geo.J10 * dri + geo.J11 * dsi;
        const double xj_ = geo.J00 * drj, yj_ = geo.J10 * drj + geo.J11 * dsj;
        const double v = xi_ * ((s0 * h) * xj_ + (s2 * h) * yj_) + yi_ * ((s2 * h) * xj_ + (s1 * h) * yj_);
        for (int q = 0; q < 3; q++) g += (1.0 / 3.0) * geo.area * v;
    } else {
        double x[4][3];
        for (int c = 0; c < 12; c++) (&x[0][0])[c] = xyz[e * 12 + c];
        QuadGeom geo;
        quad_geometry(x, geo);
        for (int k = 0; k < 3; k++) { R[0][k] = geo.e1[k]; R[1][k] = geo.e2[k]; R[2][k] = geo.e3[k]; }
        for (int q = 0; q < 4; q++) {
            const double xi = (q == 1 || q == 2) ? kGP : -kGP, eta = (q >= 2) ? kGP : -kGP;
            double dxi[4], deta[4], ia, ib, ic, id;
            dshape(xi, eta, dxi, deta);
            const double detJ = jacobian(geo, dxi, deta, ia, ib, ic, id);
            double xi_ = 0, yi_ = 0, xj_ = 0, yj_ = 0;
            for (int n = 0; n < 4; n++) {   // select nodes ti, tj without dynamic register indexing
                const double nxn = ia * dxi[n] + ib * deta[n], nyn = ic * dxi[n] + id * deta[n];
                if (n == ti) { xi_ = nxn; yi_ = nyn; }
                if (n == tj) { xj_ = nxn; yj_ = nyn; }
            }
            g += detJ * (xi_ * ((s0 * h) * xj_ + (s2 * h) * yj_) + yi_ * ((s2 * h) * xj_ + (s1 * h) * yj_));
        }
    }
    if (skip) g = 0.0;
    double G[6][6];
    for (int a = 0; a < 6; a++)
        for (int b = 0; b < 6; b++) G[a][b] = 0.0;
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) G[a][b] = skip ? 0.0 : (R[0][a] * g) * R[0][b] + (R[1][a] * g) * R[1][b] + (R[2][a] * g) * R[2][b];
    if (ti == tj)   // diagonal block exactly symmetric (K_sigma is symmetrised, MITC4.py:1219)
        for (int a = 0; a < 3; a++)
            for (int b = a + 1; b < 3; b++) { const double m = 0.5 * (G[a][b] + G[b][a]); G[a][b] = m; G[b][a] = m; }
    double2 *dst = reinterpret_cast<double2 *>(out + ((NN == 4) ? quad_blk(tiled, e, p) : t) * 36);
    for (int c = 0; c < 18; c++) dst[c] = make_double2((&G[0][0])[2 * c], (&G[0][0])[2 * c + 1]);
}

// ------------------------------------------------------------------------------------------
// Centrifugal prestress (MITC3.py:1633-1715 -> local [sxx,syy,sxy]; SOLID.py:268-320 -> global Voigt
// [xx,yy,zz,xy,yz,zx]).  One thread per element; `ax` is the unit rotation axis, `ce` a point on it.
// ------------------------------------------------------------------------------------------
struct Vec3 { double v[3]; };

template <int NN>
__device__ __forceinline__ bool radial_dir(const double *x, const Vec3 &ax, const Vec3 &ce, double *rd, double &rr) {
    double rv[3], rq[3];
    for (int k = 0; k < 3; k++) {
        double s = x[k];
        for (int i = 1; i < NN; i++) s = s + x[3 * i + k];
        rv[k] = s / (double)NN - ce.v[k];
    }
    const double d = dot3(rv, ax.v);
    for (int k = 0; k < 3; k++) { const double par = d * ax.v[k]; rq[k] = rv[k] - par; }
    rr = norm3(rq);
    if (rr < 1e-10) return false;
    for (int k = 0; k < 3; k++) rd[k] = rq[k] / rr;
    return true;
}

__global__ __launch_bounds__(256) void centrifugal_tri_kernel(const double *__restrict__ xyz,
                                                              const int32_t *__restrict__ prop_id,
                                                              const fs_shell_prop *__restrict__ props, int32_t n_props,
                                                              double omega, Vec3 ax, Vec3 ce, int64_t E,
                                                              double *__restrict__ sigma) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    double x[3][3];
    for (int c = 0; c < 9; c++) (&x[0][0])[c] = xyz[e * 9 + c];
    TriGeom geo;
    tri_geometry(x, geo);
    double rd[3], rr, s0 = 0.0, s1 = 0.0, s2 = 0.0;
    if (radial_dir<3>(&x[0][0], ax, ce, rd, rr)) {
        const double rho = props[safe_pid(prop_id[e], n_props)].rho;
        const double sm = rho * (omega * omega) * rr * __builtin_sqrt(geo.area);
        const double c = dot3(geo.e1, rd), sn = dot3(geo.e2, rd);
        s0 = sm * (c * c); s1 = sm * (sn * sn); s2 = sm * c * sn;
    }
    sigma[e * 3] = s0; sigma[e * 3 + 1] = s1; sigma[e * 3 + 2] = s2;
}

template <int NN>
__global__ __launch_bounds__(256) void centrifugal_solid_kernel(const double *__restrict__ xyz,
                                                                const int32_t *__restrict__ prop_id,
                                                                const double *__restrict__ props, int32_t n_props,
                                                                double omega, Vec3 ax, Vec3 ce, int64_t E,
                                                                double *__restrict__ sigma) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    double x[NN * 3];
    for (int c = 0; c < NN * 3; c++) x[c] = xyz[e * NN * 3 + c];
    double rd[3], rr, sg[6] = {0, 0, 0, 0, 0, 0};
    if (radial_dir<NN>(x, ax, ce, rd, rr)) {
        double d[3];
        for (int k = 0; k < 3; k++) {
            double mn = x[k], mx = x[k];
            for (int i = 1; i < NN; i++) { mn = __builtin_fmin(mn, x[3 * i + k]); mx = __builtin_fmax(mx, x[3 * i + k]); }
            d[k] = mx - mn;
        }
        const double rho = props[(int64_t)safe_pid(prop_id[e], n_props) * 37 + 36];
        const double sm = rho * (omega * omega) * rr * (norm3(d) / 1.732);
        sg[0] = sm * rd[0] * rd[0]; sg[1] = sm * rd[1] * rd[1]; sg[2] = sm * rd[2] * rd[2];
        sg[3] = sm * rd[0] * rd[1]; sg[4] = sm * rd[1] * rd[2]; sg[5] = sm * rd[0] * rd[2];
    }
    for (int c = 0; c < 6; c++) sigma[e * 6 + c] = sg[c];
}

// ------------------------------------------------------------------------------------------
// Linear solids (elements/SOLID.py): TETRA4 / PYRAMID5 / WEDGE6 / HEXA8, 3 dofs per node.
// Workgroup = EPB elements.  Phase 1: thread (element, Gauss point) -> Jacobian, w*detJ and the
// physical gradients of all NN shape functions into LDS.  Phase 2: thread (element, node pair
// i<=j) -> 3x3 block of K (B_i^T C B_j), M, K_G; threads p < NN also the body-load vector.
// Output FS_LAYOUT_SYMBLK with d = 3: [E][NN(NN+1)/2][9].
// ------------------------------------------------------------------------------------------
template <int NN, int NG, bool TAB>
__global__ __launch_bounds__(256) void solid_kernel(const double *__restrict__ xyz,
                                                    const int32_t *__restrict__ prop_id,
                                                    const double *__restrict__ props, int32_t n_props, int64_t E,
                                                    SolidRule rule, const double *__restrict__ tab, int ng_tab,
                                                    double bx, double by, double bz,
                                                    const double *__restrict__ sigma,
                                                    double *__restrict__ Ke, double *__restrict__ Me,
                                                    double *__restrict__ fe, double *__restrict__ KGe,
                                                    int32_t *__restrict__ status) {
    constexpr int NPAIR = NN * (NN + 1) / 2;
    constexpr int EPB = 256 / NPAIR;
    constexpr int GS = NN * 3;                     // doubles of one gradient set
    // gradients of phase 1; reused at the very end as the staging area of the stiffness blocks (EPB * NPAIR * 9 doubles)
    constexpr int NGRAD = EPB * NG * GS, NOUT = EPB * NPAIR * 9;
    __shared__ double s_gg[(2 * NGRAD > NOUT) ? 2 * NGRAD : NOUT];
    double *s_g = s_gg;                            // dN/dx      (SOLID.py:245-248)
    double *s_gt = s_gg + NGRAD;                   // inv_J.T form used by K_G (SOLID.py:352-354)
    __shared__ double s_wd[EPB * NG];              // w * detJ
    __shared__ double s_N[EPB * NG * NN];          // shape values (mass, body load)
    const int tid = threadIdx.x;
    const int64_t e0 = (int64_t)blockIdx.x * EPB;
    const int ng = TAB ? ng_tab : rule.ng;
    {
#pragma clang fp contract(fast)
        const int el = tid / ng, q = tid - el * ng;
        const int64_t e = e0 + el;
        if (el < EPB && e < E) {
            double N[NN], dxi[NN], deta[NN], dzeta[NN], wq;
            if constexpr (TAB) {   // quadratic solids: host-tabulated shape data (femshell_geom.h)
                const double *tq = tab + (int64_t)q * NN * 4;
#pragma unroll
                for (int i = 0; i < NN; i++) { N[i] = tq[4 * i]; dxi[i] = tq[4 * i + 1]; deta[i] = tq[4 * i + 2]; dzeta[i] = tq[4 * i + 3]; }
                wq = tab[(int64_t)ng * NN * 4 + q];
            } else {
                solid_shape<NN>(rule.pt[q][0], rule.pt[q][1], rule.pt[q][2], N, dxi, deta, dzeta);
                wq = rule.w[q];
            }
            double J[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
            const double *x = xyz + e * GS;
#pragma unroll
            for (int i = 0; i < NN; i++)
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const double xc = x[3 * i + c];
                    J[0][c] += dxi[i] * xc; J[1][c] += deta[i] * xc; J[2][c] += dzeta[i] * xc;
                }
            const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1], c01 = J[1][0] * J[2][2] - J[1][2] * J[2][0],
                         c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
            const double det = J[0][0] * c00 - J[0][1] * c01 + J[0][2] * c02;
            if (det <= 1e-14 && status) atomicOr(status + e, FS_ST_BAD_JAC);   // ValueError in SOLID.py:211-212
            double inv[3][3];
            {   // nine IEEE quotients over one denominator: one shared reciprocal (divr, femshell_geom.h), the same bits
                const double rdet = recip(det);
                inv[0][0] = divr(c00, det, rdet); inv[0][1] = divr(J[0][2] * J[2][1] - J[0][1] * J[2][2], det, rdet);
                inv[0][2] = divr(J[0][1] * J[1][2] - J[0][2] * J[1][1], det, rdet); inv[1][0] = divr(-c01, det, rdet);
                inv[1][1] = divr(J[0][0] * J[2][2] - J[0][2] * J[2][0], det, rdet);
                inv[1][2] = divr(J[0][2] * J[1][0] - J[0][0] * J[1][2], det, rdet);
                inv[2][0] = divr(c02, det, rdet); inv[2][1] = divr(J[0][1] * J[2][0] - J[0][0] * J[2][1], det, rdet);
                inv[2][2] = divr(J[0][0] * J[1][1] - J[0][1] * J[1][0], det, rdet);
            }
            double *g = s_g + (el * NG + q) * GS, *gt = s_gt + (el * NG + q) * GS;
#pragma unroll
            for (int i = 0; i < NN; i++) {
#pragma unroll
                for (int a = 0; a < 3; a++) {
                    g[3 * i + a] = inv[a][0] * dxi[i] + inv[a][1] * deta[i] + inv[a][2] * dzeta[i];
                    if (KGe) gt[3 * i + a] = inv[0][a] * dxi[i] + inv[1][a] * deta[i] + inv[2][a] * dzeta[i];
                }
                s_N[(el * NG + q) * NN + i] = N[i];
            }
            s_wd[el * NG + q] = det * wq;
        }
    }
    __syncthreads();
    {
#pragma clang fp contract(fast)
        const int el = tid / NPAIR, p = tid - el * NPAIR;
        const bool valid = el < EPB && e0 + el < E;
        const int64_t e = valid ? e0 + el : e0;
        double K[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        if (valid) {
        int ti = 0, rem = p;
        while (rem >= NN - ti) { rem -= NN - ti; ti++; }
        const int tj = ti + rem;
        const double *pr = props + 37 * (int64_t)safe_pid(prop_id[e], n_props);
        const int64_t ob = (e * NPAIR + p) * 9;
        if (Ke) {
            double C[6][6];
#pragma unroll
            for (int a = 0; a < 6; a++)
#pragma unroll
                for (int b = 0; b < 6; b++) C[a][b] = 0.5 * (pr[a * 6 + b] + pr[b * 6 + a]);   // 0.5 (K + K^T) of SOLID.py:565
            // C is constant over the element: contract over the Gauss points FIRST, G = sum_q w detJ grad N_i grad N_j^T
            // (3 x 3), then K = C : G -- ng x 12 + 81 multiply-adds instead of ng x 90 (the same sum of products
            // re-associated: eps-level differences; fs_fused_assemble_hex8 uses the same form)
            double G[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
            for (int q = 0; q < ng; q++) {
                const double *gi = s_g + (el * NG + q) * GS + 3 * ti, *gj = s_g + (el * NG + q) * GS + 3 * tj;
                const double wd = s_wd[el * NG + q];
                const double ax = gi[0] * wd, ay = gi[1] * wd, az = gi[2] * wd, bx_ = gj[0], by_ = gj[1], bz_ = gj[2];
                G[0][0] += ax * bx_; G[0][1] += ax * by_; G[0][2] += ax * bz_;
                G[1][0] += ay * bx_; G[1][1] += ay * by_; G[1][2] += ay * bz_;
                G[2][0] += az * bx_; G[2][1] += az * by_; G[2][2] += az * bz_;
            }
            {
                // Voigt row of strain component (al, ga): [xx, yy, zz, xy, yz, zx]
                constexpr int V[3][3] = {{0, 3, 5}, {3, 1, 4}, {5, 4, 2}};
#pragma unroll
                for (int al = 0; al < 3; al++)
#pragma unroll
                    for (int be = 0; be < 3; be++) {
                        double v = 0.0;
#pragma unroll
                        for (int ga = 0; ga < 3; ga++)
#pragma unroll
                            for (int de = 0; de < 3; de++) v += C[V[al][ga]][V[be][de]] * G[ga][de];
                        K[al][be] = v;
                    }
            }
            if (ti == tj) {
#pragma unroll
                for (int a = 0; a < 3; a++)
#pragma unroll
                    for (int b = a + 1; b < 3; b++) { const double m = 0.5 * (K[a][b] + K[b][a]); K[a][b] = m; K[b][a] = m; }
            }
        }
        if (Me || fe) {
            const double rho = pr[36];
            double m = 0.0, f0 = 0.0;
            for (int q = 0; q < ng; q++) {
                const double *N = s_N + (el * NG + q) * NN;
                const double wd = s_wd[el * NG + q];
                m += rho * (N[ti] * N[tj]) * wd;
                f0 += N[p < NN ? p : 0] * wd;
            }
            if (Me) {
#pragma unroll
                for (int c = 0; c < 9; c++) Me[ob + c] = (c == 0 || c == 4 || c == 8) ? m : 0.0;
            }
            if (fe && p < NN) {   // body_load: f_i = sum_q N_i b detJ w   (SOLID.py:599-631)
                fe[e * GS + 3 * p + 0] = f0 * bx; fe[e * GS + 3 * p + 1] = f0 * by; fe[e * GS + 3 * p + 2] = f0 * bz;
            }
        }
        if (KGe) {   // SOLID.py:322-374
            const double *sg = sigma + e * 6;
            const double S[3][3] = {{sg[0], sg[3], sg[5]}, {sg[3], sg[1], sg[4]}, {sg[5], sg[4], sg[2]}};
            double gsum = 0.0;
            for (int q = 0; q < ng; q++) {
                const double *gi = s_gt + (el * NG + q) * GS + 3 * ti, *gj = s_gt + (el * NG + q) * GS + 3 * tj;
                double v = 0.0;
#pragma unroll
                for (int a = 0; a < 3; a++) v += (gi[0] * S[0][a] + gi[1] * S[1][a] + gi[2] * S[2][a]) * gj[a];
                gsum += v * s_wd[el * NG + q];
            }
#pragma unroll
            for (int c = 0; c < 9; c++) KGe[ob + c] = (c == 0 || c == 4 || c == 8) ? gsum : 0.0;
        }
        }
        // stiffness blocks: a thread storing its own 72 bytes issues 9 instructions of 8-byte pieces 72 bytes apart (every
        // instruction touches every line of the wave's range); staged through LDS the workgroup's blocks -- contiguous in
        // FS_LAYOUT_SYMBLK -- leave as whole 512-byte runs per wave instruction
        if (Ke) {
            __syncthreads();        // phase-1 data no longer needed: the staging area overlays it
            if (valid)
#pragma unroll
                for (int c = 0; c < 9; c++) s_gg[tid * 9 + c] = (&K[0][0])[c];
            __syncthreads();
            const int64_t left = E - e0;
            const int nout = (int)(left < EPB ? left : EPB) * NPAIR * 9;
            double *dst = Ke + e0 * NPAIR * 9;
            for (int i = tid; i < nout; i += 256) dst[i] = s_gg[i];
        }
    }
}

// local-axes views of global-axes element results (the transform_to_global=False outputs of MITC4.py:1191-1341 and
// MITC3.py:1501-1832: K_local = T K T^T, f_local = T f with T = blockdiag of the frame rows e1, e2, e3): one thread per
// 3 x 3 sub-block of the dense [nd][nd] matrix, resp. per 3-vector.  The frame is the kernels' own geometry stage on
// `xyz` (quads: the initial coordinates; triangles: whatever coordinates the caller's T() uses -- the reference takes
// the CURRENT ones, MITC3.py:471-492).
template <int NN>
__global__ __launch_bounds__(256) void rotate_local_kernel(const double *__restrict__ xyz, int64_t E,
                                                           const double *__restrict__ Kg, const double *__restrict__ fg,
                                                           double *__restrict__ Kl, double *__restrict__ fl) {
    constexpr int NB = 2 * NN, ND = 6 * NN;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= E * NB * NB) return;
    const int64_t e = t / (NB * NB);
    const int bi = (int)(t % (NB * NB)) / NB, bj = (int)(t % (NB * NB)) % NB;
    double x[NN][3], R[3][3];
    for (int c = 0; c < NN * 3; c++) (&x[0][0])[c] = xyz[e * NN * 3 + c];
    if constexpr (NN == 4) {
        QuadGeom g;
        quad_geometry(x, g);
        for (int k = 0; k < 3; k++) { R[0][k] = g.e1[k]; R[1][k] = g.e2[k]; R[2][k] = g.e3[k]; }
    } else {
        TriGeom g;
        tri_geometry(x, g);
        for (int k = 0; k < 3; k++) { R[0][k] = g.e1[k]; R[1][k] = g.e2[k]; R[2][k] = g.e3[k]; }
    }
    if (Kg) {
        double B[3][3], Y[3][3];
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) B[a][b] = Kg[(e * ND + 3 * bi + a) * ND + 3 * bj + b];
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) Y[a][b] = R[a][0] * B[0][b] + R[a][1] * B[1][b] + R[a][2] * B[2][b];
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++)
                Kl[(e * ND + 3 * bi + a) * ND + 3 * bj + b] = Y[a][0] * R[b][0] + Y[a][1] * R[b][1] + Y[a][2] * R[b][2];
    }
    if (fg && bj == 0)
        for (int a = 0; a < 3; a++)
            fl[e * ND + 3 * bi + a] = R[a][0] * fg[e * ND + 3 * bi] + R[a][1] * fg[e * ND + 3 * bi + 1] + R[a][2] * fg[e * ND + 3 * bi + 2];
}

// geometry probe for bit-exactness tests
__global__ void mitc4_geom_kernel(const double *xyz, int64_t E, double *frame, double *xl, double *detJ,
                                  double *acoef) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    double x[4][3];
    for (int c = 0; c < 12; c++) (&x[0][0])[c] = xyz[e * 12 + c];
    QuadGeom geo;
    quad_geometry(x, geo);
    for (int k = 0; k < 3; k++) {
        frame[e * 9 + k] = geo.e1[k]; frame[e * 9 + 3 + k] = geo.e2[k]; frame[e * 9 + 6 + k] = geo.e3[k];
    }
    for (int i = 0; i < 4; i++) { xl[e * 8 + 2 * i] = geo.xl[i][0]; xl[e * 8 + 2 * i + 1] = geo.xl[i][1]; }
    for (int g = 0; g < 4; g++) {
        const double xi = (g == 1 || g == 2) ? kGP : -kGP, eta = (g >= 2) ? kGP : -kGP;
        double dxi[4], deta[4], ia, ib, ic, id;
        dshape(xi, eta, dxi, deta);
        detJ[e * 4 + g] = jacobian(geo, dxi, deta, ia, ib, ic, id);
    }
    for (int k = 0; k < 5; k++) acoef[e * 5 + k] = geo.a[k];
}

// status words of the stiffness kernels without the element matrices: frame / coefficient bits of quad_geometry and
// FS_ST_BAD_JAC over the four Gauss points (what quad_stage1 reports), FS_ST_BAD_PROP for ids outside the table
__global__ __launch_bounds__(256) void mitc4_geom_status_kernel(const double *__restrict__ xyz,
                                                               const int32_t *__restrict__ prop_id, int32_t n_props,
                                                               int64_t E, int32_t *__restrict__ status) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    double x[4][3];
    for (int c = 0; c < 12; c++) (&x[0][0])[c] = xyz[e * 12 + c];
    QuadGeom geo;
    quad_geometry(x, geo);
    int st = geo.status;
    for (int g = 0; g < 4; g++) {
        const double xi = (g == 1 || g == 2) ? kGP : -kGP, eta = (g >= 2) ? kGP : -kGP;
        double dxi[4], deta[4], ia, ib, ic, id;
        dshape(xi, eta, dxi, deta);
        st |= (jacobian(geo, dxi, deta, ia, ib, ic, id) > 0.0) ? 0 : FS_ST_BAD_JAC;
    }
    if (prop_id && (uint32_t)prop_id[e] >= (uint32_t)n_props) st |= FS_ST_BAD_PROP;
    status[e] = st;
}

// Invariants fused_quad_k_kernel relies on when it indexes LDS and `data` with values read from the plan
// (fs_fused_plan_check): one workgroup per patch (grid-stride), lanes over the patch's records.  err[0] = number of
// violations, err[1] = first violated rule (1-based, see fs_fused_plan_check), err[2..3] = patch / record of it.
__global__ __launch_bounds__(256) void fused_plan_check_kernel(int64_t n_pat, const int64_t *__restrict__ pat_eptr,
                                                              const int32_t *__restrict__ pat_elems,
                                                              const int64_t *__restrict__ pat_rptr,
                                                              const fs_fused_rec *__restrict__ recs, int64_t E,
                                                              int32_t max_patch_elems, int64_t nnz,
                                                              unsigned long long *__restrict__ err) {
    auto report = [&](int rule, int64_t p, int64_t r) {
        if (atomicAdd(&err[0], 1ull) == 0) { err[1] = (unsigned long long)rule; err[2] = (unsigned long long)p; err[3] = (unsigned long long)r; }
    };
    for (int64_t p = blockIdx.x; p < n_pat; p += gridDim.x) {
        const int64_t e0 = pat_eptr[p], e1 = pat_eptr[p + 1], r0 = pat_rptr[p], r1 = pat_rptr[p + 1];
        const int64_t ne = e1 - e0;
        if (threadIdx.x == 0) {
            if (ne < 1 || ne > max_patch_elems || e0 < 0) report(1, p, -1);
            if (r1 <= r0 || (r0 & 63) || (r1 & 63) || r0 < 0) report(2, p, -1);
            if (p == 0 && (e0 != 0 || r0 != 0)) report(2, p, -1);
        }
        if (ne < 1 || ne > max_patch_elems || r1 <= r0) continue;
        for (int64_t q = threadIdx.x; q < ne; q += blockDim.x) {
            const int64_t el = pat_elems[e0 + q];
            if (el < 0 || el >= E) report(3, p, q);
        }
        for (int64_t r = r0 + threadIdx.x; r < r1; r += blockDim.x) {
            const fs_fused_rec rc = recs[r];
            for (int k = 0; k < 2; k++) {
                const uint32_t ev = rc.ev[k];
                if (FS_EV_ACTIVE(ev) && (FS_EV_SLOT(ev) >= ne || FS_EV_TI(ev) > FS_EV_TJ(ev))) report(4, p, r);
            }
            const uint32_t fl = rc.flags;
            const int maxc = (int)((fl >> 8) & 255u), pos = (int)((fl >> 16) & 15u);
            if (maxc < 1 || maxc > 16 || pos >= maxc || ((fl & FS_FL_CHAIN) && (pos == 0 || (int)(r & 15) < pos))) report(5, p, r);
            // chain length is wave-uniform (the kernel reads it with readfirstlane)
            const uint32_t fl0 = recs[r & ~(int64_t)63].flags;
            if (((fl0 >> 8) & 255u) != (uint32_t)maxc) report(5, p, r);
            for (int k = 0; k < 2; k++) {
                const bool out = (fl & (k == 0 ? FS_FL_OUT0 : FS_FL_OUT1)) != 0;
                if (!out) continue;
                const uint32_t sd = (rc.slots >> (16 * k)) & 255u, st = (rc.slots >> (16 * k + 8)) & 255u;
                if (sd >= 128u) report(6, p, r);
                const int64_t d0 = rc.dst[k], str = rc.stride[k];
                if (d0 < 0 || (d0 & 1) || str < 6 || (str & 1) || d0 + 5 * str + 6 > nnz) report(7, p, r);
                const int64_t dt = rc.dst_t[k], strt = rc.stride_t[k];
                if (dt >= 0) {
                    if (st >= 128u) report(6, p, r);
                    if ((dt & 1) || strt < 6 || (strt & 1) || dt + 5 * strt + 6 > nnz) report(7, p, r);
                }
            }
        }
    }
}

// FS_LAYOUT_SYMBLK -> dense [E][nn*d][nn*d]
__global__ void blk_expand_kernel(const double *__restrict__ packed, int64_t E, int nn, int d,
                                  double *__restrict__ dense, int tiled) {
    const int nd = nn * d;
    const int64_t total = E * nd * nd;
    const int npair = nn * (nn + 1) / 2;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = t / (nd * nd);
        const int rc = (int)(t - e * nd * nd);
        const int r = rc / nd, c = rc - r * nd;
        int i = r / d, k = r - i * d, j = c / d, l = c - j * d;
        if (i > j) { int tmp = i; i = j; j = tmp; tmp = k; k = l; l = tmp; }
        const int p = i * nn - (i * (i - 1)) / 2 + (j - i);
        const int64_t blk = tiled ? blk_t16(e, p) : e * npair + p;
        dense[t] = packed[blk * (d * d) + k * d + l];
    }
}

// generic segmented gather-sum (CSR values or vectors), one thread per destination slot
__global__ void seg_gather_sum_kernel(const double *__restrict__ vals, const int64_t *__restrict__ src,
                                      const int64_t *__restrict__ seg_ptr, int64_t n,
                                      double *__restrict__ out) {
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n;
         s += (int64_t)gridDim.x * blockDim.x) {
        const int64_t lo = seg_ptr[s], hi = seg_ptr[s + 1];
        double v = 0.0;
        for (int64_t q = lo; q < hi; q++) v = v + vals[src[q]];  // element order == reference order
        out[s] = v;
    }
}

// d = 3 (solids), thread per ENTRY: workgroup = 28 node-pair blocks x 9 lanes.  The 28 descriptors reach the lanes
// through LDS (84 lanes x 16 B instead of 48 B per lane); the 9 lanes of a block read the 72 contiguous bytes of a
// source block with one instruction per contribution -- a transposed contribution is the same 72 bytes in another
// lane order, so no strided pieces and no second pass -- and workgroups walk the blocks XCD by XCD like the d = 6
// kernel.  Same sums in the same element order as the thread-per-row kernel it replaced
// (tools/exp/parked_kernels_r02.inc.txt; 2.46 -> 2.23 ms on 1 048 576 HEXA8, bitwise equal).
__global__ __launch_bounds__(256) void csr_numeric_desc3_lds_kernel(const double *__restrict__ vals,
                                                                    const fs_blk_desc *__restrict__ desc,
                                                                    const int64_t *__restrict__ contrib,
                                                                    int64_t nblk, double *__restrict__ data) {
    constexpr int TPB = 9, BPW = 28;
    __shared__ double2 s_desc[BPW * 3];
    const int tid = threadIdx.x;
    const int64_t per_xcd = (gridDim.x + 7) / 8;
    const int64_t wg = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    const int64_t b0 = wg * BPW;
    if (b0 >= nblk) return;
    if (tid < BPW * 3 && b0 * 3 + tid < nblk * 3)
        s_desc[tid] = reinterpret_cast<const double2 *>(desc)[b0 * 3 + tid];
    __syncthreads();
    const int bl = tid / TPB, r = tid - bl * TPB;
    if (bl >= BPW || b0 + bl >= nblk) return;
    const int k = r / 3, l = r - k * 3;
    const int rt = l * 3 + k;                   // the entry this lane takes from a transposed contribution
    const fs_blk_desc dsc = reinterpret_cast<const fs_blk_desc *>(s_desc)[bl];
    const int n = dsc.cnt & (FS_BLK_PACKED - 1);
    double v;
    if (n <= 4) {
        double x[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int64_t cc = dsc.c[c];
            x[c] = (c < n) ? vals[(cc >> 1) * 9 + ((cc & 1) ? rt : r)] : 0.0;
        }
        v = ((x[0] + x[1]) + x[2]) + x[3];      // element order == reference insertion order
    } else if (dsc.cnt & FS_BLK_PACKED) {
        // 5..8 contributions with their codes inline (uint32[8]): the diagonal block of a node shared by 8 bricks --
        // all value loads in flight together, no dependent read of the contribution list
        double y[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t cc = (uint32_t)((uint64_t)dsc.c[u >> 1] >> (32 * (u & 1)));
            y[u] = (u < n) ? vals[(int64_t)(cc >> 1) * 9 + ((cc & 1u) ? rt : r)] : 0.0;
        }
        v = 0.0;
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (u < n) v = v + y[u];
    } else {
        // more than 8 contributions (tetrahedral meshes): batches of 8 with all index loads, then all value loads, in
        // flight together; the sum still runs in element order
        v = 0.0;
        const int64_t lo = dsc.c[0], hi = lo + n;
        for (int64_t q0 = lo; q0 < hi; q0 += 8) {
            int64_t cc[8];
#pragma unroll
            for (int u = 0; u < 8; u++) cc[u] = (q0 + u < hi) ? contrib[q0 + u] : 0;
            double y[8];
#pragma unroll
            for (int u = 0; u < 8; u++) y[u] = (q0 + u < hi) ? vals[(cc[u] >> 1) * 9 + ((cc[u] & 1) ? rt : r)] : 0.0;
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (q0 + u < hi) v = v + y[u];
        }
    }
    data[dsc.dst + (int64_t)k * dsc.stride + l] = v;
}

// LDS-staged descriptor assembly (d = 6): workgroup = 14 node-pair blocks x 18 lanes.
//   * the 14 descriptors are fetched once per workgroup (42 lanes x 16 B) instead of by every lane;
//   * every contribution is fetched as ONE coalesced 16-byte load per lane (the 288-byte source
//     block is contiguous), transposed contributions are transposed through LDS -- no strided
//     8-byte loads, so the vector-memory instruction count per wave drops from ~9 to ~4.
// Sums in element order (the reference's ADD_VALUES insertion order), +0.0 pads below 4 contributions.
__global__ __launch_bounds__(256) void csr_numeric_desc6_lds_kernel(const double *__restrict__ vals,
                                                                    const fs_blk_desc *__restrict__ desc,
                                                                    const int64_t *__restrict__ contrib,
                                                                    int64_t nblk, double *__restrict__ data) {
    constexpr int D = 6, DD = 36, TPB = 18, BPW = 14;
    __shared__ double2 s_desc[BPW * 3];
    __shared__ double2 s_val[4][BPW * TPB];
    const int tid = threadIdx.x;
    // XCD-aware order: workgroup w runs on XCD w % 8, so each XCD gets one contiguous eighth of the blocks: a block
    // (I,J) and its transposed partner (J,I) read the same source block a few thousand blocks apart -- inside one
    // XCD that distance fits its 4 MB L2, across XCDs it never would
    const int64_t per_xcd = (gridDim.x + 7) / 8;
    const int64_t wg = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    const int64_t b0 = wg * BPW;
    if (b0 >= nblk) return;
    if (tid < BPW * 3 && b0 * 3 + tid < nblk * 3)
        s_desc[tid] = reinterpret_cast<const double2 *>(desc)[b0 * 3 + tid];
    __syncthreads();
    const int bl = tid / TPB, r = tid - bl * TPB;
    const bool active = bl < BPW && b0 + bl < nblk;
    const int k = r / 3, l = 2 * (r - k * 3);
    fs_blk_desc dsc;
    dsc.cnt = 0;
    if (active) dsc = reinterpret_cast<const fs_blk_desc *>(s_desc)[bl];
    const int n = dsc.cnt & (FS_BLK_PACKED - 1);
    const bool packed = (dsc.cnt & FS_BLK_PACKED) != 0;       // 5..8 contributions, codes inline as uint32[8]
    const bool staged = active && (n <= 4 || packed);         // contributions go through the LDS transposition
    // code of inline contribution u (u < 4: also the plain int64 form of n <= 4)
    auto code = [&](int u) -> int64_t {
        return packed ? (int64_t)(uint32_t)((uint64_t)dsc.c[u >> 1] >> (32 * (u & 1))) : dsc.c[u & 3];
    };
    const double *sv0 = reinterpret_cast<const double *>(&s_val[0][bl * TPB]);
    constexpr int CS = BPW * TPB * 2;   // doubles between contribution slots
    double2 x[4], y[4];
    const int n0 = n < 4 ? n : 4;
    if (staged) {
#pragma unroll
        for (int c = 0; c < 4; c++)
            x[c] = (c < n0) ? reinterpret_cast<const double2 *>(vals + (code(c) >> 1) * DD)[r] : make_double2(0.0, 0.0);
        if (packed) {       // contributions 4..7 are fetched now too: all value loads of a block in flight together
#pragma unroll
            for (int c = 0; c < 4; c++)
                y[c] = (4 + c < n) ? reinterpret_cast<const double2 *>(vals + (code(4 + c) >> 1) * DD)[r] : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int c = 0; c < 4; c++) s_val[c][tid] = x[c];
    }
    // a second round of four for the packed blocks of this workgroup (triangle meshes: the diagonal block of a node
    // shared by 6 elements); workgroups without one -- every workgroup of a quad mesh -- skip it
    const int more = __syncthreads_or(active && packed);
    double2 v = make_double2(0.0, 0.0);
    if (staged) {
#pragma unroll
        for (int c = 0; c < 4; c++) {
            if (c < n0 && (code(c) & 1)) {
                const double *sv = sv0 + c * CS;
                x[c] = make_double2(sv[l * D + k], sv[(l + 1) * D + k]);
            }
        }
        if (!packed) {
            v.x = ((x[0].x + x[1].x) + x[2].x) + x[3].x;  // element order == reference insertion order
            v.y = ((x[0].y + x[1].y) + x[2].y) + x[3].y;
        } else {
#pragma unroll
            for (int c = 0; c < 4; c++) { v.x = v.x + x[c].x; v.y = v.y + x[c].y; }     // 0.0 + x0 + x1 + ...
        }
    }
    if (more) {
        __syncthreads();        // every lane has read round one from s_val
        if (active && packed) {
#pragma unroll
            for (int c = 0; c < 4; c++) { x[c] = y[c]; s_val[c][tid] = x[c]; }
        }
        __syncthreads();
        if (active && packed) {
#pragma unroll
            for (int c = 0; c < 4; c++) {
                if (4 + c < n && (code(4 + c) & 1)) {
                    const double *sv = sv0 + c * CS;
                    x[c] = make_double2(sv[l * D + k], sv[(l + 1) * D + k]);
                }
            }
#pragma unroll
            for (int c = 0; c < 4; c++)
                if (4 + c < n) { v.x = v.x + x[c].x; v.y = v.y + x[c].y; }
        }
    }
    if (!active) return;
    if (!staged) {
        // more than 8 contributions: batches of 8 with all index loads, then all value loads, in flight together; the
        // sum still runs in element order
        const int64_t lo = dsc.c[0], hi = lo + n;
        for (int64_t q0 = lo; q0 < hi; q0 += 8) {
            int64_t cc[8];
#pragma unroll
            for (int u = 0; u < 8; u++) cc[u] = (q0 + u < hi) ? contrib[q0 + u] : 0;
            double2 y[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const double *base = vals + (cc[u] >> 1) * DD;
                y[u] = (q0 + u >= hi) ? make_double2(0.0, 0.0)
                       : ((cc[u] & 1) ? make_double2(base[l * D + k], base[(l + 1) * D + k])
                                      : *reinterpret_cast<const double2 *>(base + k * D + l));
            }
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (q0 + u < hi) { v.x = v.x + y[u].x; v.y = v.y + y[u].y; }
        }
    }
    *reinterpret_cast<double2 *>(data + dsc.dst + (int64_t)k * dsc.stride + l) = v;
}


// prop_id outside [0, n_props): FS_ST_BAD_PROP into the status array (the element kernels used row 0 for it)
__global__ void prop_check_kernel(const int32_t *__restrict__ prop_id, int64_t E, int32_t n_props,
                                  int32_t *__restrict__ status, int assign) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const bool bad = (uint32_t)prop_id[e] >= (uint32_t)n_props;
    if (assign) status[e] = bad ? FS_ST_BAD_PROP : 0;
    else if (bad) status[e] |= FS_ST_BAD_PROP;
}

template <int D>
__global__ void csr_pattern_blk_kernel(const int64_t *__restrict__ nbr_ptr, const int32_t *__restrict__ blk_row,
                                       const int32_t *__restrict__ nbr, int64_t n_nodes, int64_t nblk,
                                       int64_t *__restrict__ indptr, int32_t *__restrict__ indices) {
    constexpr int DD = D * D;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_nodes * D) {
        const int64_t I = t / D;
        const int k = (int)(t - I * D);
        const int64_t r0 = nbr_ptr[I], m = nbr_ptr[I + 1] - r0;
        indptr[t] = DD * r0 + (int64_t)k * D * m;
        if (t == n_nodes * D - 1) indptr[t + 1] = DD * nbr_ptr[n_nodes];
    }
    if (t < nblk * DD) {
        const int64_t b = t / DD;
        const int kl = (int)(t - b * DD);
        const int k = kl / D, l = kl - k * D;
        const int64_t I = blk_row[b];
        const int64_t r0 = nbr_ptr[I], m = nbr_ptr[I + 1] - r0;
        indices[DD * r0 + (int64_t)k * D * m + D * (b - r0) + l] = nbr[b] * D + l;
    }
}

__global__ void gather_dofs_kernel(const double *__restrict__ u, const int64_t *__restrict__ dofs,
                                   int64_t n, double *__restrict__ ue) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n;
         t += (int64_t)gridDim.x * blockDim.x)
        ue[t] = u[dofs[t]];
}

// interface-row exchange helpers: pack ghost-row values, add received values
__global__ void halo_pack_kernel(const double *__restrict__ data, const int64_t *__restrict__ idx,
                                 int64_t n, double *__restrict__ buf) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n;
         t += (int64_t)gridDim.x * blockDim.x)
        buf[t] = data[idx[t]];
}
__global__ void halo_add_kernel(const double *__restrict__ buf, const int64_t *__restrict__ idx,
                                int64_t n, double *__restrict__ data) {
    // idx holds each destination at most once per message, so plain read-modify-write is race-free
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n;
         t += (int64_t)gridDim.x * blockDim.x)
        data[idx[t]] = data[idx[t]] + buf[t];
}

inline unsigned grid_for(int64_t n, int block, int64_t cap = 1 << 20) {
    int64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

}  // namespace

int fs_fail_(int code, const char *what, int hip_error) { return fail(code, what, (hipError_t)hip_error); }
int fs_noalloc_prepare_();

extern "C" {

const char *fs_last_error(void) { return g_err; }

int fs_device_info(int device, char *name, int len, int *cu_count) {
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return fail(FS_EHIP, "hipGetDeviceProperties", e);
    if (name && len > 0) snprintf(name, len, "%s", prop.gcnArchName);
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return fail(FS_ENOSYS, "device is not gfx950");
    // set-up for fs_shell_quad_fint_noalloc: when `device` is the current device, the address of its static workspace
    // is resolved here (hipGetSymbolAddress may load the code object), so that the entry point itself does nothing
    // of the kind later -- e.g. under HIP graph capture
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur == device) return fs_noalloc_prepare_();
    return FS_OK;
}

// FS_ST_BAD_PROP for ids outside the table, after the element kernels (which used row 0 for them); `assign`: no
// kernel of this call wrote the status array
static int check_props(const int32_t *prop_id, int64_t E, int32_t n_props, int32_t *status, int assign, hipStream_t s) {
    if (!status || E <= 0) return FS_OK;
    hipLaunchKernelGGL(prop_check_kernel, dim3(grid_for(E, 256, INT32_MAX)), dim3(256), 0, s, prop_id, E, n_props, status,
                       assign);
    FS_LAUNCH_CHECK("prop_check_kernel");
    return FS_OK;
}

int fs_shell_quad_batch(const double *xyz, const int32_t *prop_id, const fs_shell_prop *props,
                        int32_t n_props, const double *u_e, int64_t E, uint32_t want,
                        uint32_t layout, double *Ke, double *Me, double *KTe, double *fe,
                        int32_t *status, void *stream) {
    if (E < 0 || n_props <= 0) return fail(FS_EINVAL, "fs_shell_quad_batch: bad sizes");
    if (E == 0) return FS_OK;
    if (!xyz || !prop_id || !props) return fail(FS_EINVAL, "fs_shell_quad_batch: null input");
    if (layout != FS_LAYOUT_SYMBLK && layout != FS_LAYOUT_SYMBLK_T16)
        return fail(FS_EINVAL, "fs_shell_quad_batch: kernels emit FS_LAYOUT_SYMBLK / _T16; use fs_blk_expand for dense");
    const int tiled = layout == FS_LAYOUT_SYMBLK_T16;
    hipStream_t s = (hipStream_t)stream;
    if (want & (FS_WANT_KT | FS_WANT_FINT)) {
        // the linear blocks K do not depend on u (the reference caches them: MITC4.K is a cached_property read by
        // compute_tangent_stiffness / compute_internal_forces): with Ke given and FS_WANT_K clear, Ke is an INPUT
        // (a previous FS_WANT_K result for the same elements, tiled layout) and the stiffness kernel is skipped
        const bool k_cached = tiled && Ke && !(want & FS_WANT_K);
        if (!u_e) return fail(FS_EINVAL, "fs_shell_quad_batch: KT/FINT need u_e");
        if (!KTe && ((want & FS_WANT_KT) || !k_cached))
            return fail(FS_EINVAL, "fs_shell_quad_batch: KT/FINT need the KTe buffer (workspace for FINT unless Ke holds the linear blocks)");
        if ((want & FS_WANT_FINT) && !fe) return fail(FS_EINVAL, "fs_shell_quad_batch: fe is null");
        const unsigned grid = (unsigned)((E + 15) / 16);
        if (!k_cached) {
            if (tiled) hipLaunchKernelGGL(mitc4_k_kernel<true>, dim3(grid), dim3(64), 0, s, xyz, prop_id, props, n_props, E, KTe, status);
            else hipLaunchKernelGGL(mitc4_k_kernel<false>, dim3(grid), dim3(64), 0, s, xyz, prop_id, props, n_props, E, KTe, status);
            FS_LAUNCH_CHECK("mitc4_k_kernel");
        }
        if (tiled)
            hipLaunchKernelGGL(mitc4_nl_tiled_kernel<true>, dim3(grid), dim3(64), 0, s, xyz, prop_id, props, n_props, u_e, E,
                               k_cached ? Ke : KTe, KTe, (want & FS_WANT_KT) ? 1 : 0, (want & FS_WANT_FINT) ? fe : nullptr,
                               (const int64_t *)nullptr, (const double *)nullptr);
        else
            hipLaunchKernelGGL(mitc4_nl_kernel, dim3((unsigned)((E + 63) / 64)), dim3(64), 0, s, xyz, prop_id, props, n_props, u_e,
                               E, KTe, (want & FS_WANT_KT) ? 1 : 0, (want & FS_WANT_FINT) ? fe : nullptr, tiled, nullptr);
        FS_LAUNCH_CHECK("mitc4_nl_kernel");
    }
    if (want & FS_WANT_K) {
        if (!Ke) return fail(FS_EINVAL, "fs_shell_quad_batch: Ke is null");
        const unsigned grid = (unsigned)((E + 15) / 16);
        if (tiled) hipLaunchKernelGGL(mitc4_k_kernel<true>, dim3(grid), dim3(64), 0, s, xyz, prop_id, props, n_props, E, Ke, status);
        else hipLaunchKernelGGL(mitc4_k_kernel<false>, dim3(grid), dim3(64), 0, s, xyz, prop_id, props, n_props, E, Ke, status);
        FS_LAUNCH_CHECK("mitc4_k_kernel");
    }
    if (want & FS_WANT_M) {
        if (!Me) return fail(FS_EINVAL, "fs_shell_quad_batch: Me is null");
        if (tiled)
            hipLaunchKernelGGL(mitc4_m_tiled_kernel, dim3((unsigned)((E + 15) / 16)), dim3(64), 0, s, xyz, prop_id, props, n_props, E, Me);
        else
            hipLaunchKernelGGL(mitc4_m_kernel, dim3(grid_for(E * 10, 256, INT32_MAX)), dim3(256), 0, s, xyz,
                               prop_id, props, n_props, E, Me, tiled);
        FS_LAUNCH_CHECK("mitc4_m_kernel");
    }
    return check_props(prop_id, E, n_props, status, !(want & (FS_WANT_K | FS_WANT_KT | FS_WANT_FINT)) || ((want & (FS_WANT_KT | FS_WANT_FINT)) && !(want & FS_WANT_K) && tiled && Ke), s);
}

#define M3_LAUNCH(dst)                                                                                              \
    hipLaunchKernelGGL(mitc3_k_lds_kernel, dim3((unsigned)((E + M3_EPW - 1) / M3_EPW)), dim3(64), 0, s, xyz, prop_id,  \
                       props, n_props, E, dst, status)

int fs_shell_tri_batch(const double *xyz, const int32_t *prop_id, const fs_shell_prop *props,
                       int32_t n_props, const double *u_e, int64_t E, uint32_t want, uint32_t layout,
                       double *Ke, double *Me, double *KTe, double *fe, int32_t *status, void *stream) {
    if (E < 0 || n_props <= 0) return fail(FS_EINVAL, "fs_shell_tri_batch: bad sizes");
    if (E == 0) return FS_OK;
    if (!xyz || !prop_id || !props) return fail(FS_EINVAL, "fs_shell_tri_batch: null input");
    if (layout != FS_LAYOUT_SYMBLK)
        return fail(FS_EINVAL, "fs_shell_tri_batch: kernels emit FS_LAYOUT_SYMBLK; use fs_blk_expand for dense");
    hipStream_t s = (hipStream_t)stream;
    if (want & (FS_WANT_KT | FS_WANT_FINT)) {
        // as in fs_shell_quad_batch: Ke given with FS_WANT_K clear = the cached linear blocks (input)
        const bool k_cached = Ke && !(want & FS_WANT_K);
        if (!u_e) return fail(FS_EINVAL, "fs_shell_tri_batch: KT/FINT need u_e");
        if (!KTe && ((want & FS_WANT_KT) || !k_cached))
            return fail(FS_EINVAL, "fs_shell_tri_batch: KT/FINT need the KTe buffer (workspace for FINT unless Ke holds the linear blocks)");
        if ((want & FS_WANT_FINT) && !fe) return fail(FS_EINVAL, "fs_shell_tri_batch: fe is null");
        if (!k_cached) {
            M3_LAUNCH(KTe);
            FS_LAUNCH_CHECK("mitc3_k_lds_kernel");
        }
        hipLaunchKernelGGL(mitc3_nl_kernel, dim3((unsigned)((E + 63) / 64)), dim3(64), 0, s, xyz, prop_id, props, n_props, u_e,
                           E, k_cached ? Ke : KTe, KTe, (want & FS_WANT_KT) ? 1 : 0, (want & FS_WANT_FINT) ? fe : nullptr, nullptr);
        FS_LAUNCH_CHECK("mitc3_nl_kernel");
    }
    if (want & FS_WANT_K) {
        if (!Ke) return fail(FS_EINVAL, "fs_shell_tri_batch: Ke is null");
        M3_LAUNCH(Ke);
        FS_LAUNCH_CHECK("mitc3_k_lds_kernel");
    }
    if (want & FS_WANT_M) {
        if (!Me) return fail(FS_EINVAL, "fs_shell_tri_batch: Me is null");
        hipLaunchKernelGGL(mitc3_m_kernel, dim3(grid_for(E * 6, 256, INT32_MAX)), dim3(256), 0, s, xyz, prop_id,
                           props, n_props, E, Me);
        FS_LAUNCH_CHECK("mitc3_m_kernel");
    }
    return check_props(prop_id, E, n_props, status, !(want & (FS_WANT_K | FS_WANT_KT | FS_WANT_FINT)) || ((want & (FS_WANT_KT | FS_WANT_FINT)) && !(want & FS_WANT_K) && Ke), s);
}

}  // extern "C" (reopened below)

namespace {
template <int NN, int NG, bool TAB>
int launch_solid(const double *xyz, const int32_t *prop_id, const double *props, int32_t n_props, int64_t E, const SolidRule &rule,
                 const double *tab, int ng_tab, const double *b, const double *sigma, double *Ke, double *Me,
                 double *fe, double *KGe, int32_t *status, hipStream_t s) {
    constexpr int EPB = 256 / (NN * (NN + 1) / 2);
    static_assert(EPB >= 1 && EPB * NG <= 256, "phase 1 needs one thread per (element, Gauss point)");
    hipLaunchKernelGGL((solid_kernel<NN, NG, TAB>), dim3(grid_for((E + EPB - 1) / EPB, 1, INT32_MAX)), dim3(256), 0, s,
                       xyz, prop_id, props, n_props, E, rule, tab, ng_tab, b[0], b[1], b[2], sigma, Ke, Me, fe, KGe, status);
    return 0;
}

// device copies of the quadratic-solid shape tables: built once per (device, element type), kept for the
// life of the process (17 KB at most each)
std::mutex g_tab_mutex;
double *g_tab_dev[16][4] = {};
int g_tab_ng[4] = {};
int solid_table_device(int nn, const double **tab, int *ng) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail(FS_EHIP, "hipGetDevice", e);
    if (dev < 0 || dev >= 16) return fail(FS_EINVAL, "fs_solid_batch: device ordinal above 15");
    const int slot = nn == 10 ? 0 : (nn == 15 ? 1 : (nn == 20 ? 2 : 3));
    std::lock_guard<std::mutex> lock(g_tab_mutex);
    if (!g_tab_dev[dev][slot]) {
        static SolidTable t;   // 18 KB: not on the stack
        if (make_solid_table(nn, t) != 0) return fail(FS_EINVAL, "fs_solid_batch: no table for this element");
        std::vector<double> h((size_t)t.ng * nn * 4 + t.ng);
        for (int q = 0; q < t.ng; q++) {
            for (int i = 0; i < nn; i++)
                for (int c = 0; c < 4; c++) h[((size_t)q * nn + i) * 4 + c] = t.v[q][i][c];
            h[(size_t)t.ng * nn * 4 + q] = t.w[q];
        }
        double *d = nullptr;
        e = hipMalloc(&d, h.size() * sizeof(double));
        if (e != hipSuccess) return fail(FS_EHIP, "hipMalloc(solid table)", e);
        e = hipMemcpy(d, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(d); return fail(FS_EHIP, "hipMemcpy(solid table)", e); }
        g_tab_dev[dev][slot] = d;
        g_tab_ng[slot] = t.ng;
    }
    *ng = g_tab_ng[slot];
    *tab = g_tab_dev[dev][slot];
    return FS_OK;
}
}  // namespace

int fs_solid_table_device_(int nn, const double **tab, int *ng) { return solid_table_device(nn, tab, ng); }

extern "C" {

int fs_solid_batch(int32_t nn, int32_t reduced, const double *xyz, const int32_t *prop_id, const double *props,
                   int32_t n_props, int64_t E, uint32_t want, uint32_t layout, const double *body_host,
                   const double *sigma, double *Ke, double *Me, double *fe, double *KGe, int32_t *status,
                   void *stream) {
    if (E < 0 || n_props <= 0) return fail(FS_EINVAL, "fs_solid_batch: bad sizes");
    if (nn != 4 && nn != 5 && nn != 6 && nn != 8 && nn != 10 && nn != 13 && nn != 15 && nn != 20)
        return fail(FS_ENOSYS, "fs_solid_batch: TETRA4 / PYRAMID5 / WEDGE6 / HEXA8 / TETRA10 / PYRAMID13 / WEDGE15 / HEXA20 (nn = 4, 5, 6, 8, 10, 13, 15, 20)");
    if (E == 0) return FS_OK;
    if (!xyz || !prop_id || !props) return fail(FS_EINVAL, "fs_solid_batch: null input");
    if (layout != FS_LAYOUT_SYMBLK)
        return fail(FS_EINVAL, "fs_solid_batch: kernels emit FS_LAYOUT_SYMBLK; use fs_blk_expand for dense");
    if ((want & FS_WANT_K) && !Ke) return fail(FS_EINVAL, "fs_solid_batch: Ke is null");
    if ((want & FS_WANT_M) && !Me) return fail(FS_EINVAL, "fs_solid_batch: Me is null");
    if ((want & FS_WANT_FBODY) && (!fe || !body_host)) return fail(FS_EINVAL, "fs_solid_batch: fe / body is null");
    if ((want & FS_WANT_KG) && (!KGe || !sigma)) return fail(FS_EINVAL, "fs_solid_batch: KGe / sigma is null");
    SolidRule rule;
    make_solid_rule(nn <= 8 ? nn : 8, reduced, rule);   // unused by the table-driven quadratic kernels
    const double zero3[3] = {0, 0, 0};
    const double *b3 = (want & FS_WANT_FBODY) ? body_host : zero3;
    double *k = (want & FS_WANT_K) ? Ke : nullptr, *m = (want & FS_WANT_M) ? Me : nullptr;
    double *f = (want & FS_WANT_FBODY) ? fe : nullptr, *kg = (want & FS_WANT_KG) ? KGe : nullptr;
    hipStream_t s = (hipStream_t)stream;
    const double *tab = nullptr;
    int ng_tab = 0;
    if (nn > 8) {
        const int rc = solid_table_device(nn, &tab, &ng_tab);
        if (rc != FS_OK) return rc;
    }
    switch (nn) {
        case 20: launch_solid<20, 27, true>(xyz, prop_id, props, n_props, E, rule, tab, ng_tab, b3, sigma, k, m, f, kg, status, s); break;
        case 15: launch_solid<15, 21, true>(xyz, prop_id, props, n_props, E, rule, tab, ng_tab, b3, sigma, k, m, f, kg, status, s); break;
        case 13: launch_solid<13, 18, true>(xyz, prop_id, props, n_props, E, rule, tab, ng_tab, b3, sigma, k, m, f, kg, status, s); break;
        case 10: launch_solid<10, 4, true>(xyz, prop_id, props, n_props, E, rule, tab, ng_tab, b3, sigma, k, m, f, kg, status, s); break;
        case 8: launch_solid<8, 8, false>(xyz, prop_id, props, n_props, E, rule, tab, ng_tab, b3, sigma, k, m, f, kg, status, s); break;
        case 6: launch_solid<6, 8, false>(xyz, prop_id, props, n_props, E, rule, tab, ng_tab, b3, sigma, k, m, f, kg, status, s); break;
        case 5: launch_solid<5, 8, false>(xyz, prop_id, props, n_props, E, rule, tab, ng_tab, b3, sigma, k, m, f, kg, status, s); break;
        default: launch_solid<4, 8, false>(xyz, prop_id, props, n_props, E, rule, tab, ng_tab, b3, sigma, k, m, f, kg, status, s); break;
    }
    FS_LAUNCH_CHECK("solid_kernel");
    return check_props(prop_id, E, n_props, status, 0, s);   // the solid kernel ORs into the caller-zeroed array
}

int fs_solid2_table(int32_t nn, double *pts, double *w, double *vals, int32_t *ng) {
    static SolidTable t;
    static std::mutex m;
    std::lock_guard<std::mutex> lock(m);
    if (make_solid_table(nn, t) != 0) return fail(FS_EINVAL, "fs_solid2_table: nn must be 10, 13, 15 or 20");
    if (ng) *ng = t.ng;
    for (int q = 0; q < t.ng; q++) {
        if (pts) for (int k = 0; k < 3; k++) pts[3 * q + k] = t.pt[q][k];
        if (w) w[q] = t.w[q];
        if (vals)
            for (int i = 0; i < nn; i++)
                for (int c = 0; c < 4; c++) vals[((size_t)q * nn + i) * 4 + c] = t.v[q][i][c];
    }
    return FS_OK;
}

int fs_shell_kg_batch(int32_t nn, const double *xyz, const int32_t *prop_id, const fs_shell_prop *props,
                      int32_t n_props, const double *sigma, int64_t E, uint32_t layout, double *KGe, void *stream) {
    if (E < 0 || n_props <= 0) return fail(FS_EINVAL, "fs_shell_kg_batch: bad sizes");
    if (nn != 3 && nn != 4) return fail(FS_EINVAL, "fs_shell_kg_batch: nn must be 3 (MITC3) or 4 (MITC4)");
    if (E == 0) return FS_OK;
    if (!xyz || !prop_id || !props || !sigma || !KGe) return fail(FS_EINVAL, "fs_shell_kg_batch: null argument");
    if (layout != FS_LAYOUT_SYMBLK && !(layout == FS_LAYOUT_SYMBLK_T16 && nn == 4))
        return fail(FS_EINVAL, "fs_shell_kg_batch: kernels emit FS_LAYOUT_SYMBLK (or _T16 for quads); use fs_blk_expand for dense");
    const int tiled = layout == FS_LAYOUT_SYMBLK_T16;
    hipStream_t s = (hipStream_t)stream;
    if (nn == 3)
        hipLaunchKernelGGL(shell_kg_kernel<3>, dim3(grid_for(E * 6, 256, INT32_MAX)), dim3(256), 0, s, xyz, prop_id,
                           props, n_props, sigma, E, KGe, 0);
    else
        hipLaunchKernelGGL(shell_kg_kernel<4>, dim3(grid_for(E * 10, 256, INT32_MAX)), dim3(256), 0, s, xyz, prop_id,
                           props, n_props, sigma, E, KGe, tiled);
    FS_LAUNCH_CHECK("shell_kg_kernel");
    return FS_OK;
}

int fs_shell_kt_sigma_batch(int32_t nn, const double *xyz, const int32_t *prop_id, const fs_shell_prop *props,
                            int32_t n_props, const double *u_e, const double *sigma, int64_t E, double *KTe,
                            int32_t *status, void *stream) {
    if (E < 0 || n_props <= 0) return fail(FS_EINVAL, "fs_shell_kt_sigma_batch: bad sizes");
    if (nn != 3 && nn != 4) return fail(FS_EINVAL, "fs_shell_kt_sigma_batch: nn must be 3 (MITC3) or 4 (MITC4)");
    if (E == 0) return FS_OK;
    if (!xyz || !prop_id || !props || !u_e || !sigma || !KTe) return fail(FS_EINVAL, "fs_shell_kt_sigma_batch: null argument");
    hipStream_t s = (hipStream_t)stream;
    if (nn == 4) {
        hipLaunchKernelGGL(mitc4_k_kernel<false>, dim3((unsigned)((E + 15) / 16)), dim3(64), 0, s, xyz, prop_id, props,
                           n_props, E, KTe, status);
        FS_LAUNCH_CHECK("mitc4_k_kernel");
        hipLaunchKernelGGL(mitc4_nl_kernel, dim3((unsigned)((E + 63) / 64)), dim3(64), 0, s, xyz, prop_id, props, n_props,
                           u_e, E, KTe, 1, (double *)nullptr, 0, sigma);
        FS_LAUNCH_CHECK("mitc4_nl_kernel");
    } else {
        M3_LAUNCH(KTe);
        FS_LAUNCH_CHECK("mitc3_k_lds_kernel");
        hipLaunchKernelGGL(mitc3_nl_kernel, dim3((unsigned)((E + 63) / 64)), dim3(64), 0, s, xyz, prop_id, props, n_props,
                           u_e, E, KTe, KTe, 1, (double *)nullptr, sigma);
        FS_LAUNCH_CHECK("mitc3_nl_kernel");
    }
    return check_props(prop_id, E, n_props, status, 0, s);
}

int fs_shell_rotate_local(int32_t nn, const double *xyz, int64_t E, const double *K_global, const double *f_global,
                          double *K_local, double *f_local, void *stream) {
    if (E < 0) return fail(FS_EINVAL, "fs_shell_rotate_local: E < 0");
    if (nn != 3 && nn != 4) return fail(FS_EINVAL, "fs_shell_rotate_local: nn must be 3 (MITC3) or 4 (MITC4)");
    if (E == 0) return FS_OK;
    if (!xyz || (!K_global && !f_global) || (K_global && !K_local) || (f_global && !f_local))
        return fail(FS_EINVAL, "fs_shell_rotate_local: null argument");
    hipStream_t s = (hipStream_t)stream;
    if (nn == 4)
        hipLaunchKernelGGL(rotate_local_kernel<4>, dim3(grid_for(E * 64, 256, INT32_MAX)), dim3(256), 0, s, xyz, E, K_global,
                           f_global, K_local, f_local);
    else
        hipLaunchKernelGGL(rotate_local_kernel<3>, dim3(grid_for(E * 36, 256, INT32_MAX)), dim3(256), 0, s, xyz, E, K_global,
                           f_global, K_local, f_local);
    FS_LAUNCH_CHECK("rotate_local_kernel");
    return FS_OK;
}

int fs_centrifugal_prestress(int32_t solid, int32_t nn, const double *xyz, const int32_t *prop_id,
                             const void *props, int32_t n_props, double omega, const double *axis_host,
                             const double *center_host, int64_t E, double *sigma, void *stream) {
    if (E < 0 || n_props <= 0) return fail(FS_EINVAL, "fs_centrifugal_prestress: bad sizes");
    if (!axis_host || !center_host) return fail(FS_EINVAL, "fs_centrifugal_prestress: axis / center is null");
    if (!solid && nn == 4)   // 'MITC4' object has no attribute 'compute_centrifugal_prestress'
        return fail(FS_ENOSYS, "fs_centrifugal_prestress: the reference MITC4 has no compute_centrifugal_prestress");
    if ((!solid && nn != 3) || (solid && nn != 4 && nn != 5 && nn != 6 && nn != 8 && nn != 10 && nn != 13 && nn != 15 && nn != 20))
        return fail(FS_EINVAL, "fs_centrifugal_prestress: unsupported element");
    const double la = sqrt(axis_host[0] * axis_host[0] + axis_host[1] * axis_host[1] + axis_host[2] * axis_host[2]);
    if (!(la > 0.0)) return fail(FS_EINVAL, "fs_centrifugal_prestress: zero rotation axis");
    if (E == 0) return FS_OK;
    if (!xyz || !prop_id || !props || !sigma) return fail(FS_EINVAL, "fs_centrifugal_prestress: null argument");
    Vec3 ax, ce;
    for (int k = 0; k < 3; k++) { ax.v[k] = axis_host[k] / la; ce.v[k] = center_host[k]; }
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(grid_for(E, 256, INT32_MAX)), block(256);
    if (!solid) {
        hipLaunchKernelGGL(centrifugal_tri_kernel, grid, block, 0, s, xyz, prop_id, (const fs_shell_prop *)props, n_props, omega,
                           ax, ce, E, sigma);
    } else {
        const double *pp = (const double *)props;
        switch (nn) {
            case 20: hipLaunchKernelGGL(centrifugal_solid_kernel<20>, grid, block, 0, s, xyz, prop_id, pp, n_props, omega, ax, ce, E, sigma); break;
            case 15: hipLaunchKernelGGL(centrifugal_solid_kernel<15>, grid, block, 0, s, xyz, prop_id, pp, n_props, omega, ax, ce, E, sigma); break;
            case 13: hipLaunchKernelGGL(centrifugal_solid_kernel<13>, grid, block, 0, s, xyz, prop_id, pp, n_props, omega, ax, ce, E, sigma); break;
            case 10: hipLaunchKernelGGL(centrifugal_solid_kernel<10>, grid, block, 0, s, xyz, prop_id, pp, n_props, omega, ax, ce, E, sigma); break;
            case 8: hipLaunchKernelGGL(centrifugal_solid_kernel<8>, grid, block, 0, s, xyz, prop_id, pp, n_props, omega, ax, ce, E, sigma); break;
            case 6: hipLaunchKernelGGL(centrifugal_solid_kernel<6>, grid, block, 0, s, xyz, prop_id, pp, n_props, omega, ax, ce, E, sigma); break;
            case 5: hipLaunchKernelGGL(centrifugal_solid_kernel<5>, grid, block, 0, s, xyz, prop_id, pp, n_props, omega, ax, ce, E, sigma); break;
            default: hipLaunchKernelGGL(centrifugal_solid_kernel<4>, grid, block, 0, s, xyz, prop_id, pp, n_props, omega, ax, ce, E, sigma); break;
        }
    }
    FS_LAUNCH_CHECK("centrifugal kernel");
    return FS_OK;
}

int fs_shell_quad_geom(const double *xyz, int64_t E, double *frame, double *xl, double *detJ,
                       double *acoef, void *stream) {
    if (E <= 0) return E == 0 ? FS_OK : fail(FS_EINVAL, "fs_shell_quad_geom: E < 0");
    if (!xyz || !frame || !xl || !detJ) return fail(FS_EINVAL, "fs_shell_quad_geom: null argument");
    hipLaunchKernelGGL(mitc4_geom_kernel, dim3(grid_for(E, 64, INT32_MAX)), dim3(64), 0, (hipStream_t)stream,
                       xyz, E, frame, xl, detJ, acoef);
    FS_LAUNCH_CHECK("mitc4_geom_kernel");
    return FS_OK;
}

int fs_blk_expand(const double *packed, uint32_t layout, int64_t E, int32_t nn, int32_t d, double *dense,
                  void *stream) {
    if (E <= 0) return E == 0 ? FS_OK : fail(FS_EINVAL, "fs_blk_expand: E < 0");
    if (nn <= 0 || d <= 0) return fail(FS_EINVAL, "fs_blk_expand: bad nn/d");
    if (!packed || !dense) return fail(FS_EINVAL, "fs_blk_expand: null argument");
    if (layout != FS_LAYOUT_SYMBLK && !(layout == FS_LAYOUT_SYMBLK_T16 && nn == 4 && d == 6))
        return fail(FS_EINVAL, "fs_blk_expand: layout must be FS_LAYOUT_SYMBLK, or FS_LAYOUT_SYMBLK_T16 for quads");
    hipLaunchKernelGGL(blk_expand_kernel, dim3(grid_for(E * nn * d * nn * d, 256, 65536)), dim3(256), 0,
                       (hipStream_t)stream, packed, E, nn, d, dense, layout == FS_LAYOUT_SYMBLK_T16 ? 1 : 0);
    FS_LAUNCH_CHECK("blk_expand_kernel");
    return FS_OK;
}

int fs_csr_numeric(const double *vals, const int64_t *src, const int64_t *seg_ptr, int64_t nnz,
                   double *data, void *stream) {
    if (nnz <= 0) return nnz == 0 ? FS_OK : fail(FS_EINVAL, "fs_csr_numeric: nnz < 0");
    if (!vals || !src || !seg_ptr || !data) return fail(FS_EINVAL, "fs_csr_numeric: null argument");
    hipLaunchKernelGGL(seg_gather_sum_kernel, dim3(grid_for(nnz, 256, 1 << 16)), dim3(256), 0,
                       (hipStream_t)stream, vals, src, seg_ptr, nnz, data);
    FS_LAUNCH_CHECK("seg_gather_sum_kernel");
    return FS_OK;
}

int fs_vec_gather_sum(const double *vals, const int64_t *src, const int64_t *seg_ptr, int64_t n,
                      double *out, void *stream) {
    return fs_csr_numeric(vals, src, seg_ptr, n, out, stream);
}

int fs_csr_numeric_desc(const double *vals, const fs_blk_desc *desc, const int64_t *contrib,
                        int64_t nblk, int32_t d, double *data, void *stream) {
    if (nblk <= 0) return nblk == 0 ? FS_OK : fail(FS_EINVAL, "fs_csr_numeric_desc: nblk < 0");
    if (!vals || !desc || !contrib || !data) return fail(FS_EINVAL, "fs_csr_numeric_desc: null argument");
    hipStream_t s = (hipStream_t)stream;
    if (d == 6) {
        hipLaunchKernelGGL(csr_numeric_desc6_lds_kernel, dim3(8 * grid_for(nblk, 14 * 8, INT32_MAX / 8)), dim3(256), 0, s,
                           vals, desc, contrib, nblk, data);
    } else if (d == 3) {
        hipLaunchKernelGGL(csr_numeric_desc3_lds_kernel, dim3(8 * grid_for(nblk, 28 * 8, INT32_MAX / 8)), dim3(256), 0, s,
                           vals, desc, contrib, nblk, data);
    } else {
        return fail(FS_EINVAL, "fs_csr_numeric_desc: d must be 3 or 6");
    }
    FS_LAUNCH_CHECK("csr_numeric_desc_kernel");
    return FS_OK;
}

int fs_csr_pattern_blk(const int64_t *nbr_ptr, const int32_t *blk_row, const int32_t *nbr,
                       int64_t n_nodes, int64_t nblk, int32_t d, int64_t *indptr, int32_t *indices,
                       void *stream) {
    if (n_nodes <= 0 || nblk < 0) return fail(FS_EINVAL, "fs_csr_pattern_blk: bad sizes");
    if (!nbr_ptr || !indptr || (nblk > 0 && (!blk_row || !nbr || !indices)))
        return fail(FS_EINVAL, "fs_csr_pattern_blk: null argument");
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = (nblk * d * d > n_nodes * d) ? nblk * d * d : n_nodes * d;
    if (d == 6) {
        hipLaunchKernelGGL(csr_pattern_blk_kernel<6>, dim3(grid_for(n, 256, INT32_MAX)), dim3(256), 0, s, nbr_ptr,
                           blk_row, nbr, n_nodes, nblk, indptr, indices);
    } else if (d == 3) {
        hipLaunchKernelGGL(csr_pattern_blk_kernel<3>, dim3(grid_for(n, 256, INT32_MAX)), dim3(256), 0, s, nbr_ptr,
                           blk_row, nbr, n_nodes, nblk, indptr, indices);
    } else {
        return fail(FS_EINVAL, "fs_csr_pattern_blk: d must be 3 or 6");
    }
    FS_LAUNCH_CHECK("csr_pattern_blk_kernel");
    return FS_OK;
}

int fs_halo_pack(const double *data, const int64_t *idx, int64_t n, double *buf, void *stream) {
    if (n <= 0) return n == 0 ? FS_OK : fail(FS_EINVAL, "fs_halo_pack: n < 0");
    if (!data || !idx || !buf) return fail(FS_EINVAL, "fs_halo_pack: null argument");
    hipLaunchKernelGGL(halo_pack_kernel, dim3(grid_for(n, 256, 1 << 16)), dim3(256), 0, (hipStream_t)stream, data,
                       idx, n, buf);
    FS_LAUNCH_CHECK("halo_pack_kernel");
    return FS_OK;
}

int fs_halo_add(const double *buf, const int64_t *idx, int64_t n, double *data, void *stream) {
    if (n <= 0) return n == 0 ? FS_OK : fail(FS_EINVAL, "fs_halo_add: n < 0");
    if (!data || !idx || !buf) return fail(FS_EINVAL, "fs_halo_add: null argument");
    hipLaunchKernelGGL(halo_add_kernel, dim3(grid_for(n, 256, 1 << 16)), dim3(256), 0, (hipStream_t)stream, buf,
                       idx, n, data);
    FS_LAUNCH_CHECK("halo_add_kernel");
    return FS_OK;
}

// ------------------------------------------------------------------------------------------
// fs_shell_quad_fint_noalloc: internal forces of <= FS_FINT_NOALLOC_MAX_E quads for callers that cannot allocate
// (HIP graph capture, serving loops).  Same kernels as fs_shell_quad_batch(FS_WANT_FINT) with a KTe workspace; the
// workspace (the linear blocks K in FS_LAYOUT_SYMBLK_T16) is the file-scope device array below instead of a caller
// buffer.  `scratch` is a HOST-side stand-in for the symbol: its device address is taken once per device with
// hipGetSymbolAddress and only that pointer is ever handed to a kernel.
// ------------------------------------------------------------------------------------------
}  // extern "C"

namespace {
constexpr int64_t kFintNoallocTiles = FS_FINT_NOALLOC_MAX_E / 16;
static __device__ __align__(256) double scratch[kFintNoallocTiles * 160 * 36];   // 47 185 920 B per GPU
constexpr int kMaxDevices = 64;
double *g_scratch_ptr[kMaxDevices] = {};
std::mutex g_scratch_mu;

// CU count of the CURRENT device, cached per device index (a process may drive several GPUs from several threads)
int g_cu_count[kMaxDevices] = {};
std::mutex g_cu_mu;
int device_cu_count(int *out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail(FS_EHIP, "hipGetDevice", e);
    if (dev < 0 || dev >= kMaxDevices) return fail(FS_EINVAL, "device index out of range");
    std::lock_guard<std::mutex> lock(g_cu_mu);
    if (!g_cu_count[dev]) {
        int n = 0;
        e = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess || n <= 0) return fail(FS_EHIP, "hipDeviceGetAttribute(MultiprocessorCount)", e);
        g_cu_count[dev] = n;
    }
    *out = g_cu_count[dev];
    return FS_OK;
}

// device address of `scratch` on the CURRENT device (each GPU holds its own copy at its own address)
int scratch_ptr_current(double **out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail(FS_EHIP, "hipGetDevice", e);
    if (dev < 0 || dev >= kMaxDevices) return fail(FS_EINVAL, "fs_shell_quad_fint_noalloc: device index out of range");
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void *p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) return fail(FS_EHIP, "hipGetSymbolAddress(scratch)", e);
        g_scratch_ptr[dev] = static_cast<double *>(p);
    }
    *out = g_scratch_ptr[dev];
    return FS_OK;
}
}  // namespace

int fs_noalloc_prepare_() {
    double *ws = nullptr;
    return scratch_ptr_current(&ws);
}

extern "C" {

int fs_shell_quad_fint_noalloc(const double *xyz, const int32_t *prop_id, const fs_shell_prop *props,
                               int32_t n_props, const double *u_e, int64_t E, double *fe, int32_t *status,
                               void *stream) {
    if (E < 0 || n_props <= 0) return fail(FS_EINVAL, "fs_shell_quad_fint_noalloc: bad sizes");
    if (E > FS_FINT_NOALLOC_MAX_E)
        return fail(FS_EINVAL, "fs_shell_quad_fint_noalloc: E exceeds FS_FINT_NOALLOC_MAX_E (16384); use "
                               "fs_shell_quad_batch with a caller-provided KTe workspace");
    if (E == 0) return FS_OK;
    if (!xyz || !prop_id || !props || !u_e || !fe) return fail(FS_EINVAL, "fs_shell_quad_fint_noalloc: null argument");
    double *ws = nullptr;
    const int rc = scratch_ptr_current(&ws);
    if (rc != FS_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    const unsigned grid = (unsigned)((E + 15) / 16);
    // every block of every tile (the padding of a ragged last tile included) is rewritten by the stiffness kernel
    // before the force kernel reads it, so nothing of the previous call's contents has to be cleared
    hipLaunchKernelGGL(mitc4_k_kernel<true>, dim3(grid), dim3(64), 0, s, xyz, prop_id, props, n_props, E, ws, status);
    FS_LAUNCH_CHECK("mitc4_k_kernel");
    hipLaunchKernelGGL(mitc4_nl_tiled_kernel<true>, dim3(grid), dim3(64), 0, s, xyz, prop_id, props, n_props, u_e, E, ws, ws, 0, fe,
                       (const int64_t *)nullptr, (const double *)nullptr);
    FS_LAUNCH_CHECK("mitc4_nl_tiled_kernel");
    return check_props(prop_id, E, n_props, status, 0, s);
}

int fs_shell_quad_fint_nl(const double *xyz, const int64_t *conn, const double *u, const int32_t *prop_id,
                          const fs_shell_prop *props, int32_t n_props, int64_t E, double *fe, void *stream) {
    if (E < 0 || n_props <= 0) return fail(FS_EINVAL, "fs_shell_quad_fint_nl: bad sizes");
    if (E == 0) return FS_OK;
    if (!xyz || !conn || !u || !prop_id || !props || !fe) return fail(FS_EINVAL, "fs_shell_quad_fint_nl: null argument");
    hipLaunchKernelGGL(mitc4_nl_tiled_kernel<false>, dim3((unsigned)((E + 15) / 16)), dim3(64), 0, (hipStream_t)stream, xyz, prop_id,
                       props, n_props, (const double *)nullptr, E, (const double *)nullptr, (double *)nullptr, 0, fe, conn, u);
    FS_LAUNCH_CHECK("mitc4_nl_tiled_kernel");
    return FS_OK;
}

int fs_fused_assemble_quad(const double *xyz, const int32_t *prop_id, const fs_shell_prop *props, int32_t n_props,
                           int64_t E, int64_t n_patches, const int64_t *pat_eptr, const int32_t *pat_elems,
                           const int64_t *pat_rptr, const fs_fused_rec *recs, int32_t max_patch_elems,
                           uint32_t flags, double *data, int32_t *status, void *stream) {
    if (E < 0 || n_props <= 0 || n_patches < 0) return fail(FS_EINVAL, "fs_fused_assemble_quad: bad sizes");
    if (E == 0 || n_patches == 0) return FS_OK;
    if (!xyz || !prop_id || !props || !pat_eptr || !pat_elems || !pat_rptr || !recs || !data)
        return fail(FS_EINVAL, "fs_fused_assemble_quad: null argument");
    if (max_patch_elems <= 0 || max_patch_elems > FS_FUSED_MAX_PATCH_ELEMS)
        return fail(FS_EINVAL, "fs_fused_assemble_quad: a patch may hold at most FS_FUSED_MAX_PATCH_ELEMS elements");
    hipStream_t s = (hipStream_t)stream;
    // persistent workgroups: as many as are resident at once (one per CU with the 124 KB image), a multiple of 8
    int cus = 0;
    if (int rc = device_cu_count(&cus)) return rc;
    int64_t wgs = ((int64_t)cus / 8) * 8;
    if (wgs < 8) wgs = 8;
    if (wgs > ((n_patches + 7) / 8) * 8) wgs = ((n_patches + 7) / 8) * 8;
    const unsigned grid = (unsigned)wgs;
#define FS_FUSED_LAUNCH(C)                                                                                           \
    hipLaunchKernelGGL((fused_quad_k_kernel<C, 64>), dim3(grid), dim3(256), 0, s, xyz, prop_id, props, n_props,       \
                       n_patches, pat_eptr, pat_elems, pat_rptr, recs, data, status)
    if (flags & FS_FUSED_MASS)      // (two workgroups per CU with 32 staging slots: 136 B of spills per lane, 1.43 against 1.02 ms)
        hipLaunchKernelGGL((fused_quad_k_kernel<false, 64, FS_OP_M>), dim3(grid), dim3(256), 0, s, xyz, prop_id, props, n_props,
                           n_patches, pat_eptr, pat_elems, pat_rptr, recs, data, status);
    else if (flags & FS_FUSED_HAS_COUPLED) FS_FUSED_LAUNCH(true); else FS_FUSED_LAUNCH(false);
#undef FS_FUSED_LAUNCH
    FS_LAUNCH_CHECK("fused_quad_k_kernel");
    return FS_OK;
}

#ifdef FS_FUSED_TIMING
int fs_debug_fused_timing(unsigned long long *out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fused_t), sizeof(unsigned long long) * 24) != hipSuccess) return FS_EHIP;
    if (reset) {
        unsigned long long z[24] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_fused_t), z, sizeof z) != hipSuccess) return FS_EHIP;
    }
    return FS_OK;
}
#endif

int fs_shell_quad_geom_status(const double *xyz, const int32_t *prop_id, int32_t n_props, int64_t E, int32_t *status,
                              void *stream) {
    if (E < 0) return fail(FS_EINVAL, "fs_shell_quad_geom_status: bad sizes");
    if (E == 0) return FS_OK;
    if (!xyz || !status) return fail(FS_EINVAL, "fs_shell_quad_geom_status: null argument");
    hipLaunchKernelGGL(mitc4_geom_status_kernel, dim3(grid_for(E, 256, INT32_MAX)), dim3(256), 0, (hipStream_t)stream, xyz,
                       prop_id, n_props, E, status);
    FS_LAUNCH_CHECK("mitc4_geom_status_kernel");
    return FS_OK;
}

int fs_fused_plan_check(int64_t n_patches, const int64_t *pat_eptr, const int32_t *pat_elems, const int64_t *pat_rptr,
                        const fs_fused_rec *recs, int64_t E, int32_t max_patch_elems, int64_t nnz, void *stream) {
    if (E < 0 || n_patches < 0 || nnz < 0) return fail(FS_EINVAL, "fs_fused_plan_check: bad sizes");
    if (n_patches == 0) return FS_OK;
    if (!pat_eptr || !pat_elems || !pat_rptr || !recs) return fail(FS_EINVAL, "fs_fused_plan_check: null argument");
    if (max_patch_elems <= 0 || max_patch_elems > FS_FUSED_MAX_PATCH_ELEMS)
        return fail(FS_EINVAL, "fs_fused_plan_check: a patch may hold at most FS_FUSED_MAX_PATCH_ELEMS elements");
    hipStream_t s = (hipStream_t)stream;
    unsigned long long *err = nullptr, host[4] = {0, 0, 0, 0};
    hipError_t e = hipMalloc(&err, sizeof host);
    if (e != hipSuccess) return fail(FS_EHIP, "fs_fused_plan_check: hipMalloc", e);
    e = hipMemsetAsync(err, 0, sizeof host, s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(fused_plan_check_kernel, dim3(grid_for(n_patches, 1, 8192)), dim3(256), 0, s, n_patches, pat_eptr,
                           pat_elems, pat_rptr, recs, E, max_patch_elems, nnz, err);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(host, err, sizeof host, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(err);
    if (e != hipSuccess) return fail(FS_EHIP, "fs_fused_plan_check", e);
    if (host[0]) {
        static const char *const rules[] = {"", "a patch holds no element or more than max_patch_elems",
            "record range of a patch empty or not whole waves", "element index outside [0, E)",
            "contribution names a slot beyond its patch or ti > tj", "chain flags inconsistent",
            "staging slot / rank byte out of range", "destination outside the value array or misaligned"};
        char msg[256];
        snprintf(msg, sizeof msg, "fs_fused_plan_check: %llu violations; first: %s (patch %lld, record %lld)", host[0],
                 rules[host[1] < 8 ? host[1] : 0], (long long)host[2], (long long)host[3]);
        return fail(FS_EINVAL, msg);
    }
    return FS_OK;
}

int fs_gather_dofs(const double *u, const int64_t *dofs, int64_t n, double *u_e, void *stream) {
    if (n <= 0) return n == 0 ? FS_OK : fail(FS_EINVAL, "fs_gather_dofs: n < 0");
    hipLaunchKernelGGL(gather_dofs_kernel, dim3(grid_for(n, 256, 1 << 16)), dim3(256), 0, (hipStream_t)stream, u,
                       dofs, n, u_e);
    FS_LAUNCH_CHECK("gather_dofs_kernel");
    return FS_OK;
}

}  // extern "C"

int fs_device_cu_count_(int *out) { return device_cu_count(out); }
