// femshell_fused_nl.hip -- the Newton path on the ASSEMBLED matrix (fs_fused_assemble_quad_tangent, include/femshell_hip.h).
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (csrc/Makefile).  Citations: /root/reference/src/fem_shell/.
//
// MITC4.compute_tangent_stiffness (elements/MITC4.py:1246-1298, with update_configuration :1067-1084 and the
// total-Lagrangian membrane terms :1086-1244; laminates MITC4_composite.py:499-621) returns K_T = K + T^T sym(K_L + K_sigma) T:
// the linear matrix plus a correction that lives in the translational 3 x 3 sub-blocks only.  K does not depend on u, so
// a Newton iterate needs  K_T(u) = K_csr + sum_e dK_e(u)  -- the assembled linear matrix (fs_fused_assemble_quad, once)
// plus the element-order sums of the corrections, which this kernel forms in the gather form of fused_quad_k_kernel, on
// the SAME patch plan:
//   stage 1  lane = (element slot, Gauss point): frame, Jacobian, dN/dx, dN/dy, the displacement gradient H (GLOBAL
//            components, local derivatives, MITC4.py:1092-1107) and the centre membrane stress -> LDS image of 72 doubles
//            per element; u of the element's nodes is gathered through the plan's copy of the connectivity rows (no u_e array);
//   stage 2  a lane forms up to two contributions (element, local node pair): with P_i = [[nx,0],[0,ny],[ny,nx]],
//            M = P_i^T (detJ h C) P_j (2 x 2), F = [I2;0] + H the reference's  B~_i^T C B~_j - B_L,i^T C B_L,j  is
//            H M F^T + [I2;0] M H^T (no cancellation), + K_sigma on the diagonal, summed over the Gauss points, rotated
//            R^T X R; the contributions of a CSR block are added IN ELEMENT ORDER (DPP chains of the plan);
//   stores   the blocks (I, J), I <= J, of the linear matrix are read ahead of the evaluations from a PACKED copy (36
//            contiguous doubles per block, in the plan's slot order: fs_fused_tangent_pack, once per linear matrix --
//            reading them from the CSR rows themselves costs 1.75 x the bytes, half-used lines), meet the corrections in
//            the wave's staging area and leave as block (I, J) and, transposed, as block (J, I): 288 B read and 576 B
//            written per block pair, no element matrices, no second pass over the matrix.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "femshell_hip.h"
#include "femshell_internal.h"
#include "femshell_geom.h"
#include <rocprim/device/device_scan.hpp>

namespace {

// image components (per element; [component][16] inside a 16-element tile)
constexpr int NL_GP = 15;                 // per Gauss point: nx[4], ny[4], H[3][2], detJ
constexpr int NL_NX = 0, NL_NY = 4, NL_H = 8, NL_DET = 14;
constexpr int NL_R = 60, NL_SM = 69, NL_FN = 72;      // NL_FN: nonlinear nodal forces [node][3] (kernels that form f_nl)
template <bool WF> constexpr int nl_tile() { return (WF ? 84 : 72) * 16; }
constexpr int NZ_SLOT = 21;               // staging slot, in 16-byte pieces (as in fused_quad_k_kernel)

#define NL_EV_SLOT(ev) ((int)((ev) & 255u))
#define NL_EV_TI(ev) ((int)(((ev) >> 8) & 3u))
#define NL_EV_TJ(ev) ((int)(((ev) >> 10) & 3u))
#define NL_EV_TR(ev) ((((ev) >> 12) & 1u) != 0)
#define NL_EV_ACTIVE(ev) ((((ev) >> 13) & 1u) != 0)

__device__ __forceinline__ double nl_row_shr1(double v) {      // lane i reads lane i - 1 of its 16-lane row
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x111, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x111, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

// one contribution: the translational 3 x 3 correction of pair (ti <= tj) of the element in image slot NL_EV_SLOT(ev),
// global axes, transposed when the record says so
template <bool WF>
__device__ __forceinline__ void dkt_eval(uint32_t ev, const double *lds, const int *s_pid,
                                         const fs_shell_prop *__restrict__ props, double (&Y)[WF ? 12 : 9]) {
#pragma clang fp contract(fast)
    const int slot = NL_EV_SLOT(ev), ti = NL_EV_TI(ev), tj = NL_EV_TJ(ev);
    const double *ex = lds + (slot >> 4) * nl_tile<WF>() + (slot & 15);
    const fs_shell_prop *pr = props + s_pid[slot];
    double C[3][3];
    for (int a = 0; a < 9; a++) (&C[0][0])[a] = pr->C_raw[a];
    const double h = pr->h;
    const double Sm00 = ex[(NL_SM + 0) << 4], Sm01 = ex[(NL_SM + 1) << 4], Sm11 = ex[(NL_SM + 2) << 4];
    double X[3][3];
    for (int a = 0; a < 9; a++) (&X[0][0])[a] = 0.0;
#pragma unroll 1
    for (int g = 0; g < 4; g++) {
        const double *eg = ex + ((g * NL_GP) << 4);
        const double nxi = eg[(NL_NX + ti) << 4], nyi = eg[(NL_NY + ti) << 4];
        const double nxj = eg[(NL_NX + tj) << 4], nyj = eg[(NL_NY + tj) << 4];
        double H[3][2];
        for (int t = 0; t < 6; t++) (&H[0][0])[t] = eg[(NL_H + t) << 4];
        const double det = eg[NL_DET << 4];
        const double wh = det * h;
        double CP[3][2], M[2][2];
        for (int a = 0; a < 3; a++) {
            CP[a][0] = C[a][0] * nxj + C[a][2] * nyj;
            CP[a][1] = C[a][1] * nyj + C[a][2] * nxj;
        }
        for (int n = 0; n < 2; n++) {
            M[0][n] = wh * (nxi * CP[0][n] + nyi * CP[2][n]);
            M[1][n] = wh * (nyi * CP[1][n] + nxi * CP[2][n]);
        }
        double HM[3][2];
        for (int k = 0; k < 3; k++)
            for (int n = 0; n < 2; n++) HM[k][n] = H[k][0] * M[0][n] + H[k][1] * M[1][n];
        const double ks = det * (nxi * (Sm00 * nxj + Sm01 * nyj) + nyi * (Sm01 * nxj + Sm11 * nyj));
        for (int k = 0; k < 3; k++)
            for (int l = 0; l < 3; l++) {
                // H M F^T with F = [I2;0] + H
                double v = HM[k][0] * H[l][0] + HM[k][1] * H[l][1];
                if (l < 2) v += HM[k][l];
                // [I2;0] M H^T
                if (k < 2) v += M[k][0] * H[l][0] + M[k][1] * H[l][1];
                if (k == l) v += ks;
                X[k][l] += v;
            }
    }
    if (ti == tj)
        for (int k = 0; k < 3; k++)
            for (int l = k + 1; l < 3; l++) { const double m = 0.5 * (X[k][l] + X[l][k]); X[k][l] = m; X[l][k] = m; }
    double R[3][3], Yt[3][3], T[3][3];
    for (int t = 0; t < 9; t++) (&R[0][0])[t] = ex[(NL_R + t) << 4];
    for (int a = 0; a < 3; a++)
        for (int l = 0; l < 3; l++) Yt[a][l] = R[0][a] * X[0][l] + R[1][a] * X[1][l] + R[2][a] * X[2][l];
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) T[a][b] = Yt[a][0] * R[0][b] + Yt[a][1] * R[1][b] + Yt[a][2] * R[2][b];
    if (ti == tj)      // a diagonal block stays exactly symmetric (sums of symmetric terms are)
        for (int a = 0; a < 3; a++)
            for (int b = a + 1; b < 3; b++) { const double m = 0.5 * (T[a][b] + T[b][a]); T[a][b] = m; T[b][a] = m; }
    const bool tr = NL_EV_TR(ev);
    for (int k = 0; k < 3; k++)
        for (int l = 0; l < 3; l++) Y[k * 3 + l] = tr ? T[l][k] : T[k][l];
    if constexpr (WF)      // a contribution to a DIAGONAL block also carries the element's nonlinear force at that node
        for (int a = 0; a < 3; a++) Y[9 + a] = (ti == tj) ? ex[(NL_FN + 3 * ti + a) << 4] : -0.0;
}

// SP staging slots per wave: 64 -> one workgroup per CU, 32 -> two.  WF: the kernel also forms the element-order sums
// f_nl of the nonlinear nodal forces (quad_fnl_lane in stage 1, three more values on the chains of the diagonal blocks)
template <int SP, bool WF>
__global__ __launch_bounds__(256, (SP <= 32 ? 2 : 1)) void fused_quad_dkt_kernel(
    const double *__restrict__ xyz, const int4 *__restrict__ pat_conn, const double *__restrict__ u,
    const int32_t *__restrict__ prop_id, const fs_shell_prop *__restrict__ props, int32_t n_props, int64_t n_pat,
    const int64_t *__restrict__ pat_eptr, const int32_t *__restrict__ pat_elems, const int64_t *__restrict__ pat_rptr,
    const fs_fused_rec *__restrict__ recs, const int64_t *__restrict__ wave_off, const double *__restrict__ packed,
    double *__restrict__ data, double *__restrict__ f_nl) {
    constexpr int NL_TILE = nl_tile<WF>();
    constexpr int NV = WF ? 12 : 9;                 // values per contribution: the 3 x 3 correction (+ 3 force components)
    constexpr int NQ = (3 * SP + 63) / 64;          // direct-copy pieces per lane and pass
    constexpr int NB = (18 * SP + 63) / 64;         // 16-byte pieces of the pass's linear blocks per lane
    __shared__ double lds[4 * NL_TILE];
    __shared__ int s_pid[64];
    __shared__ double2 s_stage[4 * NZ_SLOT * SP];
    __shared__ unsigned char s_tab[4 * 128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Q = lane >> 2, g = lane & 3, slot = wave * 16 + Q;
    // persistent workgroups, XCD k walks a contiguous eighth of the patches (as fused_quad_k_kernel)
    const int xcd = blockIdx.x & 7, wpx = (int)(gridDim.x >> 3);
    const int64_t per_xcd = (n_pat + 7) / 8;
    const int64_t pend = ((xcd + 1) * per_xcd < n_pat) ? (xcd + 1) * per_xcd : n_pat;
    double2 *stg = s_stage + wave * (NZ_SLOT * SP);
    unsigned char *tab = s_tab + wave * 128;

    int64_t p = xcd * per_xcd + (blockIdx.x >> 3);
    if (p >= pend) return;
    // inputs of an element slot: coordinates, property id, u of its four nodes (ids from the plan's copy of the
    // connectivity rows, so that they arrive together with the element index: two dependent loads, not three)
    auto fetch_inputs = [&](int64_t e, int4 nid, double (&xo)[12], double (&uo)[24], int &pido) __attribute__((always_inline)) {
        const double2 *xp = reinterpret_cast<const double2 *>(xyz + e * 12);
        for (int t = 0; t < 6; t++) { const double2 v = xp[t]; xo[2 * t] = v.x; xo[2 * t + 1] = v.y; }
        const int nn[4] = {nid.x, nid.y, nid.z, nid.w};
        for (int i = 0; i < 4; i++) {
            const double2 *up = reinterpret_cast<const double2 *>(u + (int64_t)nn[i] * 6);
            for (int t = 0; t < 3; t++) { const double2 v = up[t]; uo[6 * i + 2 * t] = v.x; uo[6 * i + 2 * t + 1] = v.y; }
        }
        pido = prop_id[e];
    };
    // PF (one workgroup per CU): headers travel two patches ahead, element / node ids one patch ahead (fetched at the top
    // of the current patch), coordinates / u / property id of the next patch are requested after stage 1 and arrive under
    // stage 2.  Two workgroups per CU fetch on the spot (the other workgroup covers the latency; no registers to spare).
    constexpr bool PF = SP > 32;
    int64_t e0 = pat_eptr[p];
    int ne = (int)(pat_eptr[p + 1] - e0);
    int64_t r0 = pat_rptr[p], r1 = pat_rptr[p + 1];
    int64_t e0n = e0, r0n = r0, r1n = r1;
    int nen = ne;
    double xc[12], uc[24];
    int pidc = 0;
    if constexpr (PF) {
        if (p + wpx < pend) {
            e0n = pat_eptr[p + wpx]; nen = (int)(pat_eptr[p + wpx + 1] - e0n);
            r0n = pat_rptr[p + wpx]; r1n = pat_rptr[p + wpx + 1];
        }
        const int64_t si = e0 + ((slot < ne) ? slot : ne - 1);
        fetch_inputs(pat_elems[si], pat_conn[si], xc, uc, pidc);
    }
    for (;;) {
        const int64_t pn = p + wpx, pnn = pn + wpx;
        const bool more = pn < pend;
        int64_t e0nn = 0, r0nn = 0, r1nn = 0, en = 0;
        int nenn = 0;
        int4 nidn = make_int4(0, 0, 0, 0);
        if constexpr (PF) {
            // (unconditional loads from clamped indices: a load inside a branch is waited for at the end of the branch)
            const int64_t sn = more ? e0n + ((slot < nen) ? slot : nen - 1) : e0;
            en = pat_elems[sn]; nidn = pat_conn[sn];
            const int64_t pq = (pnn < pend) ? pnn : p;
            e0nn = pat_eptr[pq]; nenn = (int)(pat_eptr[pq + 1] - e0nn);
            r0nn = pat_rptr[pq]; r1nn = pat_rptr[pq + 1];
        } else {
            e0 = pat_eptr[p]; ne = (int)(pat_eptr[p + 1] - e0);
            r0 = pat_rptr[p]; r1 = pat_rptr[p + 1];
        }
        // this lane's record of the first round: requested now, used after stage 1
        int4 pa = make_int4(-1, -1, -1, -1), pb = pa, pc4 = make_int4(0, 0, 0, 0), pd = make_int4(0, 0, 0, -1);
        if (r0 + tid < r1) {
            const int4 *rp = reinterpret_cast<const int4 *>(recs + r0 + tid);
            pa = rp[0]; pb = rp[1]; pc4 = rp[2]; pd = rp[3];
        }
        // ---------------- stage 1 ----------------
        if (wave * 16 < ne) {
#pragma clang fp contract(fast)
            if constexpr (!PF) {
                const int64_t si = e0 + ((slot < ne) ? slot : ne - 1);
                fetch_inputs(pat_elems[si], pat_conn[si], xc, uc, pidc);
            }
            double x[4][3], ue[4][6];
            for (int t = 0; t < 12; t++) (&x[0][0])[t] = xc[t];
            for (int t = 0; t < 24; t++) (&ue[0][0])[t] = uc[t];
            const int pid = safe_pid(pidc, n_props);
            const fs_shell_prop *pr = props + pid;
            QuadGeom geo;
            quad_geometry(x, geo);
            const double R[3][3] = {{geo.e1[0], geo.e1[1], geo.e1[2]}, {geo.e2[0], geo.e2[1], geo.e2[2]},
                                    {geo.e3[0], geo.e3[1], geo.e3[2]}};
            const bool nonlinear = pr->nonlinear != 0;
            // membrane stress at the centre (MITC4.py:1227-1240; laminates MITC4_composite.py:499-508)
            double Sm[3];
            {
                double ul[4][3], rl[4][3];
                for (int i = 0; i < 4; i++)
                    for (int a = 0; a < 3; a++) {
                        ul[i][a] = R[a][0] * ue[i][0] + R[a][1] * ue[i][1] + R[a][2] * ue[i][2];
                        rl[i][a] = R[a][0] * ue[i][3] + R[a][1] * ue[i][4] + R[a][2] * ue[i][5];
                    }
                double dxi[4], deta[4], ia, ib, ic, id;
                dshape(0.0, 0.0, dxi, deta);
                jacobian(geo, dxi, deta, ia, ib, ic, id);
                double eps[3] = {0, 0, 0}, kap[3] = {0, 0, 0};
                for (int i = 0; i < 4; i++) {
                    const double nx = ia * dxi[i] + ib * deta[i], ny = ic * dxi[i] + id * deta[i];
                    eps[0] += nx * ul[i][0]; eps[1] += ny * ul[i][1]; eps[2] += ny * ul[i][0] + nx * ul[i][1];
                    kap[0] += nx * rl[i][1]; kap[1] += -ny * rl[i][0]; kap[2] += ny * rl[i][1] - nx * rl[i][0];
                }
                const double h = pr->h;
                double sig[3];
                if (pr->flags & FS_PROP_LAMINATE) {
                    for (int a = 0; a < 3; a++)
                        sig[a] = ((pr->A[a * 3] * eps[0] + pr->A[a * 3 + 1] * eps[1] + pr->A[a * 3 + 2] * eps[2]) +
                                  (pr->B[a * 3] * kap[0] + pr->B[a * 3 + 1] * kap[1] + pr->B[a * 3 + 2] * kap[2])) / h;
                } else {
                    for (int a = 0; a < 3; a++)
                        sig[a] = pr->C_raw[a * 3] * eps[0] + pr->C_raw[a * 3 + 1] * eps[1] + pr->C_raw[a * 3 + 2] * eps[2];
                }
                Sm[0] = sig[0] * h; Sm[1] = sig[2] * h; Sm[2] = sig[1] * h;      // Sm00, Sm01, Sm11
            }
            // Gauss point g of this lane
            const double xi = (g == 1 || g == 2) ? kGP : -kGP, eta = (g >= 2) ? kGP : -kGP;
            double dxi[4], deta[4], ia, ib, ic, id;
            dshape(xi, eta, dxi, deta);
            const double detJ = jacobian(geo, dxi, deta, ia, ib, ic, id);
            double *ex = lds + wave * NL_TILE + Q + ((g * NL_GP) << 4);
            double nx[4], ny[4];
            for (int i = 0; i < 4; i++) {
                nx[i] = ia * dxi[i] + ib * deta[i]; ny[i] = ic * dxi[i] + id * deta[i];
                ex[(NL_NX + i) << 4] = nx[i]; ex[(NL_NY + i) << 4] = ny[i];
            }
            for (int c = 0; c < 3; c++) {
                const double h0 = dots4(ue[0][c], ue[1][c], ue[2][c], ue[3][c], nx[0], nx[1], nx[2], nx[3]);
                const double h1 = dots4(ue[0][c], ue[1][c], ue[2][c], ue[3][c], ny[0], ny[1], ny[2], ny[3]);
                // rows of the table without the geometric terms: zero gradient and stress -> a correction of +0.0
                ex[(NL_H + 2 * c) << 4] = nonlinear ? h0 : 0.0;
                ex[(NL_H + 2 * c + 1) << 4] = nonlinear ? h1 : 0.0;
            }
            ex[NL_DET << 4] = detJ;
            double *ee = lds + wave * NL_TILE + Q;
            if constexpr (WF) {
                double Hm[3][2], fn[3];
                for (int c = 0; c < 3; c++) {
                    Hm[c][0] = dots4(ue[0][c], ue[1][c], ue[2][c], ue[3][c], nx[0], nx[1], nx[2], nx[3]);
                    Hm[c][1] = dots4(ue[0][c], ue[1][c], ue[2][c], ue[3][c], ny[0], ny[1], ny[2], ny[3]);
                }
                quad_fnl_lane(x, ue, geo, pr, g, xi, eta, detJ, ia, ib, ic, id, nx, ny, Hm, R, fn);
                for (int a = 0; a < 3; a++) ee[(NL_FN + 3 * g + a) << 4] = nonlinear ? fn[a] : 0.0;
            }
            if (g == 0) {
                for (int t = 0; t < 9; t++) ee[(NL_R + t) << 4] = (&R[0][0])[t];
                s_pid[slot] = pid;
            } else if (g == 1) {
                for (int t = 0; t < 3; t++) ee[(NL_SM + t) << 4] = nonlinear ? Sm[t] : 0.0;
            }
        }
        // inputs of the next patch: its ids have arrived under stage 1 (the empty asm keeps the address arithmetic, and
        // with it the wait for the ids, from moving to the top of the loop); they arrive under stage 2
        double xn[12], un[24];
        int pidn = 0;
        if constexpr (PF) {
            int64_t en_use = en;
            int4 nid_use = nidn;
            asm volatile("" : "+v"(en_use), "+v"(nid_use.x), "+v"(nid_use.y), "+v"(nid_use.z), "+v"(nid_use.w));
            fetch_inputs(en_use, nid_use, xn, un, pidn);
        }
        __syncthreads();

        // ---------------- stage 2 ----------------
        for (int64_t rb = r0; rb < r1; rb += 256) {
            int4 ra = pa, rb4 = pb, rc = pc4, rd = pd;
            if (rb != r0) {
                ra = make_int4(-1, -1, -1, -1); rb4 = ra; rc = make_int4(0, 0, 0, 0); rd = make_int4(0, 0, 0, -1);
                if (rb + tid < r1) {
                    const int4 *rp = reinterpret_cast<const int4 *>(recs + rb + tid);
                    ra = rp[0]; rb4 = rp[1]; rc = rp[2]; rd = rp[3];
                }
            }
            const uint32_t ev0 = (uint32_t)rd.x, ev1 = (uint32_t)rd.y, fl = (uint32_t)rd.z, slots = (uint32_t)rd.w;
            const bool out0 = (fl & FS_FL_OUT0) != 0, two = (fl & FS_FL_OUT1) != 0, chain = (fl & FS_FL_CHAIN) != 0;
            const int64_t dst0 = ((int64_t)(uint32_t)ra.x) | ((int64_t)ra.y << 32), dst1 = ((int64_t)(uint32_t)ra.z) | ((int64_t)ra.w << 32);
            const int64_t dst0t = ((int64_t)(uint32_t)rb4.x) | ((int64_t)rb4.y << 32), dst1t = ((int64_t)(uint32_t)rb4.z) | ((int64_t)rb4.w << 32);
            const int s0d = out0 ? (int)(slots & 255u) : 255;
            const int s0t = (out0 && dst0t >= 0) ? (int)((slots >> 8) & 255u) : 255;
            const int s1d = two ? (int)((slots >> 16) & 255u) : 255;
            const int s1t = (two && dst1t >= 0) ? (int)((slots >> 24) & 255u) : 255;
            const int nd = __popcll(__ballot(s0d != 255)) + __popcll(__ballot(s1d != 255));
            const int ntr = __popcll(__ballot(s0t != 255)) + __popcll(__ballot(s1t != 255));

            // node of a DIAGONAL block this lane emits (its force sums leave with it): requested now, used after the sums
            int fnode0 = -1, fnode1 = -1;
            if constexpr (WF) {
                const uint32_t evA = (two || NL_EV_ACTIVE(ev0)) ? ev0 : ev1, evB = ev1;
                const bool d0 = out0 && NL_EV_ACTIVE(evA) && NL_EV_TI(evA) == NL_EV_TJ(evA);
                const bool d1 = two && NL_EV_ACTIVE(evB) && NL_EV_TI(evB) == NL_EV_TJ(evB);
                const int32_t *pcn = reinterpret_cast<const int32_t *>(pat_conn + e0);
                const int nA = pcn[d0 ? 4 * NL_EV_SLOT(evA) + NL_EV_TI(evA) : 0];
                const int nB = pcn[d1 ? 4 * NL_EV_SLOT(evB) + NL_EV_TI(evB) : 0];
                fnode0 = d0 ? nA : -1; fnode1 = d1 ? nB : -1;
            }
            // destinations of the pass's blocks into their staging slots, rank table of the transposed copies
            auto stage_addresses = [&](int s0) __attribute__((always_inline)) {
                if (nd > SP) {
                    tab[lane] = 255; tab[lane + 64] = 255;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
                const int q0d = s0d - s0, q1d = s1d - s0;
                if (s0d != 255 && q0d >= 0 && q0d < SP) {
                    double2 *sp = stg + q0d * NZ_SLOT;
                    sp[18] = make_double2(__longlong_as_double(dst0), __longlong_as_double((long long)rc.x));
                    sp[19] = make_double2(__longlong_as_double(dst0t), __longlong_as_double((long long)rc.z));
                    if (s0t != 255) tab[s0t] = (unsigned char)q0d;
                }
                if (s1d != 255 && q1d >= 0 && q1d < SP) {
                    double2 *sp = stg + q1d * NZ_SLOT;
                    sp[18] = make_double2(__longlong_as_double(dst1), __longlong_as_double((long long)rc.y));
                    sp[19] = make_double2(__longlong_as_double(dst1t), __longlong_as_double((long long)rc.w));
                    if (s1t != 255) tab[s1t] = (unsigned char)q1d;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            };
            // the linear blocks (I, J), I <= J, of the wave's pass: 36 contiguous doubles each, in slot order, in the PACKED
            // copy of the linear matrix (fs_fused_tangent_pack) -- lane <-> 16-byte piece, whole lines, no address needed
            // (unconditional loads from clamped indices: a load inside a branch is waited for at the end of the branch)
            double bvx[NB], bvy[NB];           // (scalars: an array of double2 filled from memory stays an alloca)
            const int64_t wv = (rb >> 6) + ((rb + 64 * wave < r1) ? wave : 0);      // this wave of records (clamped: a patch's
                                                                                   // last round may have fewer than four)
            const double2 *pk = reinterpret_cast<const double2 *>(packed) + wave_off[wv] * 18;
            auto load_base = [&](int s0) __attribute__((always_inline)) {
                const int nhere = (nd - s0 < SP) ? nd - s0 : SP;
                const double2 *pp = pk + (int64_t)s0 * 18;
                for (int q = 0; q < NB; q++) {
                    const int j = q * 64 + lane;
                    const double2 t2 = pp[(j < 18 * nhere) ? j : 0];
                    bvx[q] = t2.x; bvy[q] = t2.y;
                }
            };
            // ---- passes over the wave's blocks: linear block + correction -> staging -> (I, J) and (J, I) ----
            double X[NV], Y[NV];
            for (int s0 = 0; s0 < nd; s0 += SP) {       // (a wave without output blocks has nothing to evaluate)
                load_base(s0);
                stage_addresses(s0);
                if (s0 == 0) {
                    // ---- the lane's two contributions, element-order sums (as fused_quad_k_kernel, on NV values) ----
                    const double z0 = (fl & FS_FL_ZERO_START) ? 0.0 : -0.0;
                    if (NL_EV_ACTIVE(ev0)) dkt_eval<WF>(ev0, lds, s_pid, props, Y);
                    else
                        for (int t = 0; t < NV; t++) Y[t] = -0.0;
                    for (int t = 0; t < NV; t++) X[t] = z0 + Y[t];
                    if (NL_EV_ACTIVE(ev1)) dkt_eval<WF>(ev1, lds, s_pid, props, Y);
                    else
                        for (int t = 0; t < NV; t++) Y[t] = -0.0;
                    const int pos = (int)((fl >> 16) & 15u);
                    const int maxc = __builtin_amdgcn_readfirstlane((int)((fl >> 8) & 255u));
                    double S[NV];
                    for (int t = 0; t < NV; t++) S[t] = X[t] + Y[t];
                    for (int it = 1; it < maxc; it++)
                        for (int t = 0; t < NV; t++) {
                            const double T = nl_row_shr1(S[t]);
                            const double v = (T + X[t]) + Y[t];
                            S[t] = (chain && pos == it) ? v : S[t];
                        }
                    for (int t = 0; t < NV; t++) X[t] = two ? X[t] : S[t];
                    if constexpr (WF) {      // one writer per node: the lane that ends the chain of the node's diagonal block
                        if (fnode0 >= 0) {
                            double2 *fo = reinterpret_cast<double2 *>(f_nl + (int64_t)fnode0 * 6);
                            fo[0] = make_double2(X[9] + 0.0, X[10] + 0.0); fo[1] = make_double2(X[11] + 0.0, 0.0);
                            fo[2] = make_double2(0.0, 0.0);
                        }
                        if (fnode1 >= 0) {
                            double2 *fo = reinterpret_cast<double2 *>(f_nl + (int64_t)fnode1 * 6);
                            fo[0] = make_double2(Y[9] + 0.0, Y[10] + 0.0); fo[1] = make_double2(Y[11] + 0.0, 0.0);
                            fo[2] = make_double2(0.0, 0.0);
                        }
                    }
                }
                const int nhere = (nd - s0 < SP) ? nd - s0 : SP;
                for (int q = 0; q < NB; q++) {
                    const int j = q * 64 + lane;
                    const int bk = (j * 3641) >> 16;            // j / 18 for j < 18 * 64
                    if (j < 18 * nhere) stg[bk * NZ_SLOT + (j - 18 * bk)] = make_double2(bvx[q], bvy[q]);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const int q0d = s0d - s0, q1d = s1d - s0;
                if (s0d != 255 && q0d >= 0 && q0d < SP) {
                    double *sp = reinterpret_cast<double *>(stg + q0d * NZ_SLOT);
                    for (int k = 0; k < 3; k++)
                        for (int l = 0; l < 3; l++) sp[k * 6 + l] += X[k * 3 + l];
                }
                if (s1d != 255 && q1d >= 0 && q1d < SP) {
                    double *sp = reinterpret_cast<double *>(stg + q1d * NZ_SLOT);
                    for (int k = 0; k < 3; k++)
                        for (int l = 0; l < 3; l++) sp[k * 6 + l] += Y[k * 3 + l];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                // direct copies: piece pc -> (block pc / 3, piece pc % 3)
                for (int q = 0; q < NQ; q++) {
                    const int pc = q * 64 + lane;
                    const int bk = (pc * 21846) >> 16, cc = pc - 3 * bk;
                    if (pc < 3 * nhere) {
                        const double2 *sp = stg + bk * NZ_SLOT;
                        const double2 t = sp[18];
                        double *d = data + (int64_t)__double_as_longlong(t.x) + 2 * cc;
                        const int64_t str = (int64_t)__double_as_longlong(t.y);
                        double vx[6], vy[6];
                        for (int k = 0; k < 6; k++) { const double2 t2 = sp[k * 3 + cc]; vx[k] = t2.x; vy[k] = t2.y; }
                        for (int k = 0; k < 6; k++) *reinterpret_cast<double2 *>(d + k * str) = make_double2(vx[k], vy[k]);
                    }
                }
                // transposed copies in address order of the copies: rank o -> slot tab[o]
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
                        const double2 *sp = stg + tslot[q] * NZ_SLOT;
                        const double2 t = sp[19];
                        double *d = data + (int64_t)__double_as_longlong(t.x) + 2 * cc;
                        const int64_t str = (int64_t)__double_as_longlong(t.y);
                        const double *sd = reinterpret_cast<const double *>(sp) + 12 * cc;
                        double vx[6], vy[6];
                        for (int k = 0; k < 6; k++) { vx[k] = sd[k]; vy[k] = sd[6 + k]; }
                        for (int k = 0; k < 6; k++) *reinterpret_cast<double2 *>(d + k * str) = make_double2(vx[k], vy[k]);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (!more) break;
        __syncthreads();        // every wave is done with the image
        p = pn;
        if constexpr (PF) {
            e0 = e0n; ne = nen; r0 = r0n; r1 = r1n;
            e0n = e0nn; nen = nenn; r0n = r0nn; r1n = r1nn;
            for (int t = 0; t < 12; t++) xc[t] = xn[t];
            for (int t = 0; t < 24; t++) uc[t] = un[t];
            pidc = pidn;
        }
    }
}

// number of output blocks (I <= J) of every wave-round of the plan
__global__ __launch_bounds__(256) void tangent_count_kernel(const fs_fused_rec *__restrict__ recs, int64_t n_waves,
                                                            int64_t *__restrict__ counts) {
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n_waves) return;
    const uint32_t fl = recs[w * 64 + (threadIdx.x & 63)].flags;
    const int n = __popcll(__ballot((fl & FS_FL_OUT0) != 0)) + __popcll(__ballot((fl & FS_FL_OUT1) != 0));
    if ((threadIdx.x & 63) == 0) counts[w] = n;
}

// packed[wave_off[w] + slot] = block (I, J) of the CSR values, row-major 6 x 6
__global__ __launch_bounds__(256) void tangent_pack_kernel(const fs_fused_rec *__restrict__ recs, int64_t n_waves,
                                                           const int64_t *__restrict__ wave_off,
                                                           const double *__restrict__ data, double *__restrict__ packed) {
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n_waves) return;
    const fs_fused_rec r = recs[w * 64 + (threadIdx.x & 63)];
    const int64_t off = wave_off[w];
    for (int o = 0; o < 2; o++) {
        if (!(r.flags & (o ? FS_FL_OUT1 : FS_FL_OUT0))) continue;
        const int slot = (int)((r.slots >> (16 * o)) & 255u);
        const double *src = data + r.dst[o];
        double2 *dst = reinterpret_cast<double2 *>(packed + (off + slot) * 36);
        for (int k = 0; k < 6; k++)
            for (int c = 0; c < 3; c++) dst[k * 3 + c] = *reinterpret_cast<const double2 *>(src + (int64_t)k * r.stride[o] + 2 * c);
    }
}

}  // namespace

extern "C" int fs_fused_tangent_offsets(int64_t n_recs, const fs_fused_rec *recs, int64_t *wave_off, int64_t *n_blocks,
                                        void *stream) {
    if (!n_blocks) return fs_fail_(FS_EINVAL, "fs_fused_tangent_offsets: null output", 0);
    *n_blocks = 0;
    if (n_recs < 0 || (n_recs & 63)) return fs_fail_(FS_EINVAL, "fs_fused_tangent_offsets: the records of a plan are whole waves", 0);
    if (!wave_off || (n_recs && !recs)) return fs_fail_(FS_EINVAL, "fs_fused_tangent_offsets: null argument", 0);
    hipStream_t s = (hipStream_t)stream;
    const int64_t nw = n_recs >> 6;
    hipError_t e = hipMemsetAsync(wave_off + nw, 0, sizeof(int64_t), s);
    if (e == hipSuccess && nw) {
        hipLaunchKernelGGL(tangent_count_kernel, dim3((unsigned)((nw + 3) / 4)), dim3(256), 0, s, recs, nw, wave_off);
        e = hipGetLastError();
    }
    void *tmp = nullptr;
    size_t tmp_bytes = 0;
    if (e == hipSuccess)
        e = rocprim::exclusive_scan(nullptr, tmp_bytes, wave_off, wave_off, (int64_t)0, (size_t)(nw + 1), rocprim::plus<int64_t>(), s);
    if (e == hipSuccess) e = hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16);
    if (e == hipSuccess)
        e = rocprim::exclusive_scan(tmp, tmp_bytes, wave_off, wave_off, (int64_t)0, (size_t)(nw + 1), rocprim::plus<int64_t>(), s);
    if (e == hipSuccess) e = hipMemcpyAsync(n_blocks, wave_off + nw, sizeof(int64_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (tmp) (void)hipFree(tmp);
    if (e != hipSuccess) return fs_fail_(FS_EHIP, "fs_fused_tangent_offsets", (int)e);
    return FS_OK;
}

extern "C" int fs_fused_tangent_pack(int64_t n_recs, const fs_fused_rec *recs, const int64_t *wave_off, const double *k_lin,
                                     double *packed, void *stream) {
    if (n_recs < 0 || (n_recs & 63)) return fs_fail_(FS_EINVAL, "fs_fused_tangent_pack: the records of a plan are whole waves", 0);
    if (n_recs == 0) return FS_OK;
    if (!recs || !wave_off || !k_lin || !packed) return fs_fail_(FS_EINVAL, "fs_fused_tangent_pack: null argument", 0);
    const int64_t nw = n_recs >> 6;
    hipLaunchKernelGGL(tangent_pack_kernel, dim3((unsigned)((nw + 3) / 4)), dim3(256), 0, (hipStream_t)stream, recs, nw, wave_off,
                       k_lin, packed);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fs_fail_(FS_EHIP, "tangent_pack_kernel", (int)e);
    return FS_OK;
}

extern "C" int fs_fused_assemble_quad_tangent(const double *xyz, const int32_t *pat_conn, const double *u, const int32_t *prop_id,
                                              const fs_shell_prop *props, int32_t n_props, int64_t E, int64_t n_patches,
                                              const int64_t *pat_eptr, const int32_t *pat_elems, const int64_t *pat_rptr,
                                              const fs_fused_rec *recs, int32_t max_patch_elems, const int64_t *wave_off,
                                              const double *k_lin_packed, double *k_t, double *f_nl, uint32_t flags,
                                              void *stream) {
    if (E < 0 || n_props <= 0 || n_patches < 0) return fs_fail_(FS_EINVAL, "fs_fused_assemble_quad_tangent: bad sizes", 0);
    if (E == 0 || n_patches == 0) return FS_OK;
    if (!xyz || !pat_conn || !u || !prop_id || !props || !pat_eptr || !pat_elems || !pat_rptr || !recs || !wave_off || !k_lin_packed || !k_t)
        return fs_fail_(FS_EINVAL, "fs_fused_assemble_quad_tangent: null argument", 0);
    if (max_patch_elems <= 0 || max_patch_elems > FS_FUSED_MAX_PATCH_ELEMS)
        return fs_fail_(FS_EINVAL, "fs_fused_assemble_quad_tangent: a patch may hold at most FS_FUSED_MAX_PATCH_ELEMS elements", 0);
    int cus = 0;
    if (int rc = fs_device_cu_count_(&cus)) return rc;
    // (the 32-slot variant has no room for the force image: with f_nl the flag is ignored)
    const int per_cu = ((flags & FS_TANGENT_TWO_WGS_PER_CU) && !f_nl) ? 2 : 1;
    int64_t wgs = ((int64_t)cus * per_cu / 8) * 8;
    if (wgs < 8) wgs = 8;
    if (wgs > ((n_patches + 7) / 8) * 8) wgs = ((n_patches + 7) / 8) * 8;
    hipStream_t s = (hipStream_t)stream;
#define FS_DKT_LAUNCH(SP_, WF_)                                                                                          \
    hipLaunchKernelGGL((fused_quad_dkt_kernel<SP_, WF_>), dim3((unsigned)wgs), dim3(256), 0, s, xyz,                        \
                       reinterpret_cast<const int4 *>(pat_conn), u, prop_id, props, n_props, n_patches, pat_eptr, pat_elems,  \
                       pat_rptr, recs, wave_off, k_lin_packed, k_t, f_nl)
    if (f_nl) FS_DKT_LAUNCH(64, true);
    else if (per_cu == 1) FS_DKT_LAUNCH(64, false);
    else FS_DKT_LAUNCH(32, false);
#undef FS_DKT_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fs_fail_(FS_EHIP, "fused_quad_dkt_kernel", (int)e);
    return FS_OK;
}
