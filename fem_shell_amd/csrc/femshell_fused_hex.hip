// femshell_fused_hex.hip -- fused evaluate-and-assemble for HEXA8 (fs_fused_assemble_hex8, include/femshell_hip.h).
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (csrc/Makefile).  Citations: /root/reference/src/fem_shell/.
//
// One launch replaces HEXA8.K of every element (elements/SOLID.py:543-567) and the ADD_VALUES loop over the element
// matrices (core/assembler.py:287-303).  Gather form, one workgroup (512 lanes, persistent, XCD-contiguous patch ranges)
// per PATCH of nodes:
//   phase 1  lane = (element slot, Gauss point): Jacobian, inverse, w detJ and the physical gradients of the 8 shape
//            functions (SOLID.py:164-266, rule :1745-1766, N :1768-1818) with the expressions of solid_kernel, scaled by
//            sqrt(w detJ), into LDS as 16-byte pieces, so that phase 2 reads with ds_read_b128 (256 B per LDS clock;
//            ds_read2_b64 on 8-byte aligned triples moves half of that): (g_x, g_y) at [slot][q][node], g_z of two
//            Gauss points at [slot][q / 2][node]; odd slot strides (65 and 33 pieces) spread the slots over the banks,
//            and the lanes of a wave are numbered (slot fastest) so that the stores of a lane group are conflict-free.
//            Elements of the one-element halo are evaluated by every patch that touches them.
//   phase 2  lane = up to 8 contributions (element slot, local node pair) in a row.  C is constant over the element, so
//            a contribution contracts over the Gauss points first: G = sum_q g_i g_j^T (72 FMA), K = C : G (81 FMA).
//            The contributions of a CSR block follow each other on one lane and are added in element order,
//            0 + x0 + x1 + ...; with ONE material in the model (every SOLID-family model of the reference has one) the
//            G of the block's elements are added in that order and contracted once per block with C in scalar
//            registers (sum_e C : G_e = C : sum_e G_e: the same products re-associated).  A contribution is formed for
//            the pair ordered by GLOBAL node index and transposed where the block is (larger, smaller); diagonal ones
//            are symmetrised (SOLID.py:565) -- the assembled matrix is exactly symmetric.  A finished block stays in
//            registers until the phase ends.
//   stores   the blocks go to an LDS image of the patch's CSR slabs (overlaying the gradients) and leave as contiguous
//            runs: the 3 rows of a node are adjacent in CSR, and so are the slabs of consecutively numbered nodes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "femshell_hip.h"
#include "femshell_internal.h"
#include "femshell_geom.h"

namespace {

constexpr int HX_SA = 65, HX_SB = 33;            // 16-byte pieces per element slot: (g_x, g_y) and g_z pairs, + 1 (odd)
constexpr int HX_PIECES = FS_HEX_MAX_PATCH_ELEMS * (HX_SA + HX_SB);
constexpr int HX_NT = FS_HEX_MAX_PATCH_LANES;    // 512
static_assert(FS_HEX_STAGE_DOUBLES <= 2 * HX_PIECES, "staging overlays the gradient image");

#define HX_SLOT(r) ((int)((r) & 127u))
#define HX_LI(r) ((int)(((r) >> 7) & 7u))
#define HX_LJ(r) ((int)(((r) >> 10) & 7u))
#define HX_ACTIVE(r) ((((r) >> 13) & 1u) != 0)
#define HX_FLUSH(r) ((((r) >> 14) & 1u) != 0)
#define HX_NODE(r) ((int)(((r) >> 15) & 63u))
#define HX_RANK(r) ((int)(((r) >> 21) & 127u))
#define HX_TR(r) ((((r) >> 28) & 1u) != 0)

// Voigt row of the strain component (alpha, gamma): [xx, yy, zz, xy, yz, zx] (SOLID.py:218-266)
__device__ __forceinline__ constexpr int voigt(int a, int g) {
    return a == g ? a : (a + g == 1 ? 3 : (a + g == 3 ? 4 : 5));
}

template <bool UNI>
__global__ __launch_bounds__(HX_NT) void fused_hex8_kernel(const double *__restrict__ xyz, const int32_t *__restrict__ prop_id,
                                                          const double *__restrict__ props, int32_t n_props, int64_t n_pat,
                                                          const int64_t *__restrict__ pat_eptr, const int32_t *__restrict__ pat_elems,
                                                          const int64_t *__restrict__ pat_nptr, const fs_hex_node *__restrict__ nodes,
                                                          const int64_t *__restrict__ pat_rptr, const fs_hex_run *__restrict__ runs,
                                                          const int64_t *__restrict__ pat_lptr, const uint32_t *__restrict__ steps,
                                                          double *__restrict__ data, int32_t *__restrict__ status) {
    __shared__ double2 s_mem[HX_PIECES];                  // gradients, later the staging image of the patch's CSR slabs
    double2 *const s_a = s_mem, *const s_b = s_mem + FS_HEX_MAX_PATCH_ELEMS * HX_SA;
    double *const s_g = reinterpret_cast<double *>(s_mem);
    __shared__ int s_pid[UNI ? 1 : FS_HEX_MAX_PATCH_ELEMS];
    __shared__ fs_hex_node s_node[FS_HEX_MAX_PATCH_NODES];
    __shared__ fs_hex_run s_run[FS_HEX_MAX_PATCH_NODES];  // copy-out runs of the patch (at most one per node)
    const int tid = threadIdx.x;
    // persistent workgroups, XCD k (workgroup index & 7 under round-robin dispatch) walks a contiguous eighth of the patches
    const int xcd = blockIdx.x & 7, wpx = (int)(gridDim.x >> 3);
    const int64_t per_xcd = (n_pat + 7) / 8;
    const int64_t pend = ((xcd + 1) * per_xcd < n_pat) ? (xcd + 1) * per_xcd : n_pat;

    // material: one row for the whole model -> uniform (scalar) loads; C symmetrised like K (SOLID.py:565)
    double Cu[6][6];
    if (UNI) {
        for (int a = 0; a < 6; a++)
            for (int b = a; b < 6; b++) { const double v = 0.5 * (props[a * 6 + b] + props[b * 6 + a]); Cu[a][b] = v; Cu[b][a] = v; }
    }
    const double gp = kGP;                                // 1 / sqrt(3)   (SOLID.py:1745-1766: order k-j-i, weights 1)

    // a wave: 8 slots (fastest) x 8 Gauss points; a lane's two (slot, point) pairs of phase 1
    const int q1 = (tid >> 3) & 7;
    const int slotA = (tid >> 6) * 8 + (tid & 7), slotB = slotA + HX_NT / 8;
    int64_t p = xcd * per_xcd + (blockIdx.x >> 3);
    if (p >= pend) return;
    // The element ids of a patch are fetched while the patch before it runs, and one word of every 24 bytes of their
    // coordinates is touched under that patch's phase 2 (a load between arithmetic costs the wave nothing): phase 1 then
    // starts with ONE round trip to this XCD's L2 instead of three dependent ones (header -> ids -> coordinates from HBM).
    // 1 048 576 bricks: 1.44 -> 1.31 ms.
    int ne = (int)(pat_eptr[p + 1] - pat_eptr[p]);
    int32_t eA = pat_elems[pat_eptr[p] + (slotA < ne ? slotA : ne - 1)], eB = pat_elems[pat_eptr[p] + (slotB < ne ? slotB : ne - 1)];
    for (; p < pend; p += wpx) {
        const int64_t n0 = pat_nptr[p], r0 = pat_rptr[p], l0 = pat_lptr[p];
        const int nn = (int)(pat_nptr[p + 1] - n0), nr = (int)(pat_rptr[p + 1] - r0);
        const int nl = (int)(pat_lptr[p + 1] - l0);
        const int64_t pn = (p + wpx < pend) ? p + wpx : p;
        const int64_t e0n = pat_eptr[pn];
        const int nen = (int)(pat_eptr[pn + 1] - e0n);
        const int32_t eAn = pat_elems[e0n + (slotA < nen ? slotA : nen - 1)], eBn = pat_elems[e0n + (slotB < nen ? slotB : nen - 1)];
        // this lane's eight steps (used after phase 1: the loads travel under it)
        uint4 sa = make_uint4(0, 0, 0, 0), sb = sa;
        if (tid < nl) {
            const uint4 *sp = reinterpret_cast<const uint4 *>(steps + (l0 + tid) * 8);
            sa = sp[0]; sb = sp[1];
        }
        if (tid < nn) s_node[tid] = nodes[n0 + tid];
        if (tid >= 64 && tid - 64 < nr) s_run[tid - 64] = runs[r0 + tid - 64];     // (read after three barriers)
        // ---------------- phase 1: gradients of (slot, Gauss point) ----------------
        // at most FS_HEX_MAX_PATCH_ELEMS x 8 = 800 (slot, point) pairs: two passes of the workgroup, written out so that
        // the element indices and coordinates of BOTH passes are requested before the first one computes (a loop exposes
        // two dependent global round trips per pass)
        static_assert(2 * HX_NT >= ((FS_HEX_MAX_PATCH_ELEMS + 7) & ~7) * 8, "two passes cover a patch");
        auto gradients = [&](int slot, int q, int64_t e) {
#pragma clang fp contract(fast)
            const double xi = (q & 1) ? gp : -gp, eta = (q & 2) ? gp : -gp, zeta = (q & 4) ? gp : -gp;
            double N[8], dxi[8], deta[8], dzeta[8];
            solid_shape<8>(xi, eta, zeta, N, dxi, deta, dzeta);
            double J[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
            const double *x = xyz + e * 24;
#pragma unroll
            for (int i = 0; i < 8; i++)
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const double xc = x[3 * i + c];
                    J[0][c] += dxi[i] * xc; J[1][c] += deta[i] * xc; J[2][c] += dzeta[i] * xc;
                }
            const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1], c01 = J[1][0] * J[2][2] - J[1][2] * J[2][0],
                         c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
            const double det = J[0][0] * c00 - J[0][1] * c01 + J[0][2] * c02;
            if (det <= 1e-14 && status) atomicOr(status + e, FS_ST_BAD_JAC);      // ValueError in SOLID.py:211-212
            double inv[3][3];
            {
                const double rdet = recip(det);
                inv[0][0] = divr(c00, det, rdet); inv[0][1] = divr(J[0][2] * J[2][1] - J[0][1] * J[2][2], det, rdet);
                inv[0][2] = divr(J[0][1] * J[1][2] - J[0][2] * J[1][1], det, rdet); inv[1][0] = divr(-c01, det, rdet);
                inv[1][1] = divr(J[0][0] * J[2][2] - J[0][2] * J[2][0], det, rdet);
                inv[1][2] = divr(J[0][2] * J[1][0] - J[0][0] * J[1][2], det, rdet);
                inv[2][0] = divr(c02, det, rdet); inv[2][1] = divr(J[0][1] * J[2][0] - J[0][0] * J[2][1], det, rdet);
                inv[2][2] = divr(J[0][0] * J[1][1] - J[0][1] * J[1][0], det, rdet);
            }
            const double sw = fsqrt(det);                   // weights are 1: w detJ = detJ
            double2 *ga = s_a + slot * HX_SA + q * 8;
            double *gb = reinterpret_cast<double *>(s_b + slot * HX_SB + (q >> 1) * 8) + (q & 1);
#pragma unroll
            for (int i = 0; i < 8; i++) {
                double gr[3];
#pragma unroll
                for (int a = 0; a < 3; a++) gr[a] = sw * (inv[a][0] * dxi[i] + inv[a][1] * deta[i] + inv[a][2] * dzeta[i]);
                ga[i] = make_double2(gr[0], gr[1]);
                gb[2 * i] = gr[2];
            }
            if (!UNI && q == 0) s_pid[slot] = safe_pid(prop_id[e], n_props);
        };
        if (slotA < ne) gradients(slotA, q1, eA);
        if (slotB < ne) gradients(slotB, q1, eB);
        __syncthreads();
        // one word of every 24 bytes of the next patch's coordinates: the lines travel to this XCD's L2 while phase 2 runs
        // (the values are dropped behind phase 2)
        const int32_t *xw = reinterpret_cast<const int32_t *>(xyz);
        const int32_t tA = xw[(int64_t)eAn * 48 + q1 * 6], tB = xw[(int64_t)eBn * 48 + q1 * 6];

        // ---------------- phase 2: eight steps per lane ----------------
        double Kout[4][9];       // finished blocks: the plan lets blocks end at odd steps only
        {
#pragma clang fp contract(fast)
            const uint32_t rec[8] = {sa.x, sa.y, sa.z, sa.w, sb.x, sb.y, sb.z, sb.w};
            // UNI: the Gauss-point sums G of the block's contributions, added in element order (K = C : sum of G: one
            // contraction per block); else the contributions K_e = C_e : G_e themselves, added in element order
            double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            // K[al][be] = sum over (ga, de) of C[voigt(al, ga)][voigt(be, de)] G[ga][de], then oriented: the plan names
            // the local nodes of a contribution in the order (smaller, larger) of their GLOBAL indices, so blocks (I, J)
            // and (J, I) form the same numbers and one of them (HX_TR) stores the transpose; diagonal blocks are
            // symmetrised (SOLID.py:565)
            auto contract = [&](const double (&C)[6][6], const double *G, bool tr, bool dg, double *K) {
#pragma unroll
                for (int al = 0; al < 3; al++)
#pragma unroll
                    for (int be = 0; be < 3; be++) {
                        double v = C[voigt(al, 0)][voigt(be, 0)] * G[0];
#pragma unroll
                        for (int t = 1; t < 9; t++) v = FMA(C[voigt(al, t / 3)][voigt(be, t % 3)], G[t], v);
                        K[3 * al + be] = v;
                    }
#pragma unroll
                for (int al = 0; al < 3; al++)
#pragma unroll
                    for (int be = al + 1; be < 3; be++) {
                        const double u = K[3 * al + be], l = K[3 * be + al], mean = 0.5 * (u + l);
                        K[3 * al + be] = dg ? mean : (tr ? l : u);
                        K[3 * be + al] = dg ? mean : (tr ? u : l);
                    }
            };
            bool dg = false;
#pragma unroll
            for (int s = 0; s < 8; s++) {
                const uint32_t r = rec[s];
                if (HX_ACTIVE(r)) {
                    const int ci = HX_LI(r), cj = HX_LJ(r);
                    dg = ci == cj;
                    const double2 *pa = s_a + HX_SLOT(r) * HX_SA, *pb = s_b + HX_SLOT(r) * HX_SB;
                    double G[9];
#pragma unroll
                    for (int c = 0; c < 9; c++) G[c] = UNI ? acc[c] : 0.0;
#pragma unroll
                    for (int qp = 0; qp < 4; qp++) {
                        const double2 az = pb[qp * 8 + ci], bz = pb[qp * 8 + cj];
#pragma unroll
                        for (int h = 0; h < 2; h++) {
                            const double2 axy = pa[(2 * qp + h) * 8 + ci], bxy = pa[(2 * qp + h) * 8 + cj];
                            const double a0 = axy.x, a1 = axy.y, a2 = h ? az.y : az.x;
                            const double b0 = bxy.x, b1 = bxy.y, b2 = h ? bz.y : bz.x;
                            G[0] = FMA(a0, b0, G[0]); G[1] = FMA(a0, b1, G[1]); G[2] = FMA(a0, b2, G[2]);
                            G[3] = FMA(a1, b0, G[3]); G[4] = FMA(a1, b1, G[4]); G[5] = FMA(a1, b2, G[5]);
                            G[6] = FMA(a2, b0, G[6]); G[7] = FMA(a2, b1, G[7]); G[8] = FMA(a2, b2, G[8]);
                        }
                    }
                    if (UNI) {
#pragma unroll
                        for (int c = 0; c < 9; c++) acc[c] = G[c];
                    } else {
                        const double *pr = props + 37 * (int64_t)s_pid[HX_SLOT(r)];
                        double C[6][6], K[9];
                        for (int a = 0; a < 6; a++)
                            for (int b = a; b < 6; b++) { const double v = 0.5 * (pr[a * 6 + b] + pr[b * 6 + a]); C[a][b] = v; C[b][a] = v; }
                        contract(C, G, HX_TR(r), dg, K);
#pragma unroll
                        for (int c = 0; c < 9; c++) acc[c] = acc[c] + K[c];
                    }
                }
                if ((s & 1) && HX_FLUSH(r)) {
                    if (UNI) {
                        contract(Cu, acc, HX_TR(r), dg, Kout[s >> 1]);      // a pattern-only block: C : 0 = 0
#pragma unroll
                        for (int c = 0; c < 9; c++) acc[c] = 0.0;
                    } else {
#pragma unroll
                        for (int c = 0; c < 9; c++) { Kout[s >> 1][c] = acc[c]; acc[c] = 0.0; }
                    }
                }
            }
        }
        asm volatile("" :: "v"(tA), "v"(tB));
        __syncthreads();        // every wave is done with the gradients: the staging image overlays them
        {
            const uint32_t rec[8] = {sa.x, sa.y, sa.z, sa.w, sb.x, sb.y, sb.z, sb.w};
#pragma unroll
            for (int s = 1; s < 8; s += 2) {
                const uint32_t r = rec[s];
                if (HX_FLUSH(r)) {
                    const fs_hex_node nd = s_node[HX_NODE(r)];
                    double *o = s_g + nd.sbase + 3 * HX_RANK(r);
                    const int stride = 3 * nd.m;
#pragma unroll
                    for (int k = 0; k < 3; k++)
#pragma unroll
                        for (int l = 0; l < 3; l++) o[k * stride + l] = Kout[s >> 1][3 * k + l];
                }
            }
        }
        __syncthreads();
        for (int rr = 0; rr < nr; rr++) {
            // 16-byte stores on 16-byte aligned addresses: an odd first index goes out alone, and so does an odd tail
            const fs_hex_run run = s_run[rr];
            const int head = (int)(run.gdst & 1), np2 = (run.len - head) >> 1;
            double *dst = data + run.gdst + head;
            const double *src = s_g + run.sbase + head;
            for (int c = tid; c < np2; c += HX_NT)
                *reinterpret_cast<double2 *>(dst + 2 * c) = make_double2(src[2 * c], src[2 * c + 1]);
            if (tid == 0 && head) dst[-1] = src[-1];
            if (tid == 64 && ((run.len - head) & 1)) dst[2 * np2] = src[2 * np2];
        }
        __syncthreads();        // the next patch's phase 1 overwrites the image
        ne = nen; eA = eAn; eB = eBn;
    }
}

// Consistent mass matrix of HEXA8 (SolidElement.M, elements/SOLID.py:569-597: rho * sum_q N^T N detJ w) through the same
// plan and the same skeleton: phase 1 keeps sqrt(rho detJ) N_i(q) per (slot, point) -- 65 doubles per slot --, a step is
// m = sum_q n_i n_j (8 multiply-adds), the contributions of a block are added in element order and the block is m I_3.
// Products commute and blocks (I, J), (J, I) add the same numbers in the same order: exactly symmetric.
constexpr int HX_MS = 65;
__global__ __launch_bounds__(HX_NT) void fused_hex8_mass_kernel(const double *__restrict__ xyz, const int32_t *__restrict__ prop_id,
                                                               const double *__restrict__ props, int32_t n_props, int64_t n_pat,
                                                               const int64_t *__restrict__ pat_eptr, const int32_t *__restrict__ pat_elems,
                                                               const int64_t *__restrict__ pat_nptr, const fs_hex_node *__restrict__ nodes,
                                                               const int64_t *__restrict__ pat_rptr, const fs_hex_run *__restrict__ runs,
                                                               const int64_t *__restrict__ pat_lptr, const uint32_t *__restrict__ steps,
                                                               double *__restrict__ data) {
    __shared__ double s_g[FS_HEX_STAGE_DOUBLES];          // shape values, later the staging image of the patch's CSR slabs
    __shared__ fs_hex_node s_node[FS_HEX_MAX_PATCH_NODES];
    __shared__ fs_hex_run s_run[FS_HEX_MAX_PATCH_NODES];
    static_assert(FS_HEX_MAX_PATCH_ELEMS * HX_MS <= FS_HEX_STAGE_DOUBLES, "the shape values fit the staging image");
    const int tid = threadIdx.x;
    const int xcd = blockIdx.x & 7, wpx = (int)(gridDim.x >> 3);
    const int64_t per_xcd = (n_pat + 7) / 8;
    const int64_t pend = ((xcd + 1) * per_xcd < n_pat) ? (xcd + 1) * per_xcd : n_pat;
    const double gp = kGP;
    for (int64_t p = xcd * per_xcd + (blockIdx.x >> 3); p < pend; p += wpx) {
        const int64_t e0 = pat_eptr[p], n0 = pat_nptr[p], r0 = pat_rptr[p], l0 = pat_lptr[p];
        const int ne = (int)(pat_eptr[p + 1] - e0), nn = (int)(pat_nptr[p + 1] - n0), nr = (int)(pat_rptr[p + 1] - r0);
        const int nl = (int)(pat_lptr[p + 1] - l0);
        uint4 sa = make_uint4(0, 0, 0, 0), sb = sa;
        if (tid < nl) {
            const uint4 *sp = reinterpret_cast<const uint4 *>(steps + (l0 + tid) * 8);
            sa = sp[0]; sb = sp[1];
        }
        if (tid < nn) s_node[tid] = nodes[n0 + tid];
        if (tid >= 64 && tid - 64 < nr) s_run[tid - 64] = runs[r0 + tid - 64];
        // the next patch's elements of this lane: their coordinates and property ids are touched under the steps below and
        // wait in this XCD's L2 when its phase 1 asks for them (see fused_hex8_kernel)
        const int64_t pn = (p + wpx < pend) ? p + wpx : p;
        const int64_t e0n = pat_eptr[pn];
        const int nen = (int)(pat_eptr[pn + 1] - e0n);
        const int sA = tid >> 3, sB = sA + HX_NT / 8;
        const int32_t eAn = pat_elems[e0n + (sA < nen ? sA : nen - 1)], eBn = pat_elems[e0n + (sB < nen ? sB : nen - 1)];
        for (int t = tid; t < ne * 8; t += HX_NT) {
#pragma clang fp contract(fast)
            const int slot = t >> 3, q = t & 7;
            const int64_t e = pat_elems[e0 + slot];
            const double xi = (q & 1) ? gp : -gp, eta = (q & 2) ? gp : -gp, zeta = (q & 4) ? gp : -gp;
            double N[8], dxi[8], deta[8], dzeta[8];
            solid_shape<8>(xi, eta, zeta, N, dxi, deta, dzeta);
            double J[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
            const double *x = xyz + e * 24;
#pragma unroll
            for (int i = 0; i < 8; i++)
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const double xc = x[3 * i + c];
                    J[0][c] += dxi[i] * xc; J[1][c] += deta[i] * xc; J[2][c] += dzeta[i] * xc;
                }
            const double det = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) - J[0][1] * (J[1][0] * J[2][2] - J[1][2] * J[2][0]) +
                               J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
            const double rho = props[(int64_t)safe_pid(prop_id[e], n_props) * 37 + 36];
            const double sw = fsqrt(rho * det);                   // weights are 1
            double *g = s_g + slot * HX_MS + q * 8;
#pragma unroll
            for (int i = 0; i < 8; i++) g[i] = sw * N[i];
        }
        __syncthreads();
        const int32_t *xw = reinterpret_cast<const int32_t *>(xyz);
        const int32_t tA = xw[(int64_t)eAn * 48 + (tid & 7) * 6], tB = xw[(int64_t)eBn * 48 + (tid & 7) * 6];
        const int32_t tC = prop_id[(tid & 1) ? eBn : eAn];
        double Mout[4];
        {
#pragma clang fp contract(fast)
            const uint32_t rec[8] = {sa.x, sa.y, sa.z, sa.w, sb.x, sb.y, sb.z, sb.w};
            double acc = 0.0;
#pragma unroll
            for (int s = 0; s < 8; s++) {
                const uint32_t r = rec[s];
                if (HX_ACTIVE(r)) {
                    const double *ga = s_g + HX_SLOT(r) * HX_MS + HX_LI(r), *gb = s_g + HX_SLOT(r) * HX_MS + HX_LJ(r);
                    double m = 0.0;
#pragma unroll
                    for (int q = 0; q < 8; q++) m = FMA(ga[q * 8], gb[q * 8], m);
                    acc = acc + m;
                }
                if ((s & 1) && HX_FLUSH(r)) { Mout[s >> 1] = acc; acc = 0.0; }
            }
        }
        asm volatile("" :: "v"(tA), "v"(tB), "v"(tC));
        __syncthreads();
        {
            const uint32_t rec[8] = {sa.x, sa.y, sa.z, sa.w, sb.x, sb.y, sb.z, sb.w};
#pragma unroll
            for (int s = 1; s < 8; s += 2) {
                const uint32_t r = rec[s];
                if (HX_FLUSH(r)) {
                    const fs_hex_node nd = s_node[HX_NODE(r)];
                    double *o = s_g + nd.sbase + 3 * HX_RANK(r);
                    const int stride = 3 * nd.m;
#pragma unroll
                    for (int k = 0; k < 3; k++)
#pragma unroll
                        for (int l = 0; l < 3; l++) o[k * stride + l] = k == l ? Mout[s >> 1] : 0.0;
                }
            }
        }
        __syncthreads();
        for (int rr = 0; rr < nr; rr++) {
            const fs_hex_run run = s_run[rr];
            const int head = (int)(run.gdst & 1), np2 = (run.len - head) >> 1;
            double *dst = data + run.gdst + head;
            const double *src = s_g + run.sbase + head;
            for (int c = tid; c < np2; c += HX_NT)
                *reinterpret_cast<double2 *>(dst + 2 * c) = make_double2(src[2 * c], src[2 * c + 1]);
            if (tid == 0 && head) dst[-1] = src[-1];
            if (tid == 64 && ((run.len - head) & 1)) dst[2 * np2] = src[2 * np2];
        }
        __syncthreads();
    }
}

// Invariants fused_hex8_kernel relies on when it indexes LDS and `data` with values read from the plan
// (fs_fused_hex_plan_check): workgroup per patch (grid-stride).  err[0] = violations, err[1] = first rule, err[2] = patch.
__global__ __launch_bounds__(256) void hex_plan_check_kernel(int64_t n_pat, const int64_t *__restrict__ pat_eptr,
                                                            const int32_t *__restrict__ pat_elems, const int64_t *__restrict__ pat_nptr,
                                                            const fs_hex_node *__restrict__ nodes, const int64_t *__restrict__ pat_rptr,
                                                            const fs_hex_run *__restrict__ runs, const int64_t *__restrict__ pat_lptr,
                                                            const uint32_t *__restrict__ steps, int64_t E, int64_t nnz,
                                                            unsigned long long *__restrict__ err) {
    auto report = [&](int rule, int64_t p) {
        if (atomicAdd(&err[0], 1ull) == 0) { err[1] = (unsigned long long)rule; err[2] = (unsigned long long)p; }
    };
    for (int64_t p = blockIdx.x; p < n_pat; p += gridDim.x) {
        const int64_t e0 = pat_eptr[p], n0 = pat_nptr[p], r0 = pat_rptr[p], l0 = pat_lptr[p];
        const int64_t ne = pat_eptr[p + 1] - e0, nn = pat_nptr[p + 1] - n0, nr = pat_rptr[p + 1] - r0, nl = pat_lptr[p + 1] - l0;
        const bool hdr = ne >= 1 && ne <= FS_HEX_MAX_PATCH_ELEMS && nn >= 1 && nn <= FS_HEX_MAX_PATCH_NODES && nr >= 0 &&
                         nr <= FS_HEX_MAX_PATCH_NODES && nl >= 0 && nl <= FS_HEX_MAX_PATCH_LANES && (nl & 63) == 0 &&
                         e0 >= 0 && n0 >= 0 && r0 >= 0 && l0 >= 0;
        if (!hdr) { if (threadIdx.x == 0) report(1, p); continue; }
        for (int64_t q = threadIdx.x; q < ne; q += blockDim.x)
            if (pat_elems[e0 + q] < 0 || pat_elems[e0 + q] >= E) report(2, p);
        for (int64_t q = threadIdx.x; q < nn; q += blockDim.x) {
            const fs_hex_node nd = nodes[n0 + q];
            if (nd.sbase < 0 || nd.m < 1 || (int64_t)nd.sbase + 9 * (int64_t)nd.m > FS_HEX_STAGE_DOUBLES) report(3, p);
        }
        for (int64_t q = threadIdx.x; q < nr; q += blockDim.x) {
            const fs_hex_run rn = runs[r0 + q];
            if (rn.sbase < 0 || rn.len < 0 || (int64_t)rn.sbase + rn.len > FS_HEX_STAGE_DOUBLES || rn.gdst < 0 || rn.gdst + rn.len > nnz)
                report(4, p);
        }
        for (int64_t q = threadIdx.x; q < nl * 8; q += blockDim.x) {
            const uint32_t r = steps[l0 * 8 + q];
            if (HX_ACTIVE(r) && HX_SLOT(r) >= ne) report(5, p);
            if (HX_FLUSH(r)) {
                if (!(q & 1) || HX_NODE(r) >= nn) report(6, p);
                else if (HX_RANK(r) >= nodes[n0 + HX_NODE(r)].m) report(6, p);
            }
        }
    }
}

}  // namespace

extern "C" int fs_fused_assemble_hex8(const double *xyz, const int32_t *prop_id, const double *props, int32_t n_props,
                                      int64_t E, int64_t n_patches, const int64_t *pat_eptr, const int32_t *pat_elems,
                                      const int64_t *pat_nptr, const fs_hex_node *nodes, const int64_t *pat_rptr,
                                      const fs_hex_run *runs, const int64_t *pat_lptr, const uint32_t *steps, double *data,
                                      int32_t *status, void *stream) {
    if (E < 0 || n_props <= 0 || n_patches < 0) return fs_fail_(FS_EINVAL, "fs_fused_assemble_hex8: bad sizes", 0);
    if (E == 0 || n_patches == 0) return FS_OK;
    if (!xyz || !prop_id || !props || !pat_eptr || !pat_elems || !pat_nptr || !nodes || !pat_rptr || !runs || !pat_lptr ||
        !steps || !data)
        return fs_fail_(FS_EINVAL, "fs_fused_assemble_hex8: null argument", 0);
    int cus = 0;
    if (int rc = fs_device_cu_count_(&cus)) return rc;
    int64_t wgs = ((int64_t)cus / 8) * 8;
    if (wgs < 8) wgs = 8;
    if (wgs > ((n_patches + 7) / 8) * 8) wgs = ((n_patches + 7) / 8) * 8;
    hipStream_t s = (hipStream_t)stream;
    if (n_props == 1)
        hipLaunchKernelGGL(fused_hex8_kernel<true>, dim3((unsigned)wgs), dim3(HX_NT), 0, s, xyz, prop_id, props, n_props,
                           n_patches, pat_eptr, pat_elems, pat_nptr, nodes, pat_rptr, runs, pat_lptr, steps, data, status);
    else
        hipLaunchKernelGGL(fused_hex8_kernel<false>, dim3((unsigned)wgs), dim3(HX_NT), 0, s, xyz, prop_id, props, n_props,
                           n_patches, pat_eptr, pat_elems, pat_nptr, nodes, pat_rptr, runs, pat_lptr, steps, data, status);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fs_fail_(FS_EHIP, "fused_hex8_kernel", (int)e);
    return FS_OK;
}

extern "C" int fs_fused_assemble_hex8_mass(const double *xyz, const int32_t *prop_id, const double *props, int32_t n_props,
                                           int64_t E, int64_t n_patches, const int64_t *pat_eptr, const int32_t *pat_elems,
                                           const int64_t *pat_nptr, const fs_hex_node *nodes, const int64_t *pat_rptr,
                                           const fs_hex_run *runs, const int64_t *pat_lptr, const uint32_t *steps, double *data,
                                           void *stream) {
    if (E < 0 || n_props <= 0 || n_patches < 0) return fs_fail_(FS_EINVAL, "fs_fused_assemble_hex8_mass: bad sizes", 0);
    if (E == 0 || n_patches == 0) return FS_OK;
    if (!xyz || !prop_id || !props || !pat_eptr || !pat_elems || !pat_nptr || !nodes || !pat_rptr || !runs || !pat_lptr ||
        !steps || !data)
        return fs_fail_(FS_EINVAL, "fs_fused_assemble_hex8_mass: null argument", 0);
    int cus = 0;
    if (int rc = fs_device_cu_count_(&cus)) return rc;
    int64_t wgs = ((int64_t)cus / 8) * 8;
    if (wgs < 8) wgs = 8;
    if (wgs > ((n_patches + 7) / 8) * 8) wgs = ((n_patches + 7) / 8) * 8;
    hipLaunchKernelGGL(fused_hex8_mass_kernel, dim3((unsigned)wgs), dim3(HX_NT), 0, (hipStream_t)stream, xyz, prop_id, props,
                       n_props, n_patches, pat_eptr, pat_elems, pat_nptr, nodes, pat_rptr, runs, pat_lptr, steps, data);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fs_fail_(FS_EHIP, "fused_hex8_mass_kernel", (int)e);
    return FS_OK;
}

extern "C" int fs_fused_hex_plan_check(int64_t n_patches, const int64_t *pat_eptr, const int32_t *pat_elems,
                                       const int64_t *pat_nptr, const fs_hex_node *nodes, const int64_t *pat_rptr,
                                       const fs_hex_run *runs, const int64_t *pat_lptr, const uint32_t *steps, int64_t E,
                                       int64_t nnz, void *stream) {
    if (E < 0 || n_patches < 0 || nnz < 0) return fs_fail_(FS_EINVAL, "fs_fused_hex_plan_check: bad sizes", 0);
    if (n_patches == 0) return FS_OK;
    if (!pat_eptr || !pat_elems || !pat_nptr || !nodes || !pat_rptr || !runs || !pat_lptr || !steps)
        return fs_fail_(FS_EINVAL, "fs_fused_hex_plan_check: null argument", 0);
    hipStream_t s = (hipStream_t)stream;
    unsigned long long *err = nullptr, host[3] = {0, 0, 0};
    hipError_t e = hipMalloc(&err, sizeof host);
    if (e != hipSuccess) return fs_fail_(FS_EHIP, "fs_fused_hex_plan_check: hipMalloc", (int)e);
    e = hipMemsetAsync(err, 0, sizeof host, s);
    if (e == hipSuccess) {
        const unsigned grid = (unsigned)(n_patches < 8192 ? n_patches : 8192);
        hipLaunchKernelGGL(hex_plan_check_kernel, dim3(grid), dim3(256), 0, s, n_patches, pat_eptr, pat_elems, pat_nptr, nodes,
                           pat_rptr, runs, pat_lptr, steps, E, nnz, err);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(host, err, sizeof host, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(err);
    if (e != hipSuccess) return fs_fail_(FS_EHIP, "fs_fused_hex_plan_check", (int)e);
    if (host[0]) {
        static const char *const rules[] = {"", "patch header out of range (elements, nodes, runs, lanes)", "element index outside [0, E)",
            "node slab outside the staging image", "copy-out run outside the staging image or the value array",
            "step names a slot beyond its patch", "flush on an even step, or node / neighbour rank out of range"};
        char msg[200];
        snprintf(msg, sizeof msg, "fs_fused_hex_plan_check: %llu violations; first: %s (patch %lld)", host[0],
                 rules[host[1] < 7 ? host[1] : 0], (long long)host[2]);
        return fs_fail_(FS_EINVAL, msg, 0);
    }
    return FS_OK;
}
