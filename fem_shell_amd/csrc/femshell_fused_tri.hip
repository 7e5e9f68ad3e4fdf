// femshell_fused_tri.hip -- fused evaluate-and-assemble for MITC3+ shells (fs_fused_assemble_tri, include/femshell_hip.h).
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (csrc/Makefile).  Citations: /root/reference/src/fem_shell/.
//
// One launch replaces MITC3.K / MITC3Composite.K of every element (elements/MITC3.py:935-1103, MITC3_composite.py:204-240)
// and the ADD_VALUES loop over the element matrices (core/assembler.py:287-303).  Gather form, one workgroup (256 lanes,
// persistent, XCD-contiguous patch ranges) per PATCH of nodes, in the step layout of the fused HEXA8 kernel:
//   phase 1  lane = (element slot, function f = node 0..2 or bubble): triangle frame, area and the shape / tying data
//            of f at the 3 Hammer points (tri_geometry, tri_fn: the device functions of mitc3_k_lds_kernel) into an LDS
//            image fn[23][4][slots + 8], geo[10][slots]; elements of the one-element halo are evaluated by every patch that
//            touches them.  Three passes of the workgroup from registers: element indices, coordinates and property
//            ids travel one patch ahead, under phase 2 of the previous patch.
//   phase 2  the patch forms the COMPLETE 6-row CSR slabs of its nodes (7 blocks for an interior node of a structured
//            triangulation).  A lane works through eight steps, one contribution (element slot, local node pair) each:
//            the 6 x 6 block of the ordered local pair (tri_pair_block, shared with mitc3_k_lds_kernel: the same
//            bits), transposed where the block wants (larger, smaller).  The contributions of a CSR block follow each
//            other on ONE lane and are added in element order, 0 + x0 + x1 + ... (the reference's insertion order, the
//            sums of fs_csr_numeric_desc); the finished block leaves from registers as 16-byte stores (a 36-double
//            block cannot wait in registers for a staged copy-out the way the 9-double HEXA8 blocks do; the blocks of a
//            node are neighbours in memory and written by neighbouring lanes).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "femshell_hip.h"
#include "femshell_internal.h"
#include "femshell_geom.h"
#include "femshell_tri.h"

namespace {

constexpr int TR_S = FS_TRI_MAX_PATCH_ELEMS;     // element slots of a patch
constexpr int TR_SS = TR_S + 8;                  // entries per (component, function): 184 = 24 mod 32 -- the four functions
                                                 // of a slot start 24, 16, 8 eight-byte banks apart, neighbouring slots 1
constexpr int TR_FS = 4 * TR_SS;                 // entries per component of the function image
constexpr int TR_NT = FS_TRI_MAX_PATCH_LANES;    // 256

#define TR_SLOT(r) ((int)((r) & 255u))
#define TR_LI(r) ((int)(((r) >> 8) & 3u))
#define TR_LJ(r) ((int)(((r) >> 10) & 3u))
#define TR_ACTIVE(r) ((((r) >> 13) & 1u) != 0)
#define TR_FLUSH(r) ((((r) >> 14) & 1u) != 0)
#define TR_NODE(r) ((int)(((r) >> 15) & 127u))
#define TR_RANK(r) ((int)(((r) >> 22) & 127u))

// the members of a property row that tri_pair_block reads
struct TriProp { double A[9], B[9], D[9], Cs[4], k_drill; int32_t flags; };
__device__ __forceinline__ void load_tri_prop(const fs_shell_prop *src, TriProp &o) {
#pragma unroll
    for (int c = 0; c < 9; c++) { o.A[c] = src->A[c]; o.B[c] = src->B[c]; o.D[c] = src->D[c]; }
#pragma unroll
    for (int c = 0; c < 4; c++) o.Cs[c] = src->Cs[c];
    o.k_drill = src->k_drill;
    o.flags = src->flags;
}

// One contribution of the consistent mass matrix: block (ti <= tj) of the element in image slot `slot`, the arithmetic of
// mitc3_m_kernel statement for statement (MITC3.py:1105-1138)
__device__ __forceinline__ void tri_mass_block(const double *s_geo, int slot, int ti, int tj, const fs_shell_prop *pr,
                                               double (&G)[6][6]) {
    const double area = s_geo[0 * TR_S + slot];
    double R[3][3];
    for (int k = 0; k < 3; k++) {
        R[0][k] = s_geo[(1 + k) * TR_S + slot]; R[1][k] = s_geo[(4 + k) * TR_S + slot]; R[2][k] = s_geo[(7 + k) * TR_S + slot];
    }
    const double mt = pr->m_t, mr = pr->m_r;
    double ct = 0.0, cr = 0.0;
    for (int g = 0; g < 3; g++) {
        const double r = kTriR[g], s = kTriS[g];
        const double hi = (ti == 0) ? (1.0 - r - s) : ((ti == 1) ? r : s);
        const double hj = (tj == 0) ? (1.0 - r - s) : ((tj == 1) ? r : s);
        ct = ct + hi * hj * mt * (1.0 / 3.0) * area;
        cr = cr + hi * hj * mr * (1.0 / 3.0) * area;
    }
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
}

// UNI: the model has ONE property row (every lane reads the same numbers: scalar registers, loaded once).
// MASS: the same plan, steps and sums form the consistent MASS matrix (image: area and frame only; no function image, so
// several workgroups share a CU)
template <bool UNI, bool MASS = false>
__global__ __launch_bounds__(TR_NT) void fused_tri_kernel(const double *__restrict__ xyz, const int32_t *__restrict__ prop_id,
                                                         const fs_shell_prop *__restrict__ props, int32_t n_props, int64_t n_pat,
                                                         const int64_t *__restrict__ pat_eptr, const int32_t *__restrict__ pat_elems,
                                                         const int64_t *__restrict__ pat_nptr, const fs_tri_node *__restrict__ nodes,
                                                         const int64_t *__restrict__ pat_lptr, const uint32_t *__restrict__ steps,
                                                         double *__restrict__ data, int32_t *__restrict__ status) {
    __shared__ double s_fn[MASS ? 1 : 23 * TR_FS];        // shape / tying data [component][f][slot]
    __shared__ double s_geo[10 * TR_S];                   // area, frame rows [component][slot]
    __shared__ int32_t s_pid[TR_S];
    __shared__ fs_tri_node s_node[FS_TRI_MAX_PATCH_NODES];
    const int tid = threadIdx.x;
    // persistent workgroups, XCD k (workgroup index & 7 under round-robin dispatch) walks a contiguous eighth of the patches
    const int xcd = blockIdx.x & 7, wpx = (int)(gridDim.x >> 3);
    const int64_t per_xcd = (n_pat + 7) / 8;
    const int64_t pend = ((xcd + 1) * per_xcd < n_pat) ? (xcd + 1) * per_xcd : n_pat;
    // The inputs of phase 1 travel ONE PATCH AHEAD: the element indices of the next patch are requested at the top of the
    // current one, its coordinates and property ids after the phase-1 barrier, so that they arrive under phase 2 (one wave
    // per SIMD: nothing else hides a global round trip, and a plain loop exposes six of them per patch).
    const int64_t pfirst = xcd * per_xcd + (blockIdx.x >> 3);
    TriProp Pu;
    if (UNI && !MASS) load_tri_prop(props, Pu);
    static_assert(3 * TR_NT >= 4 * TR_S, "three passes of the workgroup cover the (slot, function) pairs of a patch");
    int32_t el[3] = {0, 0, 0}, pidn[3] = {0, 0, 0};
    double xs[3][9];
    auto fetch_ids = [&](int64_t pp) {
        const int64_t e1 = pat_eptr[pp];
        const int ne1 = (int)(pat_eptr[pp + 1] - e1);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int slot = (tid + k * TR_NT) >> 2;
            el[k] = pat_elems[e1 + (slot < ne1 ? slot : ne1 - 1)];
        }
    };
    auto fetch_inputs = [&]() {
#pragma unroll
        for (int k = 0; k < 3; k++) {
#pragma unroll
            for (int c = 0; c < 9; c++) xs[k][c] = xyz[(int64_t)el[k] * 9 + c];
            pidn[k] = prop_id[el[k]];
        }
    };
    if (pfirst < pend) { fetch_ids(pfirst); fetch_inputs(); }
    for (int64_t p = pfirst; p < pend; p += wpx) {
        const int64_t e0 = pat_eptr[p], n0 = pat_nptr[p], l0 = pat_lptr[p];
        const int ne = (int)(pat_eptr[p + 1] - e0), nn = (int)(pat_nptr[p + 1] - n0), nl = (int)(pat_lptr[p + 1] - l0);
        (void)e0;
        // this lane's eight steps (used after phase 1: the loads travel under it)
        uint4 sa = make_uint4(0, 0, 0, 0), sb = sa;
        if (tid < nl) {
            const uint4 *sp = reinterpret_cast<const uint4 *>(steps + (l0 + tid) * 8);
            sa = sp[0]; sb = sp[1];
        }
        if (tid < nn) s_node[tid] = nodes[n0 + tid];
        // ---------------- phase 1: (slot, function), three passes from registers ----------------
        int32_t ecur[3];
#pragma unroll
        for (int k = 0; k < 3; k++) ecur[k] = el[k];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int t = tid + k * TR_NT, slot = t >> 2, f = t & 3;
            if (slot >= ne) continue;
            double x[3][3];
            for (int c = 0; c < 9; c++) (&x[0][0])[c] = xs[k][c];
            if (MASS && f != 0) continue;
            TriGeom geo;
            tri_geometry(x, geo);
            if constexpr (!MASS) {
            TriFn F;
            tri_fn(geo, f, F);
            const int w = f * TR_SS + slot;
            for (int g = 0; g < 3; g++) {
                s_fn[(0 + g) * TR_FS + w] = F.fx[g];
                s_fn[(3 + g) * TR_FS + w] = F.fy[g];
                s_fn[(6 + 2 * g) * TR_FS + w] = F.gx[g][0];
                s_fn[(7 + 2 * g) * TR_FS + w] = F.gx[g][1];
                s_fn[(12 + 2 * g) * TR_FS + w] = F.gy[g][0];
                s_fn[(13 + 2 * g) * TR_FS + w] = F.gy[g][1];
                s_fn[(18 + g) * TR_FS + w] = F.h[g];
            }
            s_fn[21 * TR_FS + w] = F.dhx;
            s_fn[22 * TR_FS + w] = F.dhy;
            }
            if (f == 0) {
                s_geo[0 * TR_S + slot] = geo.area;
                for (int k2 = 0; k2 < 3; k2++) {
                    s_geo[(1 + k2) * TR_S + slot] = geo.e1[k2];
                    s_geo[(4 + k2) * TR_S + slot] = geo.e2[k2];
                    s_geo[(7 + k2) * TR_S + slot] = geo.e3[k2];
                }
                const int32_t pid = pidn[k];
                s_pid[slot] = safe_pid(pid, n_props);
                // every patch that touches the element writes the same word
                if (!MASS && status) status[ecur[k]] = geo.status | ((pid < 0 || pid >= n_props) ? FS_ST_BAD_PROP : 0);
            }
        }
        if (p + wpx < pend) fetch_ids(p + wpx);
        __syncthreads();
        if (p + wpx < pend) fetch_inputs();                  // arrive under phase 2
        // ---------------- phase 2: eight steps per lane ----------------
        double acc[6][6];
#pragma unroll
        for (int a = 0; a < 36; a++) (&acc[0][0])[a] = 0.0;
        auto step_word = [&](int s) {
            const uint32_t lo = (s & 2) ? ((s & 1) ? sa.w : sa.z) : ((s & 1) ? sa.y : sa.x);
            const uint32_t hi = (s & 2) ? ((s & 1) ? sb.w : sb.z) : ((s & 1) ? sb.y : sb.x);
            return (s & 4) ? hi : lo;
        };
#pragma unroll 1
        for (int s = 0; s < 8; s++) {
            const uint32_t r = step_word(s);
            if (TR_ACTIVE(r)) {
                const int li = TR_LI(r), lj = TR_LJ(r), slot = TR_SLOT(r);
                const int ti = li < lj ? li : lj, tj = li < lj ? lj : li;
                const bool tr = li > lj;
                double G[6][6];
                if constexpr (MASS) {
                    tri_mass_block(s_geo, slot, ti, tj, props + s_pid[slot], G);
                } else if (UNI) {
                    tri_pair_block<TR_FS, TR_S, 1, TR_SS>(s_fn, s_geo, slot, ti, tj, &Pu, G);
                } else {
                    TriProp pr;
                    load_tri_prop(props + s_pid[slot], pr);
                    tri_pair_block<TR_FS, TR_S, 1, TR_SS>(s_fn, s_geo, slot, ti, tj, &pr, G);
                }
#pragma unroll
                for (int a = 0; a < 6; a++)
#pragma unroll
                    for (int b = 0; b < 6; b++) acc[a][b] = acc[a][b] + (tr ? G[b][a] : G[a][b]);
            }
            if (TR_FLUSH(r)) {
                const fs_tri_node nd = s_node[TR_NODE(r)];
                double *o = data + nd.gbase + 6 * TR_RANK(r);
                const int64_t stride = 6 * (int64_t)nd.m;
#pragma unroll
                for (int a = 0; a < 6; a++) {
                    double2 *row = reinterpret_cast<double2 *>(o + a * stride);      // 16-byte aligned: every term is even
                    row[0] = make_double2(acc[a][0], acc[a][1]);
                    row[1] = make_double2(acc[a][2], acc[a][3]);
                    row[2] = make_double2(acc[a][4], acc[a][5]);
                }
#pragma unroll
                for (int a = 0; a < 36; a++) (&acc[0][0])[a] = 0.0;
            }
        }
        __syncthreads();        // the next patch's phase 1 overwrites the image
    }
}

// Invariants fused_tri_kernel relies on when it indexes LDS and `data` with values read from the plan
// (fs_fused_tri_plan_check): workgroup per patch (grid-stride).  err[0] = violations, err[1] = first rule, err[2] = patch.
__global__ __launch_bounds__(256) void tri_plan_check_kernel(int64_t n_pat, const int64_t *__restrict__ pat_eptr,
                                                            const int32_t *__restrict__ pat_elems, const int64_t *__restrict__ pat_nptr,
                                                            const fs_tri_node *__restrict__ nodes, const int64_t *__restrict__ pat_lptr,
                                                            const uint32_t *__restrict__ steps, int64_t E, int64_t nnz,
                                                            unsigned long long *__restrict__ err) {
    auto report = [&](int rule, int64_t p) {
        if (atomicAdd(&err[0], 1ull) == 0) { err[1] = (unsigned long long)rule; err[2] = (unsigned long long)p; }
    };
    for (int64_t p = blockIdx.x; p < n_pat; p += gridDim.x) {
        const int64_t e0 = pat_eptr[p], n0 = pat_nptr[p], l0 = pat_lptr[p];
        const int64_t ne = pat_eptr[p + 1] - e0, nn = pat_nptr[p + 1] - n0, nl = pat_lptr[p + 1] - l0;
        const bool hdr = ne >= 1 && ne <= FS_TRI_MAX_PATCH_ELEMS && nn >= 1 && nn <= FS_TRI_MAX_PATCH_NODES && nl >= 0 &&
                         nl <= FS_TRI_MAX_PATCH_LANES && (nl & 63) == 0 && e0 >= 0 && n0 >= 0 && l0 >= 0;
        if (!hdr) { if (threadIdx.x == 0) report(1, p); continue; }
        for (int64_t q = threadIdx.x; q < ne; q += blockDim.x)
            if (pat_elems[e0 + q] < 0 || pat_elems[e0 + q] >= E) report(2, p);
        for (int64_t q = threadIdx.x; q < nn; q += blockDim.x) {
            const fs_tri_node nd = nodes[n0 + q];
            if (nd.gbase < 0 || (nd.gbase & 1) || nd.m < 1 || nd.gbase + 36 * (int64_t)nd.m > nnz) report(3, p);
        }
        for (int64_t q = threadIdx.x; q < nl * 8; q += blockDim.x) {
            const uint32_t r = steps[l0 * 8 + q];
            if (TR_ACTIVE(r) && (TR_SLOT(r) >= ne || TR_LI(r) > 2 || TR_LJ(r) > 2)) report(4, p);
            if (TR_FLUSH(r)) {
                if (TR_NODE(r) >= nn) report(5, p);
                else if (TR_RANK(r) >= nodes[n0 + TR_NODE(r)].m) report(5, p);
            }
        }
    }
}

}  // namespace

extern "C" int fs_fused_assemble_tri(const double *xyz, const int32_t *prop_id, const fs_shell_prop *props, int32_t n_props,
                                     int64_t E, int64_t n_patches, const int64_t *pat_eptr, const int32_t *pat_elems,
                                     const int64_t *pat_nptr, const fs_tri_node *nodes, const int64_t *pat_lptr,
                                     const uint32_t *steps, double *data, int32_t *status, void *stream) {
    if (E < 0 || n_props <= 0 || n_patches < 0) return fs_fail_(FS_EINVAL, "fs_fused_assemble_tri: bad sizes", 0);
    if (E == 0 || n_patches == 0) return FS_OK;
    if (!xyz || !prop_id || !props || !pat_eptr || !pat_elems || !pat_nptr || !nodes || !pat_lptr || !steps || !data)
        return fs_fail_(FS_EINVAL, "fs_fused_assemble_tri: null argument", 0);
    int cus = 0;
    if (int rc = fs_device_cu_count_(&cus)) return rc;
    int64_t wgs = ((int64_t)cus / 8) * 8;
    if (wgs < 8) wgs = 8;
    if (wgs > ((n_patches + 7) / 8) * 8) wgs = ((n_patches + 7) / 8) * 8;
    hipStream_t s = (hipStream_t)stream;
    if (n_props == 1)
        hipLaunchKernelGGL(fused_tri_kernel<true>, dim3((unsigned)wgs), dim3(TR_NT), 0, s, xyz, prop_id, props, n_props,
                           n_patches, pat_eptr, pat_elems, pat_nptr, nodes, pat_lptr, steps, data, status);
    else
        hipLaunchKernelGGL(fused_tri_kernel<false>, dim3((unsigned)wgs), dim3(TR_NT), 0, s, xyz, prop_id, props, n_props,
                           n_patches, pat_eptr, pat_elems, pat_nptr, nodes, pat_lptr, steps, data, status);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fs_fail_(FS_EHIP, "fused_tri_kernel", (int)e);
    return FS_OK;
}

extern "C" int fs_fused_assemble_tri_mass(const double *xyz, const int32_t *prop_id, const fs_shell_prop *props, int32_t n_props,
                                          int64_t E, int64_t n_patches, const int64_t *pat_eptr, const int32_t *pat_elems,
                                          const int64_t *pat_nptr, const fs_tri_node *nodes, const int64_t *pat_lptr,
                                          const uint32_t *steps, double *data, void *stream) {
    if (E < 0 || n_props <= 0 || n_patches < 0) return fs_fail_(FS_EINVAL, "fs_fused_assemble_tri_mass: bad sizes", 0);
    if (E == 0 || n_patches == 0) return FS_OK;
    if (!xyz || !prop_id || !props || !pat_eptr || !pat_elems || !pat_nptr || !nodes || !pat_lptr || !steps || !data)
        return fs_fail_(FS_EINVAL, "fs_fused_assemble_tri_mass: null argument", 0);
    int cus = 0;
    if (int rc = fs_device_cu_count_(&cus)) return rc;
    int64_t wgs = ((int64_t)cus * 4 / 8) * 8;            // 17 KB of LDS per workgroup: four per CU
    if (wgs < 8) wgs = 8;
    if (wgs > ((n_patches + 7) / 8) * 8) wgs = ((n_patches + 7) / 8) * 8;
    hipLaunchKernelGGL((fused_tri_kernel<false, true>), dim3((unsigned)wgs), dim3(TR_NT), 0, (hipStream_t)stream, xyz, prop_id,
                       props, n_props, n_patches, pat_eptr, pat_elems, pat_nptr, nodes, pat_lptr, steps, data,
                       (int32_t *)nullptr);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fs_fail_(FS_EHIP, "fused_tri_kernel (mass)", (int)e);
    return FS_OK;
}

extern "C" int fs_fused_tri_plan_check(int64_t n_patches, const int64_t *pat_eptr, const int32_t *pat_elems,
                                       const int64_t *pat_nptr, const fs_tri_node *nodes, const int64_t *pat_lptr,
                                       const uint32_t *steps, int64_t E, int64_t nnz, void *stream) {
    if (E < 0 || n_patches < 0 || nnz < 0) return fs_fail_(FS_EINVAL, "fs_fused_tri_plan_check: bad sizes", 0);
    if (n_patches == 0) return FS_OK;
    if (!pat_eptr || !pat_elems || !pat_nptr || !nodes || !pat_lptr || !steps)
        return fs_fail_(FS_EINVAL, "fs_fused_tri_plan_check: null argument", 0);
    hipStream_t s = (hipStream_t)stream;
    unsigned long long *err = nullptr, host[3] = {0, 0, 0};
    hipError_t e = hipMalloc(&err, sizeof host);
    if (e != hipSuccess) return fs_fail_(FS_EHIP, "fs_fused_tri_plan_check: hipMalloc", (int)e);
    e = hipMemsetAsync(err, 0, sizeof host, s);
    if (e == hipSuccess) {
        const unsigned grid = (unsigned)(n_patches < 8192 ? n_patches : 8192);
        hipLaunchKernelGGL(tri_plan_check_kernel, dim3(grid), dim3(256), 0, s, n_patches, pat_eptr, pat_elems, pat_nptr, nodes,
                           pat_lptr, steps, E, nnz, err);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(host, err, sizeof host, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(err);
    if (e != hipSuccess) return fs_fail_(FS_EHIP, "fs_fused_tri_plan_check", (int)e);
    if (host[0]) {
        static const char *const rules[] = {"", "patch header out of range (elements, nodes, lanes)", "element index outside [0, E)",
            "node slab outside the value array or misaligned", "step names a slot beyond its patch or a local node > 2",
            "flush with node / neighbour rank out of range"};
        char msg[200];
        snprintf(msg, sizeof msg, "fs_fused_tri_plan_check: %llu violations; first: %s (patch %lld)", host[0],
                 rules[host[1] < 6 ? host[1] : 0], (long long)host[2]);
        return fs_fail_(FS_EINVAL, msg, 0);
    }
    return FS_OK;
}
