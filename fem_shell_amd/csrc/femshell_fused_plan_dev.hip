// femshell_fused_plan_dev.hip -- the patch plan of fs_fused_assemble_quad built ON THE DEVICE (MI355X / gfx950).
//
// Same plan, bit for bit, as the host builder fs_fused_plan_build (femshell_fused_plan.cpp; layout rules at fs_fused_rec
// in include/femshell_hip.h), without the trip of the node graph to the host and of gigabytes of records back: 10 M
// elements need 3.4 GB of records.  Integer bookkeeping only (no counterpart in the reference, whose loops need no plan):
//
//   begin : node -> elements lists by one stable radix sort of (node, element) pairs (rocPRIM; element order survives
//           inside a node), lead nodes sorted by patch id the same way, then ONE WORKGROUP PER PATCH counts the patch's
//           elements and stage-2 lanes; two scans turn the counts into pat_eptr / pat_rptr.
//   finish: the same kernel again, this time writing pat_elems and the 64-byte lane records into the caller's arrays.
//
// A patch is small (<= 64 elements, <= 256 lead nodes, <= 640 blocks), so everything a workgroup needs sits in LDS:
// the element lists of its nodes, the corner nodes of its elements, its tasks (one per CSR block it produces) and their
// contributions.  What the host code does with sorts is done here by counting ranks inside a group of <= 12 lead nodes.
// Meshes beyond the LDS capacities (FS_EINVAL, "device plan capacity") are left to the host builder.
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <stdint.h>

#include "femshell_hip.h"
#include "femshell_internal.h"

struct fs_fused_plan_dev {
    int64_t E = 0, n_nodes = 0, np = 0, n_slots = 0, n_recs = 0, n_evals = 0;
    int32_t max_patch_elems = 0;
    const int64_t *conn = nullptr, *nbr_ptr = nullptr;      // the caller's arrays: alive until finish
    const int32_t *nbr = nullptr;
    int64_t *n2e_ptr = nullptr, *pstart = nullptr, *eptr = nullptr, *rptr = nullptr;
    int32_t *n2e = nullptr, *order = nullptr;
    unsigned long long *nev = nullptr;
    int *err = nullptr;
};

namespace {

constexpr int PB_BS = 64;        // lanes per patch (one wave)
constexpr int PB_MAXN = 256;     // lead nodes of a patch (64 elements have at most 256 corners)
constexpr int PB_MAXC = 256;     // (node, element) incidences of a patch
constexpr int PB_MAXT = 640;     // tasks = CSR blocks a patch produces (64 elements x 10 node pairs), also contributions
constexpr int PB_MAXL = 1024;    // stage-2 lanes of a patch
constexpr int PB_MAXG = 64;      // lane groups of a patch

struct PatchLds {
    int64_t t_dst[PB_MAXT], t_dstt[PB_MAXT];
    int32_t t_str[PB_MAXT], t_strt[PB_MAXT], t_hb[PB_MAXT];
    int32_t node[PB_MAXN], cpre[PB_MAXN + 1], rpre[PB_MAXN + 1];
    int32_t n2e[PB_MAXC], raw[64], el[64], conn[64][4];
    uint32_t pool[PB_MAXT], lflags[PB_MAXL];
    int32_t gs[PB_MAXG + 1], half[PB_MAXG], gl[PB_MAXG + 1], mc[PB_MAXL / 64];
    int16_t t_q[PB_MAXT], t_n[PB_MAXT], t_ev[PB_MAXT], t_lead[PB_MAXT], t_long[PB_MAXT], longlist[PB_MAXT];
    int32_t scan[PB_BS];
    int32_t ne, err;
};

__device__ inline int hw_of(int n) { return n > 4 ? 32 : n >= 3 ? 4 : n == 2 ? 2 : 1; }      // half-lanes of a task
__device__ inline int cls_of(int n) { return n > 4 ? 0 : n >= 3 ? 1 : n == 2 ? 2 : 3; }

// exclusive prefix sum over the workgroup's lanes (in lane order); `total` = the sum
__device__ inline int block_scan(int v, int32_t *buf, int &total) {
    const int tid = (int)threadIdx.x;
    buf[tid] = v;
    __syncthreads();
    for (int off = 1; off < PB_BS; off <<= 1) {
        const int a = tid >= off ? buf[tid - off] : 0;
        __syncthreads();
        buf[tid] += a;
        __syncthreads();
    }
    const int incl = buf[tid];
    total = buf[PB_BS - 1];
    __syncthreads();
    return incl - v;
}

// largest q in [0, n) with pre[q] <= c (pre ascending, pre[0] = 0, c < pre[n])
__device__ inline int seg_of(const int32_t *pre, int n, int c) {
    int lo = 0, hi = n;            // invariant: pre[lo] <= c < pre[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (pre[mid] <= c) lo = mid; else hi = mid;
    }
    return lo;
}

template <class T>
__device__ inline int find_sorted(const T *b, int n, T key) {     // position of key in the ascending range or -1
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (b[mid] < key) lo = mid + 1; else hi = mid;
    }
    return (lo < n && b[lo] == key) ? lo : -1;
}

// error codes of the patch kernel (first one wins): the host builder's -1, -3, -4, -5; -6 = beyond the LDS capacities
__device__ inline void patch_fail(int *err, int code) { atomicCAS(err, 0, code); }

template <bool WRITE>
__global__ __launch_bounds__(PB_BS) void fused_plan_patch_kernel(
    const int64_t *__restrict__ conn, const int64_t *__restrict__ nbr_ptr, const int32_t *__restrict__ nbr,
    const int64_t *__restrict__ n2e_ptr, const int32_t *__restrict__ n2e, const int32_t *__restrict__ order,
    const int64_t *__restrict__ pstart, int64_t np, int max_patch_elems, int64_t *__restrict__ ne_out,
    int64_t *__restrict__ nl_out, unsigned long long *__restrict__ nev_total, int *__restrict__ err,
    const int64_t *__restrict__ pat_eptr, const int64_t *__restrict__ pat_rptr, int32_t *__restrict__ pat_elems,
    fs_fused_rec *__restrict__ recs) {
    __shared__ PatchLds L;
    const int tid = (int)threadIdx.x;
    for (int64_t p = blockIdx.x; p < np; p += gridDim.x) {
        __syncthreads();                                  // the previous patch is done with L
        int64_t out_ne = 0, out_nl = 0;
        do {                                              // every `break` below is taken by the whole workgroup
            const int64_t ns = pstart[p];
            const int nn = (int)(pstart[p + 1] - ns);
            if (nn > PB_MAXN) { if (tid == 0) patch_fail(err, -6); break; }
            for (int i = tid; i < nn; i += PB_BS) {
                const int64_t nd = order[ns + i];
                L.node[i] = (int32_t)nd;
                const int64_t c = n2e_ptr[nd + 1] - n2e_ptr[nd], r = nbr_ptr[nd + 1] - nbr_ptr[nd];
                L.cpre[i + 1] = (int32_t)(c < (1 << 20) ? c : (1 << 20));
                L.rpre[i + 1] = (int32_t)(r < (1 << 20) ? r : (1 << 20));
            }
            if (tid == 0) { L.cpre[0] = 0; L.rpre[0] = 0; L.ne = 0; L.err = 0; }
            __syncthreads();
            if (tid < 2) {                                // two short serial prefix sums (nn <= 256)
                int32_t *pre = tid == 0 ? L.cpre : L.rpre;
                int acc = 0;
                for (int i = 1; i <= nn; i++) { acc += pre[i]; if (acc > (1 << 24)) acc = 1 << 24; pre[i] = acc; }
            }
            __syncthreads();
            const int totc = L.cpre[nn], totr = L.rpre[nn];
            // every element has at most 4 incidences with the patch's nodes: more than 256 incidences = more than 64 elements
            if (totc > PB_MAXC) { if (tid == 0) patch_fail(err, -4); break; }
            // the element lists of the patch's nodes (ascending inside a node; an element that lists a node twice appears twice)
            for (int c = tid; c < totc; c += PB_BS) {
                const int q = seg_of(L.cpre, nn, c);
                L.n2e[c] = n2e[n2e_ptr[L.node[q]] + (c - L.cpre[q])];
            }
            __syncthreads();
            // elements touching the patch: an element is taken at the smallest of its corners that the patch leads
            for (int base = 0; base < totc; base += PB_BS) {
                const int c = base + tid;
                if (c < totc) {
                    const int q = seg_of(L.cpre, nn, c);
                    const int32_t e = L.n2e[c];
                    if (!(c > L.cpre[q] && L.n2e[c - 1] == e)) {
                        const int64_t *cn = conn + 4 * (int64_t)e;
                        int32_t mn = INT32_MAX;
                        for (int k = 0; k < 4; k++) {
                            const int32_t v = (int32_t)cn[k];
                            if (v < mn && find_sorted(L.node, nn, v) >= 0) mn = v;
                        }
                        if (mn == L.node[q]) {
                            const int pos = atomicAdd(&L.ne, 1);
                            if (pos < 64) L.raw[pos] = e;
                        }
                    }
                }
            }
            __syncthreads();
            const int ne = L.ne;
            out_ne = ne;
            if (ne > max_patch_elems) { if (tid == 0) patch_fail(err, -4); break; }
            for (int i = tid; i < ne; i += PB_BS) {                      // ascending element index = slot
                const int32_t v = L.raw[i];
                int r = 0;
                for (int m = 0; m < ne; m++) r += L.raw[m] < v ? 1 : 0;
                L.el[r] = v;
            }
            __syncthreads();
            for (int i = tid; i < 4 * ne; i += PB_BS) L.conn[i >> 2][i & 3] = (int32_t)conn[4 * (int64_t)L.el[i >> 2] + (i & 3)];
            if (WRITE) {
                const int64_t e0 = pat_eptr[p];
                for (int i = tid; i < ne; i += PB_BS) pat_elems[e0 + i] = L.el[i];
            }
            __syncthreads();
            // ---- tasks: the blocks (I, J) the patch produces, in (I, J) order, with their contributions in element order
            int nt = 0, nev = 0;
            bool over = false;
            for (int base = 0; base < totr; base += PB_BS) {
                const int c = base + tid;
                bool keep = false, partner = false;
                int n = 0, q = 0;
                int32_t I = 0, J = 0;
                int64_t r0 = 0, b = 0, rJ0 = 0, rJ1 = 0;
                int posI = -1;
                if (c < totr) {
                    q = seg_of(L.rpre, nn, c);
                    I = L.node[q];
                    r0 = nbr_ptr[I];
                    b = r0 + (c - L.rpre[q]);
                    J = nbr[b];
                    rJ0 = nbr_ptr[J];
                    rJ1 = nbr_ptr[J + 1];
                    posI = find_sorted(nbr + rJ0, (int)(rJ1 - rJ0), I);
                    partner = posI >= 0;
                    keep = !(I > J && partner);             // else written as the transposed copy of (J, I) by J's patch
                    if (keep) {
                        for (int a = L.cpre[q]; a < L.cpre[q + 1]; a++) {
                            const int32_t e = L.n2e[a];
                            if (a > L.cpre[q] && L.n2e[a - 1] == e) continue;
                            const int slot = find_sorted(L.el, ne, e);
                            const int32_t *cn = L.conn[slot];
                            n += (cn[0] == J || cn[1] == J || cn[2] == J || cn[3] == J) ? 1 : 0;
                        }
                        if (n > 32) { L.err = -1; n = 32; }
                        else if (n > 0 && !partner) L.err = -3;
                    }
                }
                int tot_k, tot_n;
                const int tpos = block_scan(keep ? 1 : 0, L.scan, tot_k);
                const int epos = block_scan(keep ? n : 0, L.scan, tot_n);
                if (nt + tot_k > PB_MAXT || nev + tot_n > PB_MAXT) { over = true; break; }
                if (keep) {
                    const int t = nt + tpos;
                    L.t_q[t] = (int16_t)q;
                    L.t_n[t] = (int16_t)n;
                    L.t_ev[t] = (int16_t)(nev + epos);
                    L.t_dst[t] = 36 * r0 + 6 * (b - r0);
                    L.t_str[t] = 6 * (int32_t)(nbr_ptr[I + 1] - r0);
                    L.t_dstt[t] = (I < J && partner) ? 36 * rJ0 + 6 * (int64_t)posI : -1;
                    L.t_strt[t] = 6 * (int32_t)(rJ1 - rJ0);
                    int w = nev + epos, left = n;
                    for (int a = L.cpre[q]; a < L.cpre[q + 1] && left > 0; a++) {
                        const int32_t e = L.n2e[a];
                        if (a > L.cpre[q] && L.n2e[a - 1] == e) continue;
                        const int slot = find_sorted(L.el, ne, e);
                        const int32_t *cn = L.conn[slot];
                        int li = -1, lj = -1;
                        for (int k = 3; k >= 0; k--) { if (cn[k] == I) li = k; if (cn[k] == J) lj = k; }     // first match
                        if (lj < 0) continue;
                        const int ti = li < lj ? li : lj, tj = li < lj ? lj : li;
                        L.pool[w++] = (uint32_t)slot | ((uint32_t)ti << 8) | ((uint32_t)tj << 10) | ((li > lj ? 1u : 0u) << 12);
                        left--;
                    }
                }
                nt += tot_k;
                nev += tot_n;
            }
            __syncthreads();
            if (over) { if (tid == 0) patch_fail(err, -6); break; }
            if (L.err) { if (tid == 0) patch_fail(err, L.err); break; }
            // ---- lane layout.  Lead nodes in ascending order -> groups of 12; a task with more than 4 contributions
            // (long) moves to an extra group, 4 per group.  Inside a group: 2-lane tasks, 1-lane tasks, half-lane tasks,
            // each class in (I, J) order.
            int nlead = 0, nlong = 0;
            for (int base = 0; base < nt; base += PB_BS) {
                const int t = base + tid;
                int f = 0, g = 0;
                if (t < nt) { f = (t == 0 || L.t_q[t] != L.t_q[t - 1]) ? 1 : 0; g = L.t_n[t] > 4 ? 1 : 0; }
                int tot;
                const int ex = block_scan(f | (g << 16), L.scan, tot);
                if (t < nt) {
                    L.t_lead[t] = (int16_t)(nlead + (ex & 0xffff) + f - 1);
                    L.t_long[t] = (int16_t)(g ? nlong + (ex >> 16) : -1);
                    if (g) L.longlist[nlong + (ex >> 16)] = (int16_t)t;
                }
                nlead += tot & 0xffff;
                nlong += tot >> 16;
            }
            const int gnorm = (nlead + 11) / 12, ngroups = gnorm + (nlong + 3) / 4;
            if (ngroups > PB_MAXG) { if (tid == 0) patch_fail(err, -6); break; }
            for (int g = tid; g < ngroups; g += PB_BS) {
                const int rest = nlong - 4 * (g - gnorm);
                L.half[g] = g < gnorm ? 0 : 32 * (rest < 4 ? rest : 4);
            }
            if (tid == 0) L.gs[gnorm] = nt;
            __syncthreads();
            for (int t = tid; t < nt; t += PB_BS) {
                const int g = L.t_lead[t] / 12;
                if (t == 0 || L.t_lead[t - 1] / 12 != g) L.gs[g] = t;
                if (L.t_long[t] < 0) atomicAdd(&L.half[g], hw_of(L.t_n[t]));
            }
            __syncthreads();
            if (tid == 0) {
                int lanes = 0;
                for (int g = 0; g < ngroups; g++) { L.gl[g] = lanes; lanes += ((L.half[g] + 1) / 2 + 63) / 64 * 64; }
                L.gl[ngroups] = lanes;
            }
            __syncthreads();
            const int lanes = L.gl[ngroups];
            out_nl = lanes;
            if (lanes > PB_MAXL) { if (tid == 0) patch_fail(err, -6); break; }
            if (!WRITE) { if (tid == 0) atomicAdd(nev_total, (unsigned long long)nev); break; }
            // half-lane position of every task
            for (int t = tid; t < nt; t += PB_BS) {
                const int n = L.t_n[t];
                int hb;
                if (n > 4) {
                    const int li = L.t_long[t];
                    hb = 2 * L.gl[gnorm + li / 4] + 32 * (li % 4);
                } else {
                    const int g = L.t_lead[t] / 12, cls = cls_of(n);
                    int off = 0;
                    for (int s = L.gs[g]; s < L.gs[g + 1]; s++) {
                        const int m = L.t_n[s];
                        if (m > 4) continue;
                        const int cs = cls_of(m);
                        if (cs < cls || (cs == cls && s < t)) off += hw_of(m);
                    }
                    hb = 2 * L.gl[g] + off;
                }
                L.t_hb[t] = hb;
            }
            for (int l = tid; l < lanes; l += PB_BS) L.lflags[l] = 0;
            for (int w = tid; w < lanes / 64; w += PB_BS) L.mc[w] = 1;
            __syncthreads();
            // flags of every lane
            for (int t = tid; t < nt; t += PB_BS) {
                const int n = L.t_n[t], hw = hw_of(n), hb = L.t_hb[t], lane0 = hb >> 1;
                const int nl = hw >= 2 ? ((n > 1 ? n : 1) + 1) / 2 : 1;
                const int last = lane0 + nl - 1;
                const bool half_task = hw == 1;
                const int which = half_task ? (hb & 1) : 0;
                atomicOr(&L.lflags[last], (which == 0 ? FS_FL_OUT0 : FS_FL_OUT1) | ((half_task || n < 4) ? FS_FL_PAD : 0u));
                if (n > 4) atomicOr(&L.lflags[lane0], FS_FL_ZERO_START);
                for (int c = 1; c < nl; c++) atomicOr(&L.lflags[lane0 + c], FS_FL_CHAIN | ((uint32_t)c << 16));
                atomicMax(&L.mc[lane0 >> 6], nl);
            }
            __syncthreads();
            fs_fused_rec *R = recs + pat_rptr[p];
            for (int l = tid; l < lanes; l += PB_BS) {
                fs_fused_rec r;
                r.dst[0] = r.dst[1] = r.dst_t[0] = r.dst_t[1] = -1;
                r.stride[0] = r.stride[1] = r.stride_t[0] = r.stride_t[1] = 0;
                r.ev[0] = r.ev[1] = 0;
                r.flags = L.lflags[l] | ((uint32_t)L.mc[l >> 6] << 8);
                r.slots = 0xFFFFFFFFu;
                R[l] = r;
            }
            __syncthreads();                              // the blank records are in memory before their fields are set
            // contributions, destinations, and the staging slot / transposed rank of every block inside its wave
            for (int t = tid; t < nt; t += PB_BS) {
                const int n = L.t_n[t], hw = hw_of(n), hb = L.t_hb[t], lane0 = hb >> 1;
                const int nl = hw >= 2 ? ((n > 1 ? n : 1) + 1) / 2 : 1;
                const int last = lane0 + nl - 1;
                const int which = hw == 1 ? (hb & 1) : 0;
                const int64_t dst = L.t_dst[t], dstt = L.t_dstt[t];
                for (int c = 0; c < n; c++) R[(hb + c) >> 1].ev[(hb + c) & 1] = L.pool[L.t_ev[t] + c] | (1u << 13);
                R[last].dst[which] = dst;
                R[last].stride[which] = L.t_str[t];
                R[last].dst_t[which] = dstt;
                R[last].stride_t[which] = dstt >= 0 ? L.t_strt[t] : 0;
                const int wave = last >> 6;
                int rank0 = 0, rank1 = 0;
                int s0, s1;
                const bool is_long = n > 4;
                if (is_long) { s0 = 4 * (L.t_long[t] / 4); s1 = s0 + 4 < nlong ? s0 + 4 : nlong; }
                else { const int g = L.t_lead[t] / 12; s0 = L.gs[g]; s1 = L.gs[g + 1]; }
                for (int j = s0; j < s1; j++) {
                    const int s = is_long ? (int)L.longlist[j] : j;
                    const int m = L.t_n[s];
                    if (s == t || (!is_long && m > 4)) continue;
                    const int hws = hw_of(m), hbs = L.t_hb[s];
                    const int nls = hws >= 2 ? ((m > 1 ? m : 1) + 1) / 2 : 1;
                    const int lasts = (hbs >> 1) + nls - 1;
                    if ((lasts >> 6) != wave) continue;
                    const int whichs = hws == 1 ? (hbs & 1) : 0;
                    if (whichs < which || (whichs == which && L.t_dst[s] < dst)) rank0++;
                    if (dstt >= 0 && L.t_dstt[s] >= 0 && L.t_dstt[s] < dstt) rank1++;
                }
                if (rank0 > 254 || rank1 > 254) patch_fail(err, -5);
                uint8_t *sl = reinterpret_cast<uint8_t *>(&R[last].slots);
                sl[2 * which] = (uint8_t)rank0;
                if (dstt >= 0) sl[2 * which + 1] = (uint8_t)rank1;
            }
        } while (0);
        if (!WRITE && tid == 0) { ne_out[p] = out_ne; nl_out[p] = out_nl; }
    }
}

// (node, element) pairs of the connectivity, element-major; connectivity range check
__global__ __launch_bounds__(256) void pd_incidence_kernel(const int64_t *__restrict__ conn, int64_t n, int64_t n_nodes,
                                                           uint32_t *__restrict__ keys, int32_t *__restrict__ vals,
                                                           int *__restrict__ bad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t v = conn[i];
    if (v < 0 || v >= n_nodes) { *bad = 1; keys[i] = 0; vals[i] = 0; return; }
    keys[i] = (uint32_t)v;
    vals[i] = (int32_t)(i >> 2);
}

// ptr[i] = first position of the ascending `keys` that holds a value >= i  (i = 0 .. n_nodes)
__global__ __launch_bounds__(256) void pd_ptr_kernel(const uint32_t *__restrict__ keys, int64_t n, int64_t n_nodes,
                                                     int64_t *__restrict__ ptr) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n_nodes) return;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)keys[mid] < i) lo = mid + 1; else hi = mid;
    }
    ptr[i] = lo;
}

// sort key of a node: its patch id when it leads a block row, else the largest key (sorted to the end)
__global__ __launch_bounds__(256) void pd_patch_keys_kernel(const int64_t *__restrict__ nbr_ptr,
                                                            const int64_t *__restrict__ node_patch, int64_t n_nodes,
                                                            uint64_t *__restrict__ keys, int32_t *__restrict__ vals,
                                                            int *__restrict__ bad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const bool lead = nbr_ptr[i + 1] > nbr_ptr[i];
    const int64_t pid = node_patch[i];
    if (lead && pid < 0) *bad = 2;
    keys[i] = lead ? (uint64_t)pid : ~(uint64_t)0;
    vals[i] = (int32_t)i;
}

// flag = 1 where a new patch starts in the sorted keys; n_lead = number of lead nodes
__global__ __launch_bounds__(256) void pd_patch_flags_kernel(const uint64_t *__restrict__ keys, int64_t n,
                                                             int64_t *__restrict__ flag, int64_t *__restrict__ n_lead) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool lead = keys[i] != ~(uint64_t)0;
    flag[i] = (lead && (i == 0 || keys[i] != keys[i - 1])) ? 1 : 0;
    if (lead && (i == n - 1 || keys[i + 1] == ~(uint64_t)0)) *n_lead = i + 1;
}

__global__ __launch_bounds__(256) void pd_patch_starts_kernel(const uint64_t *__restrict__ keys,
                                                              const int64_t *__restrict__ pidx, int64_t n,
                                                              int64_t *__restrict__ pstart) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (keys[i] != ~(uint64_t)0 && (i == 0 || keys[i] != keys[i - 1])) pstart[pidx[i]] = i;
}

inline unsigned grid_of(int64_t n, int block) { return (unsigned)((n + block - 1) / block); }

void release(fs_fused_plan_dev *h) {
    if (!h) return;
    (void)hipFree(h->n2e_ptr); (void)hipFree(h->pstart); (void)hipFree(h->eptr); (void)hipFree(h->rptr);
    (void)hipFree(h->n2e); (void)hipFree(h->order); (void)hipFree(h->nev); (void)hipFree(h->err);
    delete h;
}

const char *patch_error(int code) {
    return code == -4   ? "fs_fused_plan_device: a patch touches more than max_patch_elems elements"
           : code == -1 ? "fs_fused_plan_device: a node pair is shared by more than 32 elements"
           : code == -3 ? "fs_fused_plan_device: block with contributions but no transposed partner (pattern not symmetric)"
           : code == -5 ? "fs_fused_plan_device: more than 255 blocks stored by one wave"
                        : "fs_fused_plan_device: device plan capacity exceeded (use the host builder fs_fused_plan_build)";
}

}  // namespace

extern "C" {

int fs_fused_plan_device_begin(const int64_t *conn, int64_t E, int64_t n_nodes, const int64_t *nbr_ptr, const int32_t *nbr,
                               const int64_t *node_patch, int32_t max_patch_elems, fs_fused_plan_dev **plan,
                               int64_t *n_patches, int64_t *n_elem_slots, int64_t *n_recs, int64_t *n_evals, void *stream) {
    if (!conn || !nbr_ptr || !nbr || !node_patch || !plan || !n_patches || !n_elem_slots || !n_recs)
        return fs_fail_(FS_EINVAL, "fs_fused_plan_device_begin: null argument", 0);
    *plan = nullptr;
    if (E <= 0 || n_nodes <= 0 || E > INT32_MAX || n_nodes >= INT32_MAX)
        return fs_fail_(FS_EINVAL, "fs_fused_plan_device_begin: bad sizes", 0);
    if (max_patch_elems < 16 || max_patch_elems > FS_FUSED_MAX_PATCH_ELEMS)
        return fs_fail_(FS_EINVAL, "fs_fused_plan_device_begin: max_patch_elems must lie in [16, FS_FUSED_MAX_PATCH_ELEMS]", 0);
    hipStream_t s = (hipStream_t)stream;
    fs_fused_plan_dev *h = new fs_fused_plan_dev;
    h->E = E; h->n_nodes = n_nodes; h->max_patch_elems = max_patch_elems;
    h->conn = conn; h->nbr_ptr = nbr_ptr; h->nbr = nbr;
    const int64_t n_inc = 4 * E;
    uint32_t *k_in = nullptr, *k_out = nullptr;
    int32_t *v_in = nullptr, *pv_in = nullptr;
    uint64_t *pk_in = nullptr, *pk_out = nullptr;
    int64_t *flag = nullptr, *scalars = nullptr;
    void *tmp = nullptr;
    int rc = FS_OK;
    auto bail = [&](int code) {
        (void)hipFree(k_in); (void)hipFree(k_out); (void)hipFree(v_in); (void)hipFree(pv_in); (void)hipFree(pk_in);
        (void)hipFree(pk_out); (void)hipFree(flag); (void)hipFree(scalars); (void)hipFree(tmp);
        release(h);
        return code;
    };
#define PD_TRY(call, what)                                                               \
    do {                                                                                 \
        hipError_t e__ = (call);                                                         \
        if (e__ != hipSuccess) { rc = fs_fail_(FS_EHIP, what, (int)e__); return bail(rc); } \
    } while (0)
    PD_TRY(hipMalloc(&h->err, sizeof(int)), "hipMalloc");
    PD_TRY(hipMalloc(&h->nev, sizeof(unsigned long long)), "hipMalloc");
    PD_TRY(hipMalloc(&scalars, 4 * sizeof(int64_t)), "hipMalloc");
    PD_TRY(hipMemsetAsync(h->err, 0, sizeof(int), s), "hipMemsetAsync");
    PD_TRY(hipMemsetAsync(h->nev, 0, sizeof(unsigned long long), s), "hipMemsetAsync");
    PD_TRY(hipMemsetAsync(scalars, 0, 4 * sizeof(int64_t), s), "hipMemsetAsync");
    // ---- node -> elements: stable sort of the (node, element) incidences by node
    PD_TRY(hipMalloc(&k_in, (size_t)n_inc * 4), "hipMalloc incidence keys");
    PD_TRY(hipMalloc(&v_in, (size_t)n_inc * 4), "hipMalloc incidence values");
    PD_TRY(hipMalloc(&k_out, (size_t)n_inc * 4), "hipMalloc incidence keys");
    PD_TRY(hipMalloc(&h->n2e, (size_t)n_inc * 4), "hipMalloc node -> elements");
    PD_TRY(hipMalloc(&h->n2e_ptr, (size_t)(n_nodes + 1) * 8), "hipMalloc node -> elements");
    hipLaunchKernelGGL(pd_incidence_kernel, dim3(grid_of(n_inc, 256)), dim3(256), 0, s, conn, n_inc, n_nodes, k_in, v_in, h->err);
    PD_TRY(hipGetLastError(), "pd_incidence_kernel");
    int bits = 1;
    while (bits < 32 && ((int64_t)1 << bits) < n_nodes) bits++;
    size_t tmp_bytes = 0;
    PD_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, k_in, k_out, v_in, h->n2e, (size_t)n_inc, 0, bits, s), "radix_sort_pairs (size)");
    PD_TRY(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 8), "hipMalloc sort workspace");
    PD_TRY(rocprim::radix_sort_pairs(tmp, tmp_bytes, k_in, k_out, v_in, h->n2e, (size_t)n_inc, 0, bits, s), "radix_sort_pairs");
    hipLaunchKernelGGL(pd_ptr_kernel, dim3(grid_of(n_nodes + 1, 256)), dim3(256), 0, s, k_out, n_inc, n_nodes, h->n2e_ptr);
    PD_TRY(hipGetLastError(), "pd_ptr_kernel");
    (void)hipFree(tmp); tmp = nullptr;                 // hipFree synchronises
    (void)hipFree(k_in); k_in = nullptr;
    (void)hipFree(v_in); v_in = nullptr;
    (void)hipFree(k_out); k_out = nullptr;
    // ---- lead nodes by (patch, node): stable sort by patch id; patch ranges
    PD_TRY(hipMalloc(&pk_in, (size_t)n_nodes * 8), "hipMalloc patch keys");
    PD_TRY(hipMalloc(&pk_out, (size_t)n_nodes * 8), "hipMalloc patch keys");
    PD_TRY(hipMalloc(&pv_in, (size_t)n_nodes * 4), "hipMalloc patch values");
    PD_TRY(hipMalloc(&h->order, (size_t)n_nodes * 4), "hipMalloc node order");
    PD_TRY(hipMalloc(&flag, (size_t)n_nodes * 8), "hipMalloc flags");
    hipLaunchKernelGGL(pd_patch_keys_kernel, dim3(grid_of(n_nodes, 256)), dim3(256), 0, s, nbr_ptr, node_patch, n_nodes, pk_in,
                       pv_in, h->err);
    PD_TRY(hipGetLastError(), "pd_patch_keys_kernel");
    tmp_bytes = 0;
    PD_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, pk_in, pk_out, pv_in, h->order, (size_t)n_nodes, 0, 64, s), "radix_sort_pairs (size)");
    PD_TRY(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 8), "hipMalloc sort workspace");
    PD_TRY(rocprim::radix_sort_pairs(tmp, tmp_bytes, pk_in, pk_out, pv_in, h->order, (size_t)n_nodes, 0, 64, s), "radix_sort_pairs");
    hipLaunchKernelGGL(pd_patch_flags_kernel, dim3(grid_of(n_nodes, 256)), dim3(256), 0, s, pk_out, n_nodes, flag, scalars);
    PD_TRY(hipGetLastError(), "pd_patch_flags_kernel");
    (void)hipFree(tmp); tmp = nullptr;
    int64_t last_flag = 0, last_idx = 0, n_lead = 0;
    PD_TRY(hipMemcpyAsync(&last_flag, flag + n_nodes - 1, 8, hipMemcpyDeviceToHost, s), "copy");
    size_t scan_bytes = 0;
    PD_TRY(rocprim::exclusive_scan(nullptr, scan_bytes, flag, flag, (int64_t)0, (size_t)n_nodes, rocprim::plus<int64_t>(), s),
           "exclusive_scan (size)");
    PD_TRY(hipMalloc(&tmp, scan_bytes ? scan_bytes : 8), "hipMalloc scan workspace");
    PD_TRY(rocprim::exclusive_scan(tmp, scan_bytes, flag, flag, (int64_t)0, (size_t)n_nodes, rocprim::plus<int64_t>(), s),
           "exclusive_scan");
    PD_TRY(hipMemcpyAsync(&last_idx, flag + n_nodes - 1, 8, hipMemcpyDeviceToHost, s), "copy");
    PD_TRY(hipMemcpyAsync(&n_lead, scalars, 8, hipMemcpyDeviceToHost, s), "copy");
    int err_h = 0;
    PD_TRY(hipMemcpyAsync(&err_h, h->err, sizeof(int), hipMemcpyDeviceToHost, s), "copy");
    PD_TRY(hipStreamSynchronize(s), "hipStreamSynchronize");
    if (err_h) {
        rc = fs_fail_(FS_EINVAL, err_h == 1 ? "fs_fused_plan_device_begin: connectivity out of range"
                                            : "fs_fused_plan_device_begin: negative patch id", 0);
        return bail(rc);
    }
    const int64_t np = last_idx + last_flag;
    h->np = np;
    PD_TRY(hipMalloc(&h->pstart, (size_t)(np + 1) * 8), "hipMalloc patch ranges");
    PD_TRY(hipMalloc(&h->eptr, (size_t)(np + 1) * 8), "hipMalloc patch ranges");
    PD_TRY(hipMalloc(&h->rptr, (size_t)(np + 1) * 8), "hipMalloc patch ranges");
    hipLaunchKernelGGL(pd_patch_starts_kernel, dim3(grid_of(n_nodes, 256)), dim3(256), 0, s, pk_out, flag, n_nodes, h->pstart);
    PD_TRY(hipGetLastError(), "pd_patch_starts_kernel");
    PD_TRY(hipMemcpyAsync(h->pstart + np, &n_lead, 8, hipMemcpyHostToDevice, s), "copy");
    PD_TRY(hipMemsetAsync(h->eptr + np, 0, 8, s), "hipMemsetAsync");
    PD_TRY(hipMemsetAsync(h->rptr + np, 0, 8, s), "hipMemsetAsync");
    // ---- count pass: elements and lanes of every patch, then their prefix sums
    if (np > 0) {
        const unsigned grid = (unsigned)(np < (1 << 20) ? np : (1 << 20));
        hipLaunchKernelGGL(fused_plan_patch_kernel<false>, dim3(grid), dim3(PB_BS), 0, s, conn, nbr_ptr, nbr, h->n2e_ptr, h->n2e,
                           h->order, h->pstart, np, (int)max_patch_elems, h->eptr, h->rptr, h->nev, h->err,
                           (const int64_t *)nullptr, (const int64_t *)nullptr, (int32_t *)nullptr, (fs_fused_rec *)nullptr);
        PD_TRY(hipGetLastError(), "fused_plan_patch_kernel (count)");
    }
    (void)hipFree(tmp); tmp = nullptr;
    scan_bytes = 0;
    PD_TRY(rocprim::exclusive_scan(nullptr, scan_bytes, h->eptr, h->eptr, (int64_t)0, (size_t)(np + 1), rocprim::plus<int64_t>(), s),
           "exclusive_scan (size)");
    PD_TRY(hipMalloc(&tmp, scan_bytes ? scan_bytes : 8), "hipMalloc scan workspace");
    PD_TRY(rocprim::exclusive_scan(tmp, scan_bytes, h->eptr, h->eptr, (int64_t)0, (size_t)(np + 1), rocprim::plus<int64_t>(), s),
           "exclusive_scan");
    PD_TRY(rocprim::exclusive_scan(tmp, scan_bytes, h->rptr, h->rptr, (int64_t)0, (size_t)(np + 1), rocprim::plus<int64_t>(), s),
           "exclusive_scan");
    unsigned long long nev_h = 0;
    PD_TRY(hipMemcpyAsync(&h->n_slots, h->eptr + np, 8, hipMemcpyDeviceToHost, s), "copy");
    PD_TRY(hipMemcpyAsync(&h->n_recs, h->rptr + np, 8, hipMemcpyDeviceToHost, s), "copy");
    PD_TRY(hipMemcpyAsync(&nev_h, h->nev, 8, hipMemcpyDeviceToHost, s), "copy");
    PD_TRY(hipMemcpyAsync(&err_h, h->err, sizeof(int), hipMemcpyDeviceToHost, s), "copy");
    PD_TRY(hipStreamSynchronize(s), "hipStreamSynchronize");
    if (err_h) { rc = fs_fail_(FS_EINVAL, patch_error(err_h), err_h); return bail(rc); }
    (void)hipFree(tmp); (void)hipFree(pk_in); (void)hipFree(pk_out); (void)hipFree(pv_in); (void)hipFree(flag); (void)hipFree(scalars);
    h->n_evals = (int64_t)nev_h;
    *plan = h;
    *n_patches = np;
    *n_elem_slots = h->n_slots;
    *n_recs = h->n_recs;
    if (n_evals) *n_evals = h->n_evals;
    return FS_OK;
#undef PD_TRY
}

int fs_fused_plan_device_finish(fs_fused_plan_dev *h, int64_t *pat_eptr, int32_t *pat_elems, int64_t *pat_rptr,
                                fs_fused_rec *recs, void *stream) {
    if (!h) return fs_fail_(FS_EINVAL, "fs_fused_plan_device_finish: null handle", 0);
    if (!pat_eptr || !pat_rptr || (h->n_slots > 0 && !pat_elems) || (h->n_recs > 0 && !recs)) {
        release(h);
        return fs_fail_(FS_EINVAL, "fs_fused_plan_device_finish: null output", 0);
    }
    hipStream_t s = (hipStream_t)stream;
    int rc = FS_OK, err_h = 0;
    hipError_t e = hipMemcpyAsync(pat_eptr, h->eptr, (size_t)(h->np + 1) * 8, hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(pat_rptr, h->rptr, (size_t)(h->np + 1) * 8, hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess && h->np > 0) {
        const unsigned grid = (unsigned)(h->np < (1 << 20) ? h->np : (1 << 20));
        hipLaunchKernelGGL(fused_plan_patch_kernel<true>, dim3(grid), dim3(PB_BS), 0, s, h->conn, h->nbr_ptr, h->nbr, h->n2e_ptr,
                           h->n2e, h->order, h->pstart, h->np, (int)h->max_patch_elems, (int64_t *)nullptr, (int64_t *)nullptr,
                           h->nev, h->err, (const int64_t *)h->eptr, (const int64_t *)h->rptr, pat_elems, recs);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&err_h, h->err, sizeof(int), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);        // the handle's buffers are freed below
    if (e != hipSuccess) rc = fs_fail_(FS_EHIP, "fs_fused_plan_device_finish", (int)e);
    else if (err_h) rc = fs_fail_(FS_EINVAL, patch_error(err_h), err_h);
    release(h);
    return rc;
}

void fs_fused_plan_device_free(fs_fused_plan_dev *h) { release(h); }

}  // extern "C"
