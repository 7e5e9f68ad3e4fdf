// femshell_symbolic.hip -- device symbolic phase of the CSR assembly (MI355X / gfx950).
//
// Replaces MeshAssembler._precompute_elements' DOF map + _compute_sparsity_pattern
// (core/assembler.py:166-187, 222-241 of the reference: per-row Python sets) by a sort/unique of
// node-pair keys (I * N + J) on the GPU.  For meshes whose nodes all carry d DOFs the pattern of row
// d*I+k is {d*J+l : J neighbour of I}, so node-pair blocks describe it completely.
//
//   begin : one (key, source) entry per local node pair of every element, written in ELEMENT ORDER;
//           a STABLE radix sort (rocPRIM) by key therefore keeps, inside each block, the reference's
//           insertion order (core/assembler.py:295-300); count blocks and contributions.
//   finish: block rows / neighbours / contribution lists / block descriptors (fs_blk_desc).
// Integer work, HBM-bound; the sort dominates (2 x 16 B per entry and radix pass).
#include <string.h>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <stdint.h>
#include <stdio.h>

#include "femshell_hip.h"
#include "femshell_internal.h"

struct fs_symbolic {
    int64_t n = 0;          // entries (contributions + pattern-only extras)
    int64_t n_nodes = 0;
    int64_t nblk = 0, ncontrib = 0;
    int64_t *keys = nullptr, *vals = nullptr;   // sorted
    int64_t *bidx = nullptr, *rpos = nullptr;   // exclusive scans of (first-of-block) and (is-contribution)
    void *tmp = nullptr;
};

namespace {

#define FS_HIP(call, what)                                               \
    do {                                                                 \
        hipError_t e__ = (call);                                         \
        if (e__ != hipSuccess) return fs_fail_(FS_EHIP, what, (int)e__); \
    } while (0)

inline unsigned grid_of(int64_t n, int block) { return (unsigned)((n + block - 1) / block); }

__global__ __launch_bounds__(256) void sym_keys_kernel(const int64_t *__restrict__ conn,
                                                       const int64_t *__restrict__ entry_ptr,
                                                       const int64_t *__restrict__ src_base, int64_t E,
                                                       int nnmax, int64_t n_nodes, int tiled,
                                                       int64_t *__restrict__ keys, int64_t *__restrict__ vals,
                                                       int *__restrict__ bad) {
    const int per = nnmax * nnmax;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= E * per) return;
    const int64_t e = t / per;
    const int ab = (int)(t - e * per);
    const int a = ab / nnmax, b = ab - a * nnmax;
    const int64_t *row = conn + e * nnmax;
    int nn = 0;
    for (int c = 0; c < nnmax; c++) nn += (row[c] >= 0);     // -1 padding trails the node list
    if (!entry_ptr && nn != nnmax) { *bad = 2; return; }   // ragged rows need entry_ptr / src_base
    if (a >= nn || b >= nn) return;
    const int64_t I = row[a], J = row[b];
    if (I < 0 || J < 0 || I >= n_nodes || J >= n_nodes) { *bad = 1; return; }
    const int64_t pos = (entry_ptr ? entry_ptr[e] : e * per) + a * nn + b;
    const int i = a < b ? a : b, j = a < b ? b : a;
    const int p = i * nn - (i * (i - 1)) / 2 + (j - i);       // FS_LAYOUT_SYMBLK pair index
    const int64_t base = src_base ? src_base[e] : e * (int64_t)(nn * (nn + 1) / 2);
    int64_t src = base + p;
    if (tiled) {   // FS_LAYOUT_SYMBLK_T16 (uniform quads): same formula as blk_t16() of femshell_hip.hip
        const int rg = (int)((0x3629517840ull >> (4 * p)) & 15);
        const int round = rg >> 2, g = rg & 3, Q = (int)(e & 15);
        src = (e >> 4) * 160 + ((round < 2) ? (round * 64 + ((Q >> 3) << 5) + (g << 3) + (Q & 7)) : (128 + (g << 4) + Q));
    }
    keys[pos] = I * n_nodes + J;
    vals[pos] = (src << 1) | (a > b ? 1 : 0);
}

__global__ __launch_bounds__(256) void sym_extra_kernel(const int64_t *__restrict__ extra, int64_t n_extra,
                                                        int64_t *__restrict__ keys, int64_t *__restrict__ vals) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_extra) return;
    keys[t] = extra[t];
    vals[t] = -1;      // pattern-only entry (ghost column): sorts after the contributions of its block
}

__global__ __launch_bounds__(256) void sym_flags_kernel(const int64_t *__restrict__ keys,
                                                        const int64_t *__restrict__ vals, int64_t n,
                                                        int64_t *__restrict__ first, int64_t *__restrict__ real) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    first[t] = (t == 0 || keys[t] != keys[t - 1]) ? 1 : 0;
    real[t] = vals[t] >= 0 ? 1 : 0;
}

__global__ __launch_bounds__(256) void sym_scatter_kernel(const int64_t *__restrict__ keys,
                                                          const int64_t *__restrict__ vals,
                                                          const int64_t *__restrict__ bidx,
                                                          const int64_t *__restrict__ rpos, int64_t n,
                                                          int64_t n_nodes, int64_t nblk, int64_t ncontrib,
                                                          int32_t *__restrict__ blk_row, int32_t *__restrict__ nbr,
                                                          int64_t *__restrict__ blk_ptr,
                                                          int64_t *__restrict__ contrib) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int64_t k = keys[t];
    if (t == 0 || k != keys[t - 1]) {
        const int64_t b = bidx[t];
        blk_row[b] = (int32_t)(k / n_nodes);
        nbr[b] = (int32_t)(k % n_nodes);
        blk_ptr[b] = rpos[t];
    }
    const int64_t v = vals[t];
    if (v >= 0) contrib[rpos[t]] = v;
    if (t == n - 1) blk_ptr[nblk] = ncontrib;
}

__global__ __launch_bounds__(256) void sym_rowptr_kernel(const int32_t *__restrict__ blk_row, int64_t nblk,
                                                         int64_t n_nodes, int64_t *__restrict__ nbr_ptr) {
    const int64_t I = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (I > n_nodes) return;
    int64_t lo = 0, hi = nblk;                 // first block whose row is >= I
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (blk_row[mid] < I) lo = mid + 1; else hi = mid;
    }
    nbr_ptr[I] = lo;
}

__global__ __launch_bounds__(256) void sym_desc_kernel(const int64_t *__restrict__ nbr_ptr,
                                                       const int32_t *__restrict__ blk_row,
                                                       const int64_t *__restrict__ blk_ptr,
                                                       const int64_t *__restrict__ contrib, int64_t nblk, int d,
                                                       fs_blk_desc *__restrict__ desc) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblk) return;
    const int64_t I = blk_row[b];
    const int64_t r0 = nbr_ptr[I], m = nbr_ptr[I + 1] - r0;
    fs_blk_desc out;
    out.dst = (int64_t)d * d * r0 + (int64_t)d * (b - r0);
    out.stride = (int32_t)(d * m);
    const int64_t lo = blk_ptr[b], cnt = blk_ptr[b + 1] - lo;
    out.cnt = (int32_t)cnt;
    for (int c = 0; c < 4; c++) out.c[c] = 0;
    if (cnt <= 4) {
        for (int c = 0; c < 4; c++)
            if (c < cnt) out.c[c] = contrib[lo + c];
    } else {
        out.c[0] = lo;
        if (cnt <= 8) {          // FS_BLK_PACKED: the codes inline as uint32[8] when they all fit
            int64_t code[8];
            bool fits = true;
            for (int c = 0; c < 8; c++) {
                code[c] = (c < cnt) ? contrib[lo + c] : 0;
                fits = fits && code[c] >= 0 && code[c] < (int64_t(1) << 32);
            }
            if (fits) {
                for (int c = 0; c < 4; c++) out.c[c] = code[2 * c] | (code[2 * c + 1] << 32);
                out.cnt = (int32_t)cnt | FS_BLK_PACKED;
            }
        }
    }
    desc[b] = out;
}

void release(fs_symbolic *h) {
    if (!h) return;
    (void)hipFree(h->keys); (void)hipFree(h->vals); (void)hipFree(h->bidx); (void)hipFree(h->rpos);
    (void)hipFree(h->tmp);
    delete h;
}

}  // namespace

extern "C" {

int fs_csr_symbolic_begin(const int64_t *conn, const int64_t *entry_ptr, const int64_t *src_base, int64_t E,
                          int32_t nnmax, uint32_t layout, int64_t n_entries, int64_t n_nodes, const int64_t *extra_keys,
                          int64_t n_extra, fs_symbolic **handle, int64_t *nblk, int64_t *ncontrib, void *stream) {
    if (!handle || !nblk || !ncontrib) return fs_fail_(FS_EINVAL, "fs_csr_symbolic_begin: null output", 0);
    *handle = nullptr; *nblk = 0; *ncontrib = 0;
    if (E < 0 || nnmax <= 0 || nnmax > 32 || n_nodes <= 0 || n_extra < 0 || n_entries < 0)
        return fs_fail_(FS_EINVAL, "fs_csr_symbolic_begin: bad sizes", 0);
    if (E > 0 && !conn) return fs_fail_(FS_EINVAL, "fs_csr_symbolic_begin: conn is null", 0);
    if (n_extra > 0 && !extra_keys) return fs_fail_(FS_EINVAL, "fs_csr_symbolic_begin: extra_keys is null", 0);
    if (layout != FS_LAYOUT_SYMBLK && !(layout == FS_LAYOUT_SYMBLK_T16 && nnmax == 4 && !entry_ptr && !src_base))
        return fs_fail_(FS_EINVAL, "fs_csr_symbolic_begin: layout must be FS_LAYOUT_SYMBLK, or _T16 for uniform quads", 0);
    if (!entry_ptr && n_entries != E * nnmax * nnmax)
        return fs_fail_(FS_EINVAL, "fs_csr_symbolic_begin: n_entries must be E*nnmax^2 without entry_ptr", 0);
    if ((double)n_nodes * (double)n_nodes >= 9.0e18)
        return fs_fail_(FS_EINVAL, "fs_csr_symbolic_begin: n_nodes^2 overflows the 63-bit key", 0);
    hipStream_t s = (hipStream_t)stream;
    fs_symbolic *h = new fs_symbolic;
    h->n = n_entries + n_extra;
    h->n_nodes = n_nodes;
    *handle = h;
    if (h->n == 0) return FS_OK;
    const size_t bytes = (size_t)h->n * sizeof(int64_t);
    int64_t *kin = nullptr, *vin = nullptr;
    int *bad = nullptr;
    int rc = FS_OK;
    auto bail = [&](int code) {
        (void)hipFree(kin); (void)hipFree(vin); (void)hipFree(bad);
        release(h); *handle = nullptr;
        return code;
    };
#define SYM_TRY(call, what)                                                              \
    do {                                                                                 \
        hipError_t e__ = (call);                                                         \
        if (e__ != hipSuccess) { rc = fs_fail_(FS_EHIP, what, (int)e__); return bail(rc); } \
    } while (0)
    SYM_TRY(hipMalloc(&kin, bytes), "hipMalloc keys");
    SYM_TRY(hipMalloc(&vin, bytes), "hipMalloc vals");
    SYM_TRY(hipMalloc(&h->keys, bytes), "hipMalloc keys");
    SYM_TRY(hipMalloc(&h->vals, bytes), "hipMalloc vals");
    SYM_TRY(hipMalloc(&bad, sizeof(int)), "hipMalloc flag");
    SYM_TRY(hipMemsetAsync(bad, 0, sizeof(int), s), "hipMemsetAsync");
    if (E > 0) {
        hipLaunchKernelGGL(sym_keys_kernel, dim3(grid_of(E * nnmax * nnmax, 256)), dim3(256), 0, s, conn, entry_ptr,
                           src_base, E, (int)nnmax, n_nodes, layout == FS_LAYOUT_SYMBLK_T16 ? 1 : 0, kin, vin, bad);
        SYM_TRY(hipGetLastError(), "sym_keys_kernel");
    }
    if (n_extra > 0) {
        hipLaunchKernelGGL(sym_extra_kernel, dim3(grid_of(n_extra, 256)), dim3(256), 0, s, extra_keys, n_extra,
                           kin + n_entries, vin + n_entries);
        SYM_TRY(hipGetLastError(), "sym_extra_kernel");
    }
    int bits = 1;
    while (bits < 63 && ((int64_t)1 << bits) < n_nodes * n_nodes) bits++;
    size_t tmp_bytes = 0;
    SYM_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, kin, h->keys, vin, h->vals, (size_t)h->n, 0, bits, s),
            "radix_sort_pairs (size)");
    SYM_TRY(hipMalloc(&h->tmp, tmp_bytes ? tmp_bytes : 8), "hipMalloc sort workspace");
    SYM_TRY(rocprim::radix_sort_pairs(h->tmp, tmp_bytes, kin, h->keys, vin, h->vals, (size_t)h->n, 0, bits, s),
            "radix_sort_pairs");
    (void)hipFree(h->tmp); h->tmp = nullptr;     // hipFree synchronises: the sort has finished
    // first-of-block and is-contribution flags -> exclusive scans (the unsorted buffers are reused)
    h->bidx = kin; h->rpos = vin; kin = vin = nullptr;
    hipLaunchKernelGGL(sym_flags_kernel, dim3(grid_of(h->n, 256)), dim3(256), 0, s, h->keys, h->vals, h->n, h->bidx,
                       h->rpos);
    SYM_TRY(hipGetLastError(), "sym_flags_kernel");
    int64_t last[2] = {0, 0}, tot[2] = {0, 0};
    SYM_TRY(hipMemcpyAsync(&last[0], h->bidx + h->n - 1, sizeof(int64_t), hipMemcpyDeviceToHost, s), "copy");
    SYM_TRY(hipMemcpyAsync(&last[1], h->rpos + h->n - 1, sizeof(int64_t), hipMemcpyDeviceToHost, s), "copy");
    size_t scan_bytes = 0;
    SYM_TRY(rocprim::exclusive_scan(nullptr, scan_bytes, h->bidx, h->bidx, (int64_t)0, (size_t)h->n,
                                    rocprim::plus<int64_t>(), s), "exclusive_scan (size)");
    SYM_TRY(hipMalloc(&h->tmp, scan_bytes ? scan_bytes : 8), "hipMalloc scan workspace");
    SYM_TRY(rocprim::exclusive_scan(h->tmp, scan_bytes, h->bidx, h->bidx, (int64_t)0, (size_t)h->n,
                                    rocprim::plus<int64_t>(), s), "exclusive_scan");
    SYM_TRY(rocprim::exclusive_scan(h->tmp, scan_bytes, h->rpos, h->rpos, (int64_t)0, (size_t)h->n,
                                    rocprim::plus<int64_t>(), s), "exclusive_scan");
    SYM_TRY(hipMemcpyAsync(&tot[0], h->bidx + h->n - 1, sizeof(int64_t), hipMemcpyDeviceToHost, s), "copy");
    SYM_TRY(hipMemcpyAsync(&tot[1], h->rpos + h->n - 1, sizeof(int64_t), hipMemcpyDeviceToHost, s), "copy");
    int bad_h = 0;
    SYM_TRY(hipMemcpyAsync(&bad_h, bad, sizeof(int), hipMemcpyDeviceToHost, s), "copy");
    SYM_TRY(hipStreamSynchronize(s), "hipStreamSynchronize");
    (void)hipFree(bad); bad = nullptr;
    (void)hipFree(h->tmp); h->tmp = nullptr;
    if (bad_h) {
        rc = fs_fail_(FS_EINVAL, bad_h == 2 ? "fs_csr_symbolic_begin: -1-padded connectivity needs entry_ptr and src_base"
                                            : "fs_csr_symbolic_begin: connectivity refers to nodes outside [0, n_nodes)", 0);
        return bail(rc);
    }
    h->nblk = tot[0] + last[0];
    h->ncontrib = tot[1] + last[1];
    *nblk = h->nblk;
    *ncontrib = h->ncontrib;
    return FS_OK;
#undef SYM_TRY
}

int fs_csr_symbolic_finish(fs_symbolic *h, int32_t d, int64_t *nbr_ptr, int32_t *nbr, int32_t *blk_row,
                           int64_t *blk_ptr, int64_t *contrib, fs_blk_desc *desc, void *stream) {
    if (!h) return fs_fail_(FS_EINVAL, "fs_csr_symbolic_finish: null handle", 0);
    if (d <= 0 || !nbr_ptr || !blk_ptr || (h->nblk > 0 && (!nbr || !blk_row)) || (h->ncontrib > 0 && !contrib)) {
        release(h);
        return fs_fail_(FS_EINVAL, "fs_csr_symbolic_finish: null output or bad d", 0);
    }
    hipStream_t s = (hipStream_t)stream;
    int rc = FS_OK;
    hipError_t e = hipSuccess;
    if (h->n > 0) {
        hipLaunchKernelGGL(sym_scatter_kernel, dim3(grid_of(h->n, 256)), dim3(256), 0, s, h->keys, h->vals, h->bidx,
                           h->rpos, h->n, h->n_nodes, h->nblk, h->ncontrib, blk_row, nbr, blk_ptr, contrib);
    } else {
        e = hipMemsetAsync(blk_ptr, 0, sizeof(int64_t), s);
    }
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) {
        hipLaunchKernelGGL(sym_rowptr_kernel, dim3(grid_of(h->n_nodes + 1, 256)), dim3(256), 0, s, blk_row, h->nblk,
                           h->n_nodes, nbr_ptr);
        e = hipGetLastError();
    }
    if (e == hipSuccess && desc && h->nblk > 0) {
        hipLaunchKernelGGL(sym_desc_kernel, dim3(grid_of(h->nblk, 256)), dim3(256), 0, s, nbr_ptr, blk_row, blk_ptr,
                           contrib, h->nblk, (int)d, desc);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);   // the handle's buffers are freed below
    if (e != hipSuccess) rc = fs_fail_(FS_EHIP, "fs_csr_symbolic_finish", (int)e);
    release(h);
    return rc;
}

void fs_csr_symbolic_free(fs_symbolic *h) { release(h); }

}  // extern "C"
