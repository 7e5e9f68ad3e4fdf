// femshell_sparse.hip -- CSR post-assembly operations on the device (MI355X / gfx950): the step right
// after the assembly path.
//
//   fs_csr_zero_rows        BoundaryConditionManager._apply_bc_to_system (core/bc.py:157-175):
//                           K.zeroRows(fixed, 1.0, U_fixed, F) / M.zeroRows(fixed, 0.0)
//   fs_csr_reduce_symbolic  free x free sub-matrix pattern (createSubMatrix(free_is, free_is),
//   fs_csr_reduce_fill      core/bc.py:178-196) + the load correction F_free - K_free,fixed u_fixed
//                           (_adjusted_load_vector, core/bc.py:198-239)
//   fs_expand_solution      expand_solution (core/bc.py:241-290)
//   fs_csr_row_sums         lump_mass_matrix: M.getRowSum (solvers/fsi/linear_dynamic.py:227-234)
// Byte / index work bounded by HBM; one 64-lane wave per matrix row (rows hold 18..200 entries).
#include <string.h>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_scan.hpp>
#include <stdint.h>

#include "femshell_hip.h"
#include "femshell_internal.h"

namespace {

inline unsigned grid_of(int64_t n, int block) { return (unsigned)((n + block - 1) / block); }
constexpr int WPB = 4;   // waves (rows) per 256-thread workgroup

__global__ __launch_bounds__(256) void zero_rows_kernel(const int64_t *__restrict__ indptr,
                                                        const int32_t *__restrict__ indices,
                                                        double *__restrict__ data,
                                                        const int64_t *__restrict__ rows, int64_t n_rows,
                                                        double diag, const double *__restrict__ x,
                                                        double *__restrict__ b) {
    const int lane = threadIdx.x & 63;
    const int64_t t = (int64_t)blockIdx.x * WPB + (threadIdx.x >> 6);
    if (t >= n_rows) return;
    const int64_t r = rows[t];
    const int64_t lo = indptr[r], hi = indptr[r + 1];
    for (int64_t q = lo + lane; q < hi; q += 64) data[q] = (indices[q] == r) ? diag : 0.0;
    if (lane == 0 && b && x) b[r] = diag * x[r];
}

__global__ __launch_bounds__(256) void free_flag_kernel(const uint8_t *__restrict__ fixed_mask, int64_t ndof,
                                                        int64_t *__restrict__ flag) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < ndof) flag[t] = fixed_mask[t] ? 0 : 1;
}

__global__ __launch_bounds__(256) void free_rank_kernel(const uint8_t *__restrict__ fixed_mask,
                                                        const int64_t *__restrict__ scan, int64_t ndof,
                                                        int32_t *__restrict__ free_rank) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < ndof) free_rank[t] = fixed_mask[t] ? -1 : (int32_t)scan[t];
}

// counts[rank(r)] = number of free columns in free row r
__global__ __launch_bounds__(256) void reduce_count_kernel(const int64_t *__restrict__ indptr,
                                                           const int32_t *__restrict__ indices, int64_t ndof,
                                                           const int32_t *__restrict__ free_rank,
                                                           int64_t *__restrict__ counts) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * WPB + (threadIdx.x >> 6);
    if (r >= ndof) return;
    const int32_t rr = free_rank[r];
    if (rr < 0) return;
    const int64_t lo = indptr[r], hi = indptr[r + 1];
    int c = 0;
    for (int64_t q = lo + lane; q < hi; q += 64) c += (free_rank[indices[q]] >= 0);
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if (lane == 0) counts[rr] = c;
}

__global__ __launch_bounds__(256) void reduce_fill_kernel(const int64_t *__restrict__ indptr,
                                                          const int32_t *__restrict__ indices,
                                                          const double *__restrict__ data, int64_t ndof,
                                                          const int32_t *__restrict__ free_rank,
                                                          const int64_t *__restrict__ red_indptr,
                                                          int32_t *__restrict__ red_indices,
                                                          double *__restrict__ red_data,
                                                          const double *__restrict__ u_fixed,
                                                          const double *__restrict__ f,
                                                          double *__restrict__ f_red) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * WPB + (threadIdx.x >> 6);
    if (r >= ndof) return;
    const int32_t rr = free_rank[r];
    if (rr < 0) return;
    const int64_t lo = indptr[r], hi = indptr[r + 1];
    int64_t out = red_indptr[rr];
    double corr = 0.0;
    for (int64_t q0 = lo; q0 < hi; q0 += 64) {          // uniform trip count: the ballot needs all lanes
        const int64_t q = q0 + lane;
        const bool in = q < hi;
        const int32_t c = in ? indices[q] : 0;
        const int32_t cr = in ? free_rank[c] : -1;
        const bool keep = in && cr >= 0;
        const unsigned long long m = __ballot(keep);
        if (keep) {
            const int64_t dst = out + __popcll(m & ((1ull << lane) - 1ull));
            if (red_indices) red_indices[dst] = cr;
            if (red_data) red_data[dst] = data[q];
        }
        if (in && cr < 0 && u_fixed && data) corr += data[q] * u_fixed[c];
        out += __popcll(m);
    }
    if (f_red) {
        for (int o = 32; o > 0; o >>= 1) corr += __shfl_xor(corr, o);
        if (lane == 0) f_red[rr] = (f ? f[r] : 0.0) - corr;
    }
}

__global__ __launch_bounds__(256) void expand_kernel(const double *__restrict__ u_red,
                                                     const int32_t *__restrict__ free_rank,
                                                     const double *__restrict__ u_fixed, int64_t ndof,
                                                     double *__restrict__ u_full) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ndof) return;
    const int32_t rr = free_rank[t];
    u_full[t] = rr >= 0 ? u_red[rr] : (u_fixed ? u_fixed[t] : 0.0);
}

__global__ __launch_bounds__(256) void row_sums_kernel(const int64_t *__restrict__ indptr,
                                                       const double *__restrict__ data, int64_t nrows,
                                                       double *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * WPB + (threadIdx.x >> 6);
    if (r >= nrows) return;
    const int64_t lo = indptr[r], hi = indptr[r + 1];
    double s = 0.0;
    for (int64_t q = lo + lane; q < hi; q += 64) s += data[q];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) out[r] = s;
}

// Inertial (fictitious) nodal forces of a rotating frame (solvers/fsi/corotational.py:347-469), one thread per node:
//   centrifugal  m w^2 (r - (r.a) a)      Coriolis  -2 m (w a) x v      Euler  -m (alpha a) x r
// summed in that order like compute_all_inertial_forces (:511-552).  Products in the reference's order
// (contraction is off for this translation unit's arithmetic: -ffp-contract=off).
__global__ __launch_bounds__(256) void inertial_forces_kernel(const double *__restrict__ coords,
                                                              const double *__restrict__ vel,
                                                              const double *__restrict__ mass, int64_t N,
                                                              double omega, double alpha, double ax, double ay,
                                                              double az, double cx, double cy, double cz,
                                                              unsigned parts, double *__restrict__ F) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const double m = mass[n];
    const double rx = coords[3 * n] - cx, ry = coords[3 * n + 1] - cy, rz = coords[3 * n + 2] - cz;
    double fx = 0.0, fy = 0.0, fz = 0.0;
    if ((parts & 1u) && omega != 0.0) {
        const double rd = (rx * ax + ry * ay) + rz * az;                 // r @ axis
        const double w2 = omega * omega;
        fx += (m * w2) * (rx - rd * ax); fy += (m * w2) * (ry - rd * ay); fz += (m * w2) * (rz - rd * az);
    }
    if ((parts & 2u) && omega != 0.0) {
        const double wx = omega * ax, wy = omega * ay, wz = omega * az;
        const double vx = vel[3 * n], vy = vel[3 * n + 1], vz = vel[3 * n + 2];
        fx += (-2.0 * m) * (wy * vz - wz * vy); fy += (-2.0 * m) * (wz * vx - wx * vz); fz += (-2.0 * m) * (wx * vy - wy * vx);
    }
    if ((parts & 4u) && alpha != 0.0) {
        const double bx = alpha * ax, by = alpha * ay, bz = alpha * az;
        fx += (-m) * (by * rz - bz * ry); fy += (-m) * (bz * rx - bx * rz); fz += (-m) * (bx * ry - by * rx);
    }
    F[3 * n] = fx; F[3 * n + 1] = fy; F[3 * n + 2] = fz;
}

// Diagnostic: fp64 FMA issue rate, 8 independent dependent-chains per lane (bench.py measures the box's own fp64
// ceiling with it, at one and at four waves per SIMD).
__global__ __launch_bounds__(64) void fp64_fma_kernel(double *__restrict__ out, int iters, double a, double b) {
    double acc[8];
    for (int c = 0; c < 8; c++) acc[c] = (double)(threadIdx.x + c);
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int c = 0; c < 8; c++) acc[c] = __builtin_fma(acc[c], a, b);
    }
    double s = 0.0;
    for (int c = 0; c < 8; c++) s += acc[c];
    out[(int64_t)blockIdx.x * 64 + threadIdx.x] = s;
}

// y = A x (+ y0) for the block-structured CSR of fs_csr_pattern_blk (D dofs per node: the D rows of node I are
// contiguous, row r holds the D-entry pieces of the node's blocks one after the other).  A lane owns a node-pair BLOCK:
// its D row pieces (neighbouring lanes read neighbouring pieces of every row: whole lines), x of its column node once,
// D dot products to LDS; then a lane per matrix row adds the partial products of its blocks from left to right --
// one writer per row, the same sum in every run.
constexpr int MV_NB = 28;      // nodes per workgroup (28 x 9 blocks = 252 lanes on a structured quad mesh)
constexpr int MV_CAP = 512;    // blocks per pass through LDS
template <int D>
__global__ __launch_bounds__(256) void blk_matvec_kernel(const int64_t *__restrict__ nbr_ptr, const int32_t *__restrict__ nbr,
                                                         const int32_t *__restrict__ blk_row, int64_t n_nodes,
                                                         const double *__restrict__ data, const double *__restrict__ x,
                                                         const double *__restrict__ y0, double *__restrict__ y) {
    __shared__ double s_p[D * MV_CAP];
    const int tid = threadIdx.x;
    const int64_t n0 = (int64_t)blockIdx.x * MV_NB;
    const int64_t n1 = (n0 + MV_NB < n_nodes) ? n0 + MV_NB : n_nodes;
    const int64_t T0 = nbr_ptr[n0], T1 = nbr_ptr[n1];
    // phase-B role of this lane: row r of node In
    const int64_t In = n0 + tid / D;
    const int r_own = tid % D;
    const bool owner = In < n1;
    int64_t b0 = 0, b1 = 0;
    if (owner) { b0 = nbr_ptr[In]; b1 = nbr_ptr[In + 1]; }
    double acc = 0.0;
    for (int64_t t0 = T0; t0 < T1; t0 += MV_CAP) {
        const int64_t t1 = (t0 + MV_CAP < T1) ? t0 + MV_CAP : T1;
        for (int64_t t = t0 + tid; t < t1; t += 256) {
            const int64_t I = blk_row[t], J = nbr[t];
            const int64_t f0 = nbr_ptr[I];
            const int64_t L = (nbr_ptr[I + 1] - f0) * D;
            const double *a = data + f0 * (D * D) + (t - f0) * D;
            double xv[D], av[D][D];
            if constexpr ((D & 1) == 0) {
                for (int c = 0; c < D / 2; c++) {
                    const double2 v = reinterpret_cast<const double2 *>(x + J * D)[c];
                    xv[2 * c] = v.x; xv[2 * c + 1] = v.y;
                }
                for (int r = 0; r < D; r++)
                    for (int c = 0; c < D / 2; c++) {
                        const double2 v = reinterpret_cast<const double2 *>(a + r * L)[c];
                        av[r][2 * c] = v.x; av[r][2 * c + 1] = v.y;
                    }
            } else {
                for (int c = 0; c < D; c++) xv[c] = x[J * D + c];
                for (int r = 0; r < D; r++)
                    for (int c = 0; c < D; c++) av[r][c] = a[r * L + c];
            }
            for (int r = 0; r < D; r++) {
                double v = av[r][0] * xv[0];
                for (int c = 1; c < D; c++) v = __builtin_fma(av[r][c], xv[c], v);
                s_p[r * MV_CAP + (int)(t - t0)] = v;
            }
        }
        __syncthreads();
        if (owner) {
            const int64_t lo = (b0 > t0) ? b0 : t0, hi = (b1 < t1) ? b1 : t1;
            for (int64_t t = lo; t < hi; t++) acc += s_p[r_own * MV_CAP + (int)(t - t0)];
        }
        __syncthreads();
    }
    if (owner) {
        const int64_t row = In * D + r_own;
        y[row] = y0 ? acc + y0[row] : acc;
    }
}

}  // namespace

extern "C" {

int fs_csr_zero_rows(const int64_t *indptr, const int32_t *indices, double *data, const int64_t *rows,
                     int64_t n_rows, double diag, const double *x, double *b, void *stream) {
    if (n_rows <= 0) return n_rows == 0 ? FS_OK : fs_fail_(FS_EINVAL, "fs_csr_zero_rows: n_rows < 0", 0);
    if (!indptr || !indices || !data || !rows) return fs_fail_(FS_EINVAL, "fs_csr_zero_rows: null argument", 0);
    hipLaunchKernelGGL(zero_rows_kernel, dim3(grid_of(n_rows, WPB)), dim3(256), 0, (hipStream_t)stream, indptr,
                       indices, data, rows, n_rows, diag, x, b);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? FS_OK : fs_fail_(FS_EHIP, "zero_rows_kernel", (int)e);
}

int fs_csr_reduce_symbolic(const int64_t *indptr, const int32_t *indices, int64_t ndof,
                           const uint8_t *fixed_mask, int32_t *free_rank, int64_t *red_indptr,
                           int64_t *n_free, int64_t *nnz_red, void *stream) {
    if (!n_free || !nnz_red) return fs_fail_(FS_EINVAL, "fs_csr_reduce_symbolic: null output", 0);
    *n_free = 0; *nnz_red = 0;
    if (ndof < 0) return fs_fail_(FS_EINVAL, "fs_csr_reduce_symbolic: ndof < 0", 0);
    if (!red_indptr) return fs_fail_(FS_EINVAL, "fs_csr_reduce_symbolic: null argument", 0);
    hipStream_t s = (hipStream_t)stream;
    if (ndof == 0) {
        hipError_t e0 = hipMemsetAsync(red_indptr, 0, sizeof(int64_t), s);
        return e0 == hipSuccess ? FS_OK : fs_fail_(FS_EHIP, "hipMemsetAsync", (int)e0);
    }
    if (!indptr || !indices || !fixed_mask || !free_rank)
        return fs_fail_(FS_EINVAL, "fs_csr_reduce_symbolic: null argument", 0);
    int64_t *scan = nullptr;
    void *tmp = nullptr;
    hipError_t e = hipMalloc(&scan, (size_t)(ndof + 1) * sizeof(int64_t));
    size_t tmp_bytes = 0;
    int64_t last_flag = 0, last_scan = 0, last_cnt = 0, last_ptr = 0;
    auto done = [&](const char *what) {
        (void)hipFree(scan); (void)hipFree(tmp);
        return e == hipSuccess ? FS_OK : fs_fail_(FS_EHIP, what, (int)e);
    };
    if (e != hipSuccess) return done("hipMalloc");
    hipLaunchKernelGGL(free_flag_kernel, dim3(grid_of(ndof, 256)), dim3(256), 0, s, fixed_mask, ndof, scan);
    if ((e = hipGetLastError()) != hipSuccess) return done("free_flag_kernel");
    if ((e = hipMemcpyAsync(&last_flag, scan + ndof - 1, 8, hipMemcpyDeviceToHost, s)) != hipSuccess) return done("copy");
    if ((e = rocprim::exclusive_scan(nullptr, tmp_bytes, scan, scan, (int64_t)0, (size_t)ndof,
                                     rocprim::plus<int64_t>(), s)) != hipSuccess) return done("exclusive_scan (size)");
    if ((e = hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 8)) != hipSuccess) return done("hipMalloc");
    if ((e = rocprim::exclusive_scan(tmp, tmp_bytes, scan, scan, (int64_t)0, (size_t)ndof,
                                     rocprim::plus<int64_t>(), s)) != hipSuccess) return done("exclusive_scan");
    if ((e = hipMemcpyAsync(&last_scan, scan + ndof - 1, 8, hipMemcpyDeviceToHost, s)) != hipSuccess) return done("copy");
    hipLaunchKernelGGL(free_rank_kernel, dim3(grid_of(ndof, 256)), dim3(256), 0, s, fixed_mask, scan, ndof, free_rank);
    if ((e = hipGetLastError()) != hipSuccess) return done("free_rank_kernel");
    if ((e = hipStreamSynchronize(s)) != hipSuccess) return done("hipStreamSynchronize");
    const int64_t nf = last_scan + last_flag;
    *n_free = nf;
    if (nf == 0) {
        e = hipMemsetAsync(red_indptr, 0, sizeof(int64_t), s);
        return done("hipMemsetAsync");
    }
    // per-row counts into red_indptr[0..nf), then an exclusive scan in place; red_indptr[nf] = total
    hipLaunchKernelGGL(reduce_count_kernel, dim3(grid_of(ndof, WPB)), dim3(256), 0, s, indptr, indices, ndof,
                       free_rank, red_indptr);
    if ((e = hipGetLastError()) != hipSuccess) return done("reduce_count_kernel");
    if ((e = hipMemcpyAsync(&last_cnt, red_indptr + nf - 1, 8, hipMemcpyDeviceToHost, s)) != hipSuccess) return done("copy");
    if ((e = rocprim::exclusive_scan(tmp, tmp_bytes, red_indptr, red_indptr, (int64_t)0, (size_t)nf,
                                     rocprim::plus<int64_t>(), s)) != hipSuccess) return done("exclusive_scan");
    if ((e = hipMemcpyAsync(&last_ptr, red_indptr + nf - 1, 8, hipMemcpyDeviceToHost, s)) != hipSuccess) return done("copy");
    if ((e = hipStreamSynchronize(s)) != hipSuccess) return done("hipStreamSynchronize");
    const int64_t total = last_ptr + last_cnt;
    *nnz_red = total;
    e = hipMemcpyAsync(red_indptr + nf, &total, 8, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);     // `total` lives on this stack frame
    return done("copy");
}

int fs_csr_reduce_fill(const int64_t *indptr, const int32_t *indices, const double *data, int64_t ndof,
                       const int32_t *free_rank, const int64_t *red_indptr, int32_t *red_indices,
                       double *red_data, const double *u_fixed, const double *f, double *f_red, void *stream) {
    if (ndof <= 0) return ndof == 0 ? FS_OK : fs_fail_(FS_EINVAL, "fs_csr_reduce_fill: ndof < 0", 0);
    if (!indptr || !indices || !free_rank || !red_indptr)
        return fs_fail_(FS_EINVAL, "fs_csr_reduce_fill: null argument", 0);
    if ((red_data || (f_red && u_fixed)) && !data) return fs_fail_(FS_EINVAL, "fs_csr_reduce_fill: data is null", 0);
    hipLaunchKernelGGL(reduce_fill_kernel, dim3(grid_of(ndof, WPB)), dim3(256), 0, (hipStream_t)stream, indptr,
                       indices, data, ndof, free_rank, red_indptr, red_indices, red_data, u_fixed, f, f_red);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? FS_OK : fs_fail_(FS_EHIP, "reduce_fill_kernel", (int)e);
}

int fs_expand_solution(const double *u_red, const int32_t *free_rank, const double *u_fixed, int64_t ndof,
                       double *u_full, void *stream) {
    if (ndof <= 0) return ndof == 0 ? FS_OK : fs_fail_(FS_EINVAL, "fs_expand_solution: ndof < 0", 0);
    if (!free_rank || !u_full) return fs_fail_(FS_EINVAL, "fs_expand_solution: null argument", 0);
    hipLaunchKernelGGL(expand_kernel, dim3(grid_of(ndof, 256)), dim3(256), 0, (hipStream_t)stream, u_red, free_rank,
                       u_fixed, ndof, u_full);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? FS_OK : fs_fail_(FS_EHIP, "expand_kernel", (int)e);
}

int fs_csr_row_sums(const int64_t *indptr, const double *data, int64_t nrows, double *out, void *stream) {
    if (nrows <= 0) return nrows == 0 ? FS_OK : fs_fail_(FS_EINVAL, "fs_csr_row_sums: nrows < 0", 0);
    if (!indptr || !data || !out) return fs_fail_(FS_EINVAL, "fs_csr_row_sums: null argument", 0);
    hipLaunchKernelGGL(row_sums_kernel, dim3(grid_of(nrows, WPB)), dim3(256), 0, (hipStream_t)stream, indptr, data,
                       nrows, out);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? FS_OK : fs_fail_(FS_EHIP, "row_sums_kernel", (int)e);
}

int fs_csr_blk_matvec(const int64_t *nbr_ptr, const int32_t *nbr, const int32_t *blk_row, int64_t n_nodes, int32_t d,
                      const double *data, const double *x, const double *y0, double *y, void *stream) {
    if (n_nodes <= 0) return n_nodes == 0 ? FS_OK : fs_fail_(FS_EINVAL, "fs_csr_blk_matvec: n_nodes < 0", 0);
    if (!nbr_ptr || !nbr || !blk_row || !data || !x || !y) return fs_fail_(FS_EINVAL, "fs_csr_blk_matvec: null argument", 0);
    if (d != 3 && d != 6) return fs_fail_(FS_EINVAL, "fs_csr_blk_matvec: d must be 3 or 6", 0);
    if (y == x) return fs_fail_(FS_EINVAL, "fs_csr_blk_matvec: y must not alias x", 0);
    const unsigned grid = grid_of(n_nodes, MV_NB);
    if (d == 6)
        hipLaunchKernelGGL(blk_matvec_kernel<6>, dim3(grid), dim3(256), 0, (hipStream_t)stream, nbr_ptr, nbr, blk_row, n_nodes,
                           data, x, y0, y);
    else
        hipLaunchKernelGGL(blk_matvec_kernel<3>, dim3(grid), dim3(256), 0, (hipStream_t)stream, nbr_ptr, nbr, blk_row, n_nodes,
                           data, x, y0, y);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? FS_OK : fs_fail_(FS_EHIP, "blk_matvec_kernel", (int)e);
}

int fs_inertial_forces(const double *coords, const double *vel, const double *mass, int64_t N, double omega,
                       double alpha, const double *axis_host, const double *center_host, uint32_t parts, double *F,
                       void *stream) {
    if (N <= 0) return N == 0 ? FS_OK : fs_fail_(FS_EINVAL, "fs_inertial_forces: N < 0", 0);
    if (!coords || !mass || !F || !axis_host) return fs_fail_(FS_EINVAL, "fs_inertial_forces: null argument", 0);
    if ((parts & 2u) && !vel) return fs_fail_(FS_EINVAL, "fs_inertial_forces: Coriolis part needs velocities", 0);
    const double c[3] = {center_host ? center_host[0] : 0.0, center_host ? center_host[1] : 0.0,
                         center_host ? center_host[2] : 0.0};
    hipLaunchKernelGGL(inertial_forces_kernel, dim3(grid_of(N, 256)), dim3(256), 0, (hipStream_t)stream, coords, vel,
                       mass, N, omega, alpha, axis_host[0], axis_host[1], axis_host[2], c[0], c[1], c[2], parts, F);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? FS_OK : fs_fail_(FS_EHIP, "inertial_forces_kernel", (int)e);
}

int fs_bench_fp64_fma(int32_t waves, int32_t iters, double *out, void *stream) {
    if (waves <= 0 || iters <= 0 || !out) return fs_fail_(FS_EINVAL, "fs_bench_fp64_fma: bad arguments", 0);
    hipLaunchKernelGGL(fp64_fma_kernel, dim3((unsigned)waves), dim3(64), 0, (hipStream_t)stream, out, (int)iters,
                       1.0000001, 1e-9);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? FS_OK : fs_fail_(FS_EHIP, "fp64_fma_kernel", (int)e);
}

}  // extern "C"
