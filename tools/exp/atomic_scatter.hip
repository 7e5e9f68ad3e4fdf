// Microbenchmark for a fused evaluate-and-scatter path: the scatter half alone.  One wave per 16-element tile of a
// structured nx x ny quad mesh (row-major elements, as bench.py's strip) adds its 16 x 16 node-pair blocks (6 x 6 each)
// into the CSR value array with fp64 atomics, 10 block rows (48 contiguous bytes each) per wave instruction.
// Values are 1.0, so the result must equal the number of contributions of every entry.
//   hipcc --offload-arch=gfx950 -O3 tools/exp/atomic_scatter.hip -o /tmp/atomic_scatter && /tmp/atomic_scatter
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void block_pos(int nx, int ny, int I, int J, const int64_t *nbr_ptr, int64_t &dst, int &stride) {
    const int w = nx + 1;
    const int ix = I % w, iy = I / w, jx = J % w, jy = J / w;
    const int x0 = ix > 0 ? ix - 1 : 0, x1 = ix < nx ? ix + 1 : nx;          // neighbour columns of I
    const int y0 = iy > 0 ? iy - 1 : 0;
    const int per_row = x1 - x0 + 1;
    const int rank = (jy - y0) * per_row + (jx - x0);
    const int64_t r0 = nbr_ptr[I];
    const int m = (int)(nbr_ptr[I + 1] - r0);
    dst = 36 * r0 + 6 * (int64_t)rank;
    stride = 6 * m;
}

template <bool ATOMIC>
__global__ __launch_bounds__(64) void scatter_kernel(int nx, int ny, const int64_t *__restrict__ nbr_ptr,
                                                     double *__restrict__ data, int64_t E) {
    const int lane = threadIdx.x;
    const int r = lane / 6, l = lane - 6 * r;
    if (r >= 10) return;
    const int w = nx + 1;
    for (int it = 0; it < 154; it++) {                     // 1536 block rows of the tile, 10 per instruction
        const int R = it * 10 + r;
        if (R >= 1536) break;
        const int q = R / 96, rem = R - q * 96, blk = rem / 6, k = rem - blk * 6;
        const int64_t e = (int64_t)blockIdx.x * 16 + q;
        if (e >= E) continue;
        const int ex = (int)(e % nx), ey = (int)(e / nx);
        const int n[4] = {ey * w + ex, ey * w + ex + 1, (ey + 1) * w + ex + 1, (ey + 1) * w + ex};
        const int a = blk >> 2, b = blk & 3;
        int64_t dst; int stride;
        block_pos(nx, ny, n[a], n[b], nbr_ptr, dst, stride);
        double *p = data + dst + (int64_t)k * stride + l;
        if (ATOMIC) unsafeAtomicAdd(p, 1.0);
        else *p = 1.0;                                    // plain stores of the same shape (racy, timing only)
    }
}

int main(int argc, char **argv) {
    const int nx = argc > 1 ? atoi(argv[1]) : 400, ny = argc > 2 ? atoi(argv[2]) : 3125;
    const int64_t E = (int64_t)nx * ny, N = (int64_t)(nx + 1) * (ny + 1);
    std::vector<int64_t> ptr(N + 1);
    ptr[0] = 0;
    for (int64_t I = 0; I < N; I++) {
        const int ix = I % (nx + 1), iy = I / (nx + 1);
        const int cx = (ix > 0) + 1 + (ix < nx), cy = (iy > 0) + 1 + (iy < ny);
        ptr[I + 1] = ptr[I] + cx * cy;
    }
    const int64_t nnz = 36 * ptr[N];
    int64_t *d_ptr; double *d_data;
    CHK(hipMalloc(&d_ptr, (N + 1) * 8)); CHK(hipMalloc(&d_data, nnz * 8));
    CHK(hipMemcpy(d_ptr, ptr.data(), (N + 1) * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1, e2; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
    const unsigned grid = (unsigned)((E + 15) / 16);
    for (int mode = 0; mode < 2; mode++) {
        for (int rep = 0; rep < 4; rep++) {
            CHK(hipEventRecord(e0));
            CHK(hipMemsetAsync(d_data, 0, nnz * 8, 0));
            CHK(hipEventRecord(e1));
            if (mode == 0) hipLaunchKernelGGL(scatter_kernel<true>, dim3(grid), dim3(64), 0, 0, nx, ny, d_ptr, d_data, E);
            else hipLaunchKernelGGL(scatter_kernel<false>, dim3(grid), dim3(64), 0, 0, nx, ny, d_ptr, d_data, E);
            CHK(hipEventRecord(e2));
            CHK(hipEventSynchronize(e2));
            float ms0, ms1; hipEventElapsedTime(&ms0, e0, e1); hipEventElapsedTime(&ms1, e1, e2);
            printf("%s: memset %.3f ms  scatter %.3f ms  (%.2f G adds/s, nnz %.3g)\n", mode == 0 ? "atomic" : "store ", ms0, ms1,
                   E * 576.0 / ms1 / 1e6, (double)nnz);
        }
        if (mode == 0) {                                   // check: interior diagonal entry has 4 contributions
            std::vector<double> h(36 * 9);
            const int64_t I = (int64_t)(ny / 2) * (nx + 1) + nx / 2;
            CHK(hipMemcpy(h.data(), d_data + 36 * ptr[I], 36 * 9 * 8, hipMemcpyDeviceToHost));
            printf("row of an interior node: ");
            for (int c = 0; c < 9; c++) printf("%g ", h[6 * c]);
            printf("(expect 1 2 1 2 4 2 1 2 1)\n");
        }
    }
    return 0;
}
