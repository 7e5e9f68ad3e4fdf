// Microbenchmark: device copy bandwidth (read + write) for a few kernel shapes, 4 GiB per direction.
//   hipcc --offload-arch=gfx950 -O3 tools/exp/copy_bw.hip -o /tmp/copy_bw
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef double double2_ __attribute__((ext_vector_type(2)));

template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void copy_kernel(const double2_ *__restrict__ src, double2_ *__restrict__ dst, int64_t n) {
    // each workgroup copies contiguous chunks of 256 * UNROLL 16-byte pieces; grid-stride over chunks
    const int64_t chunk = 256 * UNROLL;
    for (int64_t base = (int64_t)blockIdx.x * chunk; base < n; base += (int64_t)gridDim.x * chunk) {
        double2_ v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const int64_t i = base + u * 256 + threadIdx.x;
            if (i < n) v[u] = NT ? __builtin_nontemporal_load(src + i) : src[i];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const int64_t i = base + u * 256 + threadIdx.x;
            if (i < n) { if (NT) __builtin_nontemporal_store(v[u], dst + i); else dst[i] = v[u]; }
        }
    }
}

template <int UNROLL, bool NT>
void run(const char *name, const double2_ *src, double2_ *dst, int64_t n, unsigned grid) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL((copy_kernel<UNROLL, NT>), dim3(grid), dim3(256), 0, 0, src, dst, n);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL((copy_kernel<UNROLL, NT>), dim3(grid), dim3(256), 0, 0, src, dst, n);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-28s grid %8u : %.3f ms  %.2f TB/s (read + write)\n", name, grid, ms / 5, 2.0 * n * 16 / (ms / 5) / 1e9);
}

int main() {
    const int64_t n = (int64_t)1 << 28;      // 16-byte pieces: 4 GiB
    double2_ *a, *b;
    CHK(hipMalloc(&a, n * 16)); CHK(hipMalloc(&b, n * 16));
    CHK(hipMemset(a, 1, n * 16));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    CHK(hipMemcpy(b, a, n * 16, hipMemcpyDeviceToDevice));
    CHK(hipEventRecord(e0));
    for (int r = 0; r < 5; r++) CHK(hipMemcpyAsync(b, a, n * 16, hipMemcpyDeviceToDevice, 0));
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-28s               : %.3f ms  %.2f TB/s\n", "hipMemcpyDtoD", ms / 5, 2.0 * n * 16 / (ms / 5) / 1e9);
    for (unsigned grid : {2048u, 8192u, 65536u, 1048576u}) {
        run<1, false>("1 piece/lane", a, b, n, grid);
        run<4, false>("4 pieces/lane", a, b, n, grid);
        run<8, false>("8 pieces/lane", a, b, n, grid);
        run<4, true>("4 pieces/lane non-temporal", a, b, n, grid);
    }
    return 0;
}
