// Does a wave's store stream overlap its own arithmetic at one wave per SIMD?  Each workgroup (256 lanes, the whole LDS, so one
// per CU) repeats: NF dependent-free fp64 FMAs per lane, NS 1 KB store instructions (global_store_dwordx4) to fresh addresses.
// Modes: 0 = FMAs only, 1 = stores only, 2 = FMAs then the NS stores in one burst, 3 = one store after every NF / NS FMAs.
// build: hipcc --offload-arch=gfx950 -O3 tools/exp/store_overlap.hip -o tools/exp/bin/store_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
constexpr int NF = 4608, NS = 36, CH = NF / NS;
template <int MODE>
__global__ __launch_bounds__(256) void k(double2 *out, int iters, double seed) {
    __shared__ double big[19000];          // 152 KB: one workgroup per CU
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    big[threadIdx.x] = seed;
    double a[8];
    for (int i = 0; i < 8; i++) a[i] = seed + i + lane;
    const double m = 1.0000001, c = 1e-9;
    double ld[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const double2 sd = make_double2(seed + lane, seed - lane);      // store data: registers the FMAs never write
    // per workgroup and iteration: 4 waves x NS KB, fresh addresses
    double2 *base = out + ((size_t)blockIdx.x * iters * 4 + wave) * NS * 64 + lane;
    for (int it = 0; it < iters; it++) {
        double2 *o = base + (size_t)it * 4 * NS * 64;
        if (MODE == 0 || MODE == 2) {
#pragma unroll 64
            for (int f = 0; f < NF / 8; f++)
                for (int i = 0; i < 8; i++) a[i] = __builtin_fma(a[i], m, c);
        }
        if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int s = 0; s < NS; s++) o[s * 64] = sd;
        }
        if (MODE == 3) {
#pragma unroll
            for (int s = 0; s < NS; s++) {
#pragma unroll
                for (int f = 0; f < CH / 8; f++)
                    for (int i = 0; i < 8; i++) a[i] = __builtin_fma(a[i], m, c);
                o[s * 64] = sd;
            }
        }
        if (MODE == 4) {          // one 1 KB LOAD after every NF / NS FMAs; the values are used after the loop only
#pragma unroll
            for (int s = 0; s < NS; s++) {
#pragma unroll
                for (int f = 0; f < CH / 8; f++)
                    for (int i = 0; i < 8; i++) a[i] = __builtin_fma(a[i], m, c);
                const double2 v = o[s * 64];
                ld[s & 7] = v.x + v.y;
            }
        }
        if (MODE == 5) {          // the NS loads in one burst behind the FMAs
#pragma unroll 64
            for (int f = 0; f < NF / 8; f++)
                for (int i = 0; i < 8; i++) a[i] = __builtin_fma(a[i], m, c);
#pragma unroll
            for (int s = 0; s < NS; s++) { const double2 v = o[s * 64]; ld[s & 7] = v.x + v.y; }
        }
    }
    for (int i = 0; i < 8; i++) a[i] += ld[i];
    if (a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7] == 12345.678) out[0] = make_double2(big[lane], 0);
}
template <int MODE>
float run(double2 *d, int grid, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<grid, 256>>>(d, iters, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<grid, 256>>>(d, iters, 1.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main(int argc, char **argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 256, iters = 100;
    const size_t bytes = (size_t)grid * iters * 4 * NS * 1024;
    double2 *d;
    if (hipMalloc(&d, bytes) != hipSuccess) return 1;
    hipMemset(d, 0, bytes);
    const float t0 = run<0>(d, grid, iters), t1 = run<1>(d, grid, iters), t2 = run<2>(d, grid, iters), t3 = run<3>(d, grid, iters);
    printf("per iteration (4608 FMAs per lane, 36 KB per wave = 144 KB per CU; %d CUs x %d iterations, %.2f GB):\n", grid, iters, bytes / 1e9);
    printf("  FMAs only           %7.3f us\n  stores only         %7.3f us  (%.2f TB/s, %.1f B/clk/CU at 2.4 GHz)\n", 1e3 * t0 / iters, 1e3 * t1 / iters,
           bytes / (t1 * 1e-3) / 1e12, bytes / (double)grid / (t1 * 1e-3 * 2.4e9));
    printf("  FMAs, then burst    %7.3f us\n  interleaved         %7.3f us\n", 1e3 * t2 / iters, 1e3 * t3 / iters);
    const float t4 = run<4>(d, grid, iters), t5 = run<5>(d, grid, iters);
    printf("  LOADS interleaved   %7.3f us\n  FMAs, then 36 loads %7.3f us\n", 1e3 * t4 / iters, 1e3 * t5 / iters);
    return 0;
}
