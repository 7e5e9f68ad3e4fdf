// Microbenchmark: fp64 FMA throughput of ONE wave per SIMD as a function of independent chains,
// and of two waves per SIMD.  hipcc --offload-arch=gfx950 -O3 tools/exp/fma_ilp.hip -o /tmp/fma_ilp
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int NCH>
__global__ __launch_bounds__(64) void k(double *out, int iters, double a, double b) {
    double acc[NCH];
    for (int c = 0; c < NCH; c++) acc[c] = threadIdx.x + c;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int c = 0; c < NCH; c++) acc[c] = __builtin_fma(acc[c], a, b);
    }
    double s = 0;
    for (int c = 0; c < NCH; c++) s += acc[c];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int NCH>
void run(int waves, double *out) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NCH>, dim3(waves), dim3(64), 0, 0, out, 10, 1.0000001, 1e-9);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NCH>, dim3(waves), dim3(64), 0, 0, out, iters, 1.0000001, 1e-9);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = 2.0 * 64 * waves * (double)iters * 8 * NCH;
    printf("waves %5d chains %d : %.3f ms  %.2f TFLOP/s\n", waves, NCH, ms, flops / ms / 1e9);
}
int main() {
    double *out; hipMalloc(&out, 8 * 64 * 8192);
    for (int waves : {1024, 2048, 4096}) {
        run<1>(waves, out); run<2>(waves, out); run<4>(waves, out); run<8>(waves, out);
    }
    return 0;
}
