// Probe of gfx950's v_permlane32_swap / v_permlane16_swap as a transposing reduction over the four 16-lane rows of a wave:
// lane = g * 16 + Q holds four values v0..v3; afterwards row r holds the sum over g of v_{pi(r)} of its column Q.
// build: hipcc --offload-arch=gfx950 -O3 tools/exp/permlane_probe.hip -o tools/exp/bin/permlane_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __forceinline__ void swap32(double &a, double &b) {
    unsigned alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
    auto r0 = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
    auto r1 = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
    a = __hiloint2double(r1[0], r0[0]); b = __hiloint2double(r1[1], r0[1]);
}
__device__ __forceinline__ void swap16(double &a, double &b) {
    unsigned alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
    auto r0 = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
    auto r1 = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
    a = __hiloint2double(r1[0], r0[0]); b = __hiloint2double(r1[1], r0[1]);
}
__global__ void k(double *out) {
    const int lane = threadIdx.x, g = lane >> 4, Q = lane & 15;
    // value k of lane (g, Q): 1000 * k + 100 * g + Q  -> sum over g of value k = 4000 k + 600 + 4 Q
    double v0 = 0 * 1000 + 100 * g + Q, v1 = 1000 + 100 * g + Q, v2 = 2000 + 100 * g + Q, v3 = 3000 + 100 * g + Q;
    swap32(v0, v1); swap32(v2, v3);
    double w01 = v0 + v1, w23 = v2 + v3;
    swap16(w01, w23);
    out[lane] = w01 + w23;
    // all-reduce of one value over the rows
    double a = 100 * g + Q, b = a;
    swap32(a, b); a = a + b; b = a; swap16(a, b);
    out[64 + lane] = a + b;
}
int main() {
    double *d, h[128];
    hipMalloc(&d, sizeof h);
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int g = 0; g < 4; g++) {
        int which = (int)((h[g * 16] - 600) / 4000 + 0.5), ok = 1;
        for (int Q = 0; Q < 16; Q++) ok &= h[g * 16 + Q] == 4000.0 * which + 600 + 4 * Q;
        printf("row %d holds the sum of value %d: %s\n", g, which, ok ? "ok" : "MISMATCH");
    }
    int ok = 1;
    for (int l = 0; l < 64; l++) ok &= h[64 + l] == 600 + 4 * (l & 15);
    printf("all-reduce: %s\n", ok ? "ok" : "MISMATCH");
    return 0;
}
