// mfma_pair_proto.hip -- MEASUREMENT prototype (not product code): the Gauss-point contraction of the MITC4+ stiffness,
//   K_local (24 x 24) = sum over k of Bt[k][:]^T  CB[k][:],   k = 40 strain rows per element
//   (membrane 3 x 4 Gauss points, bending 3 x 4, shear 2 x 4, drilling 1 x 4, bubble condensation 2, padded 38 -> 40;
//   elements/MITC4.py:934-1016 written as one product),
// on the fp64 matrix pipe (v_mfma_f64_16x16x4_f64) instead of the structured VALU form of quad_pair_block.  north_star asks
// for this measurement ("fp64 MFMA only if rocprof shows the 24x24 contractions are matrix-bound").
//
// The MFMA route gets every advantage: its operands (Bt and CB, 40 x 32 doubles each, dofs padded 24 -> 32) already sit
// in LDS in exactly the order the instruction wants -- in a real kernel stage 1 would have to BUILD them from the
// compact image (rotation to global axes and the C products included), and the accumulators would still have to be
// condensed, summed over elements in element order and stored.  Only the 3 upper 16 x 16 tiles are formed (30 MFMAs
// per element).  Variants: V extra independent fp64 FMAs per MFMA on the VALU (do the two pipes overlap?), and 1 / 2
// waves per SIMD.
//
//   hipcc --offload-arch=gfx950 -O3 tools/exp/mfma_pair_proto.hip -o gpurun_out/mfma_proto && gpurun_out/mfma_proto
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int KR = 40, ND = 32;
constexpr int OPER = KR * ND;     // doubles per operand matrix of one element

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

// One wave per element slot; the wave's operands are staged in LDS once and the contraction is repeated `reps` times
// (different accumulators each time would be the honest form for distinct elements; the instruction stream is the same).
template <int V>
__global__ __launch_bounds__(256) void mfma_elem_kernel(const double *__restrict__ Bt, const double *__restrict__ CB,
                                                       double *__restrict__ C, int reps, double *__restrict__ sink) {
    __shared__ double lds[4 * 2 * OPER];          // 4 waves x (Bt, CB) x 40 x 32 doubles = 81 920 B
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t w = (int64_t)blockIdx.x * 4 + wave;
    double *a = lds + wave * 2 * OPER, *b = a + OPER;
    for (int t = lane; t < OPER; t += 64) { a[t] = Bt[w * OPER + t]; b[t] = CB[w * OPER + t]; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int i = lane & 15, kq = lane >> 4;
    d4 tot00 = {0, 0, 0, 0}, tot01 = tot00, tot11 = tot00;
    double v[8] = {1.0 + lane, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0, 8.0};
    const double vm = 1.0000001, va = 1e-9;
    for (int r = 0; r < reps; r++) {
        d4 c00 = {0, 0, 0, 0}, c01 = c00, c11 = c00;
#pragma unroll
        for (int kk = 0; kk < KR / 4; kk++) {
            const int k = 4 * kk + kq;
            // A operand: lane l holds A[i = l & 15][k = l >> 4] = Bt[k][i]; B operand: B[k = l >> 4][j = l & 15] = CB[k][j]
            const double a0 = a[k * ND + i], a1 = a[k * ND + 16 + i];
            const double b0 = b[k * ND + i], b1 = b[k * ND + 16 + i];
            c00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, c00, 0, 0, 0);
            if (V > 0) { _Pragma("unroll") for (int q = 0; q < V; q++) v[q & 7] = __builtin_fma(v[q & 7], vm, va); }
            c01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, c01, 0, 0, 0);
            if (V > 0) { _Pragma("unroll") for (int q = 0; q < V; q++) v[q & 7] = __builtin_fma(v[q & 7], vm, va); }
            c11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, c11, 0, 0, 0);
            if (V > 0) { _Pragma("unroll") for (int q = 0; q < V; q++) v[q & 7] = __builtin_fma(v[q & 7], vm, va); }
        }
        tot00 += c00; tot01 += c01; tot11 += c11;
        asm volatile("" ::: "memory");
    }
    // C/D layout of the f64 MFMA: col = lane & 15, row = (lane >> 4) + 4 * reg
    double *c = C + w * ND * ND;
    for (int q = 0; q < 4; q++) {
        const int row = kq + 4 * q;
        c[row * ND + i] = tot00[q];
        c[row * ND + 16 + i] = tot01[q];
        c[(16 + row) * ND + 16 + i] = tot11[q];
    }
    double s = 0;
    for (int q = 0; q < 8; q++) s += v[q];
    if (s == 12345.678) sink[0] = s;
}

// VALU reference of the same DENSE product (no structure exploited), for the numbers only: lane = output entry
__global__ __launch_bounds__(256) void valu_dense_kernel(const double *__restrict__ Bt, const double *__restrict__ CB,
                                                        double *__restrict__ C, int reps) {
    __shared__ double lds[4 * 2 * OPER];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t w = (int64_t)blockIdx.x * 4 + wave;
    double *a = lds + wave * 2 * OPER, *b = a + OPER;
    for (int t = lane; t < OPER; t += 64) { a[t] = Bt[w * OPER + t]; b[t] = CB[w * OPER + t]; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // 24 x 24 upper triangle incl. diagonal = 300 entries -> 5 per lane (64 lanes, 320 slots)
    double acc[5] = {0, 0, 0, 0, 0};
    int ri[5], ci[5];
    for (int q = 0; q < 5; q++) {
        int t = lane * 5 + q; if (t >= 300) t = 299;
        int r = 0; while (t >= 24 - r) { t -= 24 - r; r++; }
        ri[q] = r; ci[q] = r + t;
    }
    for (int rep = 0; rep < reps; rep++) {
        for (int k = 0; k < KR; k++)
            for (int q = 0; q < 5; q++) acc[q] = __builtin_fma(a[k * ND + ri[q]], b[k * ND + ci[q]], acc[q]);
        asm volatile("" ::: "memory");
    }
    for (int q = 0; q < 5; q++) C[w * ND * ND + ri[q] * ND + ci[q]] = acc[q];
}

template <int V>
static double run(int nwg, int reps, const double *dA, const double *dB, double *dC, double *sink) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(mfma_elem_kernel<V>, dim3(nwg), dim3(256), 0, 0, dA, dB, dC, 8, sink);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(mfma_elem_kernel<V>, dim3(nwg), dim3(256), 0, 0, dA, dB, dC, reps, sink);
    CHK(hipEventRecord(e1));
    CHK(hipDeviceSynchronize());
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

int main() {
    const int max_wg = 512, nw = max_wg * 4;
    std::vector<double> hA((size_t)nw * OPER), hB((size_t)nw * OPER), hC((size_t)nw * ND * ND);
    srand(1);
    for (auto &x : hA) x = (rand() % 2001 - 1000) / 1000.0;
    for (auto &x : hB) x = (rand() % 2001 - 1000) / 1000.0;
    double *dA, *dB, *dC, *sink;
    CHK(hipMalloc(&dA, hA.size() * 8)); CHK(hipMalloc(&dB, hB.size() * 8)); CHK(hipMalloc(&dC, hC.size() * 8)); CHK(hipMalloc(&sink, 8));
    CHK(hipMemcpy(dA, hA.data(), hA.size() * 8, hipMemcpyHostToDevice));
    CHK(hipMemcpy(dB, hB.data(), hB.size() * 8, hipMemcpyHostToDevice));
    // ---- layout check: reps = 1, every wave's three tiles against a host triple loop (asymmetric operands)
    CHK(hipMemset(dC, 0, hC.size() * 8));
    hipLaunchKernelGGL(mfma_elem_kernel<0>, dim3(4), dim3(256), 0, 0, dA, dB, dC, 1, sink);
    CHK(hipDeviceSynchronize());
    CHK(hipMemcpy(hC.data(), dC, 16 * ND * ND * 8, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int w = 0; w < 16; w++)
        for (int r = 0; r < ND; r++)
            for (int c = 0; c < ND; c++) {
                if (r >= 16 && c < 16) continue;
                double ref = 0;
                for (int k = 0; k < KR; k++) ref = fma(hA[(size_t)w * OPER + k * ND + r], hB[(size_t)w * OPER + k * ND + c], ref);
                worst = fmax(worst, fabs(ref - hC[(size_t)w * ND * ND + r * ND + c]));
            }
    printf("layout check (3 tiles x 16 waves vs host fma chain): max |diff| = %.3g %s\n", worst, worst < 1e-12 ? "OK" : "WRONG");
    // ---- timing
    const int reps = 2000;
    const double flop_tile = 2.0 * 16 * 16 * 4, mf_per_el = 3.0 * KR / 4;
    printf("%-34s %10s %14s %16s %18s\n", "variant", "ms", "cycles/elem", "dense TFLOP/s", "ms per 1.25 M el");
    for (int wps = 1; wps <= 2; wps++) {
        const int nwg = 256 * wps;                 // 256 CUs x wps workgroups of 4 waves = wps waves per SIMD
        const double elems = (double)nwg * 4 * reps;
        struct { const char *n; double ms; int v; } rows[4] = {
            {"MFMA only", run<0>(nwg, reps, dA, dB, dC, sink), 0},
            {"MFMA + 4 VALU FMA per MFMA", run<4>(nwg, reps, dA, dB, dC, sink), 4},
            {"MFMA + 8 VALU FMA per MFMA", run<8>(nwg, reps, dA, dB, dC, sink), 8},
            {"MFMA + 16 VALU FMA per MFMA", run<16>(nwg, reps, dA, dB, dC, sink), 16}};
        for (auto &r : rows) {
            const double s_per_el_simd = r.ms * 1e-3 / reps / wps;        // one SIMD finishes `wps` elements per rep
            printf("%d wave/SIMD, %-20s %10.3f %14.0f %16.2f %18.3f\n", wps, r.n, r.ms, s_per_el_simd * 2.4e9,
                   elems * mf_per_el * flop_tile / (r.ms * 1e-3) / 1e12, s_per_el_simd * 1.25e6 / 1024 * 1e3);
        }
    }
    {   // dense VALU form of the same product (upper triangle, 300 entries x 40 k), for scale
        hipEvent_t e0, e1;
        CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        hipLaunchKernelGGL(valu_dense_kernel, dim3(256), dim3(256), 0, 0, dA, dB, dC, 8);
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(valu_dense_kernel, dim3(256), dim3(256), 0, 0, dA, dB, dC, reps);
        CHK(hipEventRecord(e1));
        CHK(hipDeviceSynchronize());
        float ms = 0;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        const double s_per_el = ms * 1e-3 / reps;
        printf("1 wave/SIMD, dense VALU upper triangle (LDS operands) %8.3f ms %10.0f cycles/elem %12.3f ms per 1.25 M el\n", ms,
               s_per_el * 2.4e9, s_per_el * 1.25e6 / 1024 * 1e3);
    }
    return worst < 1e-12 ? 0 : 1;
}
