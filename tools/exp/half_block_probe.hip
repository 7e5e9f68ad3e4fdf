// Register need of quad_pair_block cut into part blocks (round 3): the whole 6 x 6 node-pair block, the half block
// (6 x 3) and the quarter (3 x 3); the columns / rows that are not stored are dead code for the compiler.
//   cd fem_shell_amd/csrc && hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -I../../include -I. \
//       -Rpass-analysis=kernel-resource-usage --cuda-device-only -c ../../tools/exp/half_block_probe.hip -o /dev/null
// and the same with  -DPROBE_WAVES=4  (amdgpu_waves_per_eu(4, 4): at most 128 registers).  Results: profiles/r03_half_block_probe.txt
#include "femshell_hip.hip"
namespace {
// NC = number of block columns kept (6 = whole 6 x 6 block, 3 = half block, the rest is dead code for the compiler)
#ifdef PROBE_WAVES
#define PROBE_ATTR __attribute__((amdgpu_waves_per_eu(PROBE_WAVES, PROBE_WAVES)))
#else
#define PROBE_ATTR
#endif
template <bool C, int NR, int NC>
__global__ __launch_bounds__(256) PROBE_ATTR void probe_part(const double *in, double *out, int ti, int tj) {
    __shared__ double lds[LDS_WAVE];
    for (int t = threadIdx.x; t < LDS_WAVE; t += 256) lds[t] = in[t];
    __syncthreads();
    const double *p = in + LDS_WAVE + threadIdx.x * 128;
    double R[3][3], Cb[3][3];      // (membrane core, tying tangents and the shear core are read from the image since round 3)
    for (int t = 0; t < 9; t++) { (&R[0][0])[t] = p[43 + t]; (&Cb[0][0])[t] = p[52 + t]; }
    double G[6][6];
    quad_pair_block(lds, threadIdx.x & 15, ti, tj, false, R, p[65], p[66], p[67], Cb, p[68], C, G);
    for (int a = 0; a < NR; a++)
        for (int b = 0; b < NC; b++) out[threadIdx.x * 36 + a * 6 + b] = G[a][b];
}
template __global__ void probe_part<false, 6, 6>(const double *, double *, int, int);
template __global__ void probe_part<false, 6, 3>(const double *, double *, int, int);
template __global__ void probe_part<false, 3, 3>(const double *, double *, int, int);
template __global__ void probe_part<true, 6, 6>(const double *, double *, int, int);
template __global__ void probe_part<true, 6, 3>(const double *, double *, int, int);
}
