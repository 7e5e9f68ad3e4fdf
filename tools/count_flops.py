#!/usr/bin/env python3
"""Count the fp64 operations of the MITC4+ stiffness path from the gfx950 ISA of the SHIPPED device functions
(fem_shell_amd/csrc/femshell_hip.hip: quad_geometry, quad_stage1, quad_pair_block) -- SURVEY.md 8(d) asks that the
FLOP figures be re-derived by instrumenting the implementation instead of being estimated.  Every function is fully
unrolled straight-line code, so the static instruction count of a probe kernel that calls it once IS the dynamic count
per call.  FMA = 2 flop, add / mul / rcp / sqrt = 1; moves, selects, compares, DPP exchanges and integer work are not
flops.  Writes bench/FLOPS.md.   usage: python tools/count_flops.py"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "fem_shell_amd", "csrc")
PROBE = r'''
#include "femshell_hip.hip"
namespace {
__global__ void probe_geom(const double *xyz, double *out) {
    double x[4][3];
    for (int t = 0; t < 12; t++) (&x[0][0])[t] = xyz[threadIdx.x * 12 + t];
    QuadGeom g; quad_geometry(x, g);
    double *o = out + threadIdx.x * 32;
    for (int k = 0; k < 3; k++) { o[k] = g.e1[k]; o[3 + k] = g.e2[k]; o[6 + k] = g.e3[k]; }
    for (int t = 0; t < 8; t++) o[9 + t] = (&g.xl[0][0])[t];
    for (int t = 0; t < 5; t++) o[17 + t] = g.a[t];
}
template <bool C>
__global__ void probe_stage1(const double *xyz, const fs_shell_prop *props, double *out) {
    __shared__ double lds[LDS_WAVE];
    double x[4][3];
    const int Q = threadIdx.x >> 2, g = threadIdx.x & 3;
    for (int t = 0; t < 12; t++) (&x[0][0])[t] = xyz[Q * 12 + t];
    QuadGeom geo; QuadElem el;
    quad_stage1(x, props, C, C, g, Q, lds, geo, el);
    __syncthreads();
    double *o = out + threadIdx.x * 64;
    for (int t = 0; t < 15; t++) o[t] = el.S[t];
    for (int t = 0; t < 18; t++) o[15 + t] = (&el.gt[0][0])[t];
    o[33] = el.i00; o[34] = el.i01; o[35] = el.i11; o[36] = lds[threadIdx.x * 37];
}
template <bool C, bool DIAG>
__global__ void probe_pair(const double *in, double *out, int ti, int tj) {
    __shared__ double lds[LDS_WAVE];
    for (int t = threadIdx.x; t < LDS_WAVE; t += 64) lds[t] = in[t];
    __syncthreads();
    const double *p = in + LDS_WAVE + threadIdx.x * 128;
    double Sf[5][5], gt[6][3], R[3][3], Cb[3][3], Cs[2][2];
    for (int t = 0; t < 25; t++) (&Sf[0][0])[t] = p[t];
    for (int t = 0; t < 18; t++) (&gt[0][0])[t] = p[25 + t];
    for (int t = 0; t < 9; t++) { (&R[0][0])[t] = p[43 + t]; (&Cb[0][0])[t] = p[52 + t]; }
    for (int t = 0; t < 4; t++) (&Cs[0][0])[t] = p[61 + t];
    double G[6][6];
    quad_pair_block(lds, threadIdx.x & 15, DIAG ? ti : ti, DIAG ? ti : tj, DIAG, Sf, gt, R, p[65], p[66], p[67], Cb, Cs, p[68], C, G);
    for (int t = 0; t < 36; t++) out[threadIdx.x * 36 + t] = (&G[0][0])[t];
}
template __global__ void probe_stage1<false>(const double *, const fs_shell_prop *, double *);
template __global__ void probe_stage1<true>(const double *, const fs_shell_prop *, double *);
template __global__ void probe_pair<false, false>(const double *, double *, int, int);
template __global__ void probe_pair<false, true>(const double *, double *, int, int);
template __global__ void probe_pair<true, false>(const double *, double *, int, int);
}
'''
FLOP = {"v_fma_f64": 2, "v_mul_f64": 1, "v_add_f64": 1, "v_rcp_f64": 1, "v_sqrt_f64": 1, "v_rsq_f64": 1,
        "v_fmac_f64": 2}


def count(asm, key):
    lines = asm.split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    for k, i in enumerate(starts):
        if key in lines[i]:
            j = starts[k + 1] if k + 1 < len(starts) else len(lines)
            c = collections.Counter()
            for l in lines[i:j]:
                t = l.strip().split()
                if not t:
                    continue
                op = re.sub(r"_e(32|64)$|_dpp$|_sdwa$", "", t[0])
                if op == "s_endpgm":
                    break
                if op in FLOP:
                    c[op] += 1
                if t[0].startswith("v_") and "f64" in t[0]:
                    c["_all_f64"] += 1
            return c
    raise KeyError(key)


def flops(c):
    return sum(FLOP[k] * v for k, v in c.items() if k in FLOP)


def main():
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "probe.hip")
        open(src, "w").write(PROBE)
        out = os.path.join(td, "probe.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17",
                               "-I" + SRC, "-I" + os.path.join(ROOT, "include"), "-Wno-unused-function", "-S",
                               "--cuda-device-only", src, "-o", out])
        asm = open(out).read()
    geom = count(asm, "probe_geom")
    s1 = count(asm, "probe_stage1ILb0")
    s1c = count(asm, "probe_stage1ILb1")
    pr = count(asm, "probe_pairILb0ELb0")
    prd = count(asm, "probe_pairILb0ELb1")
    prc = count(asm, "probe_pairILb1ELb0")
    fg, f1, f1c, fp, fpd, fpc = map(flops, (geom, s1, s1c, pr, prd, prc))
    gp = f1 - fg                      # per-Gauss-point part of stage 1 (4 per element)
    gpc = f1c - fg
    alg = fg + 4 * gp + 6 * fp + 4 * fpd
    alg_c = fg + 4 * gpc + 6 * fpc + 4 * (fpd + (fpc - fp))
    exe = 4 * f1 + 6 * fp + 4 * fpd
    exe_c = 4 * f1c + 6 * fpc + 4 * (fpd + (fpc - fp))
    md = f"""# FLOP count of the MITC4+ stiffness path (SURVEY.md 8d), re-derived from the shipped code

Method (`tools/count_flops.py`): the device functions the kernels are built from are straight-line code, so the static
fp64 instruction count of a probe kernel calling one of them once is the dynamic count per call (gfx950 ISA of this
repository's `fem_shell_amd/csrc/femshell_hip.hip`, hipcc -O3).  FMA = 2 flop; add, mul, rcp, sqrt = 1; moves, selects,
compares, DPP lane exchanges and integer work are not counted.

| piece | calls per element | fp64 arithmetic instructions | flop |
|---|---|---|---|
| `quad_geometry` (frame, local coordinates, characteristic vectors, a_A..a_E) | 1 (algorithmic) / 4 (executed: every Gauss-point lane repeats it) | {geom['_all_f64']} | {fg} |
| rest of `quad_stage1` (one Gauss point: J, tying tangents, membrane core, shear scale factors, bubble rows, quad reductions) | 4 | {s1['_all_f64'] - geom['_all_f64']} | {gp} |
| same with membrane-bending coupling (asymmetric laminates) | 4 | {s1c['_all_f64'] - geom['_all_f64']} | {gpc} |
| `quad_pair_block`, off-diagonal node pair | 6 | {pr['_all_f64']} | {fp} |
| `quad_pair_block`, diagonal node pair (+ symmetrisation) | 4 | {prd['_all_f64']} | {fpd} |
| `quad_pair_block` with coupling, off-diagonal | 6 | {prc['_all_f64']} | {fpc} |

| per element | flop |
|---|---|
| **algorithmic, isotropic / symmetric laminate** (geometry once) | **{alg}** |
| **algorithmic, with coupling** | **{alg_c}** |
| executed by `mitc4_k_kernel` (geometry on all 4 Gauss-point lanes) | {exe} ({exe_c} with coupling) |
| executed by the fused kernel (stage 1 also for the halo elements of a patch: x 1.31 on the C5 plate) | {int(1.31 * 4 * f1 + 6 * fp + 4 * fpd)} ({int(1.31 * 4 * f1c + 6 * fpc + 4 * (fpd + fpc - fp))} with coupling) |

SURVEY.md 8(d) estimated 27 000 (+4 000 with coupling).  `bench.py` prices `roofline.achieved` with the ALGORITHMIC
figures of this table ({alg} / {alg_c}), weighted by the share of coupled elements of the workload; the executed counts are
given for orientation only (redundant work is not credited).
"""
    os.makedirs(os.path.join(ROOT, "bench"), exist_ok=True)
    open(os.path.join(ROOT, "bench", "FLOPS.md"), "w").write(md)
    import json
    json.dump({"mitc4_k": alg, "mitc4_k_coupled": alg_c, "executed_k_kernel": exe, "executed_k_kernel_coupled": exe_c},
              open(os.path.join(ROOT, "bench", "flops.json"), "w"), indent=1)
    print(md)


if __name__ == "__main__":
    main()
