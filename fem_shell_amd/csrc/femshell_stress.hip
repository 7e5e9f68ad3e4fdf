// femshell_stress.hip -- batched stress / strain recovery on the device (MI355X / gfx950), the consumer
// right after the solve.  Restates postprocess/stress_recovery.py of the reference:
//
//   fs_shell_stress_batch   StressRecovery._compute_shell_stress / _compute_shell_strain (:529-650) at a
//                           list of parametric points of every MITC4 / MITC3 element, + the element
//                           area of the area-weighted smoothing (:1372-1388)
//   fs_solid_stress_batch   _compute_solid_gauss_stresses (:655-705) at every Gauss point, their mean
//                           (compute_element_stresses, :761-764) or the Gauss -> node extrapolation
//                           E @ sigma_gp (:850-855), + the element volume (:1391-1402)
//   fs_nodal_average        the weighted node-by-node accumulation and division (:844-864)
//   fs_stress_invariants    _build_stress_result / _build_strain_result (:1087-1341): von Mises, Mohr
//                           circle, Cardano principal stresses
// Citations: file:line relative to /root/reference/src/fem_shell/.  Gather / B u / C eps / scatter-average
// is byte work bounded by HBM (a few hundred bytes per element); one thread per (element, point).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "femshell_hip.h"
#include "femshell_internal.h"
#include "femshell_geom.h"

namespace {

#define ST_LAUNCH_CHECK(name)                                                     \
    do {                                                                          \
        hipError_t e__ = hipGetLastError();                                       \
        if (e__ != hipSuccess) return fs_fail_(FS_EHIP, name, (int)e__);          \
    } while (0)

inline unsigned grid_of(int64_t n, int block) { return (unsigned)((n + block - 1) / block); }

struct StressPts { double r[4], s[4]; int n; };

// ------------------------------------------------------------------------------------------
// Shells.  eps_m = B_m u, kappa = B_kappa u with the DISPLACEMENT-based B_m / B_kappa of the
// element classes (MITC4.py:756-790, MITC3.py:670-739) applied to the element displacement vector
// exactly as the reference does: u_e is the slice of the global vector (global axes), no
// T-transformation (stress_recovery.py:497-524,588-595 -- kept as is).
//   stress = (Cm / h) eps_m  [+]  (Cm / h) (z kappa),   strain = eps_m + z kappa
// ------------------------------------------------------------------------------------------
template <int NN>
__global__ __launch_bounds__(256) void shell_stress_kernel(const double *__restrict__ xyz,
                                                           const int32_t *__restrict__ prop_id,
                                                           const fs_shell_prop *__restrict__ props, int32_t n_props,
                                                           const double *__restrict__ u_e, int64_t E,
                                                           StressPts pts, int location, uint32_t parts,
                                                           int strain, double *__restrict__ out,
                                                           double *__restrict__ weight) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= E * pts.n) return;
    const int64_t e = t / pts.n;
    const int p = (int)(t - e * pts.n);
    double r = 0.0, s = 0.0;
    for (int k = 0; k < 4; k++) { r = (k == p) ? pts.r[k] : r; s = (k == p) ? pts.s[k] : s; }
    const fs_shell_prop *pr = props + safe_pid(prop_id[e], n_props);
    double dNx[NN], dNy[NN], area = 0.0;
    if constexpr (NN == 4) {
        double x[4][3];
        for (int c = 0; c < 12; c++) (&x[0][0])[c] = xyz[e * 12 + c];
        QuadGeom geo;
        quad_geometry(x, geo);
        double dxi[4], deta[4], ia, ib, ic, id;
        dshape(r, s, dxi, deta);
        jacobian(geo, dxi, deta, ia, ib, ic, id);
        for (int i = 0; i < 4; i++) {   // MITC4.py:652-653: two rounded products, one sum
            dNx[i] = ia * dxi[i] + ib * deta[i];
            dNy[i] = ic * dxi[i] + id * deta[i];
        }
        if (weight) {                   // stress_recovery.py:1384-1388 over the 2x2 rule, weights 1
            for (int g = 0; g < 4; g++) {
                const double xg = (g == 1 || g == 2) ? kGP : -kGP, yg = (g >= 2) ? kGP : -kGP;
                double a, b, c, d;
                dshape(xg, yg, dxi, deta);
                area += jacobian(geo, dxi, deta, a, b, c, d) * 1.0;
            }
        }
    } else {
        double x[3][3];
        for (int c = 0; c < 9; c++) (&x[0][0])[c] = xyz[e * 9 + c];
        TriGeom geo;
        tri_geometry(x, geo);
        const double dr[3] = {-1.0, 1.0, 0.0}, ds[3] = {-1.0, 0.0, 1.0};
        for (int i = 0; i < 3; i++) {   // MITC3.py:553-554 with J^-1[0][1] == 0
            dNx[i] = geo.J00 * dr[i] + 0.0 * ds[i];
            dNy[i] = geo.J10 * dr[i] + geo.J11 * ds[i];
        }
        if (weight) {                   // Hammer rule, weights 1/3, det J = x2 y3 (MITC3.py:143-149,467)
            const double det = geo.x2 * geo.y3;
            for (int g = 0; g < 3; g++) area += det * (1.0 / 3.0);
        }
    }
    const double *u = u_e + e * (6 * NN);
    double em[3] = {0.0, 0.0, 0.0}, kp[3] = {0.0, 0.0, 0.0};
    for (int i = 0; i < NN; i++) {
        const double uu = u[6 * i], vv = u[6 * i + 1], tx = u[6 * i + 3], ty = u[6 * i + 4];
        em[0] += dNx[i] * uu;
        em[1] += dNy[i] * vv;
        em[2] += dNy[i] * uu + dNx[i] * vv;
        kp[0] += dNx[i] * ty;
        kp[1] += -dNy[i] * tx;
        kp[2] += -dNx[i] * tx + dNy[i] * ty;
    }
    const double h = pr->h;
    const double z = (location > 0) ? h / 2 : ((location < 0) ? -h / 2 : 0.0);
    double o[3];
    if (strain) {
        for (int a = 0; a < 3; a++) o[a] = em[a] + z * kp[a];
    } else {
        double C[3][3];
        for (int a = 0; a < 9; a++) (&C[0][0])[a] = pr->A[a] / h;    // element.Cm() / h  (:583)
        for (int a = 0; a < 3; a++) {
            double v = 0.0;
            if (parts & FS_STRESS_MEMBRANE) v += C[a][0] * em[0] + C[a][1] * em[1] + C[a][2] * em[2];
            if (parts & FS_STRESS_BENDING) v += C[a][0] * (z * kp[0]) + C[a][1] * (z * kp[1]) + C[a][2] * (z * kp[2]);
            o[a] = v;
        }
    }
    for (int a = 0; a < 3; a++) out[t * 3 + a] = o[a];
    if (weight) weight[t] = __builtin_fabs(area);
}

// ------------------------------------------------------------------------------------------
// Solids.  Workgroup = 32 elements; phase 1: thread (element, Gauss point) -> eps = B u, sigma = C eps
// (SOLID.py:218-266 B, element.C as stored -- not symmetrised here) into LDS; phase 2: thread
// (element, output row) -> Gauss values / their mean / E @ values.
// ------------------------------------------------------------------------------------------
template <int NN, int NG, bool TAB>
__global__ __launch_bounds__(256) void solid_stress_kernel(const double *__restrict__ xyz,
                                                           const int32_t *__restrict__ prop_id,
                                                           const double *__restrict__ props, int32_t n_props,
                                                           const double *__restrict__ u_e, int64_t E,
                                                           SolidRule rule, const double *__restrict__ tab, int ng_tab,
                                                           const double *__restrict__ extrap,
                                                           int mode, double *__restrict__ sig_out,
                                                           double *__restrict__ eps_out,
                                                           double *__restrict__ weight) {
    constexpr int SLOTS = (NG > NN) ? NG : NN;     // threads per element: Gauss points, then output rows
    constexpr int EPB = 256 / SLOTS;
    __shared__ double s_sig[EPB * NG * 6], s_eps[EPB * NG * 6], s_vol[EPB * NG];
    const int tid = threadIdx.x, el = tid / SLOTS, q = tid - el * SLOTS;
    const int64_t e = (int64_t)blockIdx.x * EPB + el;
    const int ng = TAB ? ng_tab : rule.ng;
    const bool live = el < EPB && e < E;
    if (live && q < ng) {
        double N[NN], dxi[NN], deta[NN], dzeta[NN], wq;
        if constexpr (TAB) {
            const double *tq = tab + (int64_t)q * NN * 4;
#pragma unroll
            for (int i = 0; i < NN; i++) { N[i] = tq[4 * i]; dxi[i] = tq[4 * i + 1]; deta[i] = tq[4 * i + 2]; dzeta[i] = tq[4 * i + 3]; }
            wq = tab[(int64_t)ng * NN * 4 + q];
        } else {
            solid_shape<NN>(rule.pt[q][0], rule.pt[q][1], rule.pt[q][2], N, dxi, deta, dzeta);
            wq = rule.w[q];
        }
        const double *x = xyz + e * (3 * NN);
        double J[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll
        for (int i = 0; i < NN; i++)
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const double xc = x[3 * i + c];
                J[0][c] += dxi[i] * xc; J[1][c] += deta[i] * xc; J[2][c] += dzeta[i] * xc;
            }
        const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1], c01 = J[1][0] * J[2][2] - J[1][2] * J[2][0],
                     c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
        const double det = J[0][0] * c00 - J[0][1] * c01 + J[0][2] * c02;
        double inv[3][3];
        inv[0][0] = c00 / det; inv[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) / det;
        inv[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) / det; inv[1][0] = -c01 / det;
        inv[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) / det; inv[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) / det;
        inv[2][0] = c02 / det; inv[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) / det;
        inv[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) / det;
        const double *u = u_e + e * (3 * NN);
        double ep[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < NN; i++) {
            const double gx = inv[0][0] * dxi[i] + inv[0][1] * deta[i] + inv[0][2] * dzeta[i];
            const double gy = inv[1][0] * dxi[i] + inv[1][1] * deta[i] + inv[1][2] * dzeta[i];
            const double gz = inv[2][0] * dxi[i] + inv[2][1] * deta[i] + inv[2][2] * dzeta[i];
            const double ux = u[3 * i], uy = u[3 * i + 1], uz = u[3 * i + 2];
            ep[0] += gx * ux; ep[1] += gy * uy; ep[2] += gz * uz;
            ep[3] += gy * ux + gx * uy; ep[4] += gz * uy + gy * uz; ep[5] += gz * ux + gx * uz;
        }
        const double *C = props + 37 * (int64_t)safe_pid(prop_id[e], n_props);
#pragma unroll
        for (int a = 0; a < 6; a++) {
            double v = 0.0;
#pragma unroll
            for (int b = 0; b < 6; b++) v += C[a * 6 + b] * ep[b];
            s_sig[(el * NG + q) * 6 + a] = v;
            s_eps[(el * NG + q) * 6 + a] = ep[a];
        }
        s_vol[el * NG + q] = __builtin_fabs(det) * wq;
    }
    __syncthreads();
    if (!live) return;
    const double *sg = s_sig + el * NG * 6, *sp = s_eps + el * NG * 6;
    if (weight && q == 0) {                 // stress_recovery.py:1397-1402
        double v = 0.0;
        for (int g = 0; g < ng; g++) v += s_vol[el * NG + g];
        weight[e] = v;
    }
    if (mode == FS_SOLID_STRESS_GP) {
        if (q < ng)
            for (int a = 0; a < 6; a++) {
                if (sig_out) sig_out[(e * ng + q) * 6 + a] = sg[q * 6 + a];
                if (eps_out) eps_out[(e * ng + q) * 6 + a] = sp[q * 6 + a];
            }
    } else if (mode == FS_SOLID_STRESS_MEAN) {   // sig_gp.mean(axis=0)
        if (q < 6) {
            double a = 0.0, b = 0.0;
            for (int g = 0; g < ng; g++) { a += sg[g * 6 + q]; b += sp[g * 6 + q]; }
            if (sig_out) sig_out[e * 6 + q] = a / (double)ng;
            if (eps_out) eps_out[e * 6 + q] = b / (double)ng;
        }
    } else {                                     // E_mat @ sig_gp, E_mat [NN][ng]
        if (q < NN)
            for (int a = 0; a < 6; a++) {
                double va = 0.0, vb = 0.0;
                for (int g = 0; g < ng; g++) {
                    const double w = extrap[q * ng + g];
                    va += w * sg[g * 6 + a]; vb += w * sp[g * 6 + a];
                }
                if (sig_out) sig_out[(e * NN + q) * 6 + a] = va;
                if (eps_out) eps_out[(e * NN + q) * 6 + a] = vb;
            }
    }
}

// node n: sum over its element-node sources (element order) of w * vals, divided by the weight sum
__global__ __launch_bounds__(256) void nodal_average_kernel(const double *__restrict__ vals, int ncomp,
                                                            const double *__restrict__ w_src,
                                                            const int64_t *__restrict__ seg_ptr,
                                                            const int64_t *__restrict__ src, int64_t n_nodes,
                                                            double *__restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_nodes * 6) return;
    const int64_t n = t / 6;
    const int c = (int)(t - n * 6);
    double sum = 0.0, wsum = 0.0;
    if (c < ncomp)
        for (int64_t q = seg_ptr[n]; q < seg_ptr[n + 1]; q++) {
            const int64_t k = src[q];
            const double w = w_src ? w_src[k] : 1.0;
            sum += w * vals[k * ncomp + c];
            wsum += w;
        }
    out[t] = (wsum > 0.0) ? sum / wsum : 0.0;
}

// stress_recovery.py:1087-1296.  out rows: 0 von Mises, 1 s1, 2 s2, 3 s3 (3-D only), 4 tau_max, 5 angle
__global__ __launch_bounds__(256) void stress_invariants_kernel(const double *__restrict__ s, int64_t N, int is3d,
                                                                double *__restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N) return;
    const double sxx = s[t * 6], syy = s[t * 6 + 1], szz = s[t * 6 + 2], txy = s[t * 6 + 3], tyz = s[t * 6 + 4],
                 tzx = s[t * 6 + 5];
    const double d0 = sxx - syy, d1 = syy - szz, d2 = szz - sxx;
    out[t] = sqrt(0.5 * ((d0 * d0 + d1 * d1) + d2 * d2) + 3.0 * ((txy * txy + tyz * tyz) + tzx * tzx));
    if (!is3d) {   // _principal_2d(sigma[:,0], sigma[:,1], sigma[:,3])
        const double avg = (sxx + syy) / 2, diff = (sxx - syy) / 2;
        const double R = sqrt(diff * diff + txy * txy);
        out[N + t] = avg + R; out[2 * N + t] = avg - R; out[3 * N + t] = 0.0; out[4 * N + t] = R;
        out[5 * N + t] = 0.5 * atan2(2 * txy, sxx - syy);
        return;
    }
    const double I1 = (sxx + syy) + szz;
    const double I2 = ((((sxx * syy + syy * szz) + szz * sxx) - txy * txy) - tyz * tyz) - tzx * tzx;
    const double I3 = (((sxx * syy * szz + 2 * txy * tyz * tzx) - sxx * (tyz * tyz)) - syy * (tzx * tzx)) - szz * (txy * txy);
    double p = (I1 * I1) / 3.0 - I2;
    const double q = (2.0 * ((I1 * I1) * I1) / 27.0 - I1 * I2 / 3.0) + I3;
    p = fmax(p, 0.0);
    const double sqp = sqrt(p / 3.0);
    double denom = 2.0 * (p / 3.0) * sqp;
    denom = (denom == 0.0) ? 1.0 : denom;
    const double ca = fmin(fmax(-q / denom, -1.0), 1.0);
    const double phi = acos(ca) / 3.0;
    const double mean = I1 / 3.0;
    const double kPi = 3.141592653589793;
    double a = mean + 2.0 * sqp * cos(phi), b = mean + 2.0 * sqp * cos(phi - 2.0 * kPi / 3.0),
           c = mean + 2.0 * sqp * cos(phi - 4.0 * kPi / 3.0);
    double lo = fmin(a, fmin(b, c)), hi = fmax(a, fmax(b, c));
    double mid;   // the exact middle element, like stacked.sort(axis=-1)
    if ((a <= b && b <= c) || (c <= b && b <= a)) mid = b;
    else if ((b <= a && a <= c) || (c <= a && a <= b)) mid = a;
    else mid = c;
    out[N + t] = hi; out[2 * N + t] = mid; out[3 * N + t] = lo; out[4 * N + t] = (hi - lo) / 2.0;
    out[5 * N + t] = 0.0;
}

// _build_strain_result (:1298-1341): rows 0 e1, 1 e2, 2 gamma_max from (exx, eyy, gxy = eps[:,3])
__global__ __launch_bounds__(256) void strain_invariants_kernel(const double *__restrict__ s, int64_t N,
                                                                double *__restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N) return;
    const double exx = s[t * 6], eyy = s[t * 6 + 1], gxy = s[t * 6 + 3];
    const double avg = (exx + eyy) / 2, diff = (exx - eyy) / 2, hg = gxy / 2;
    const double R = sqrt(diff * diff + hg * hg);
    out[t] = avg + R; out[N + t] = avg - R; out[2 * N + t] = 2 * R;
}

template <int NN, int NG, bool TAB>
void launch_solid_stress(const double *xyz, const int32_t *prop_id, const double *props, int32_t n_props, const double *u_e, int64_t E,
                         const SolidRule &rule, const double *tab, int ng_tab, const double *extrap, int mode,
                         double *sig, double *eps, double *w, hipStream_t s) {
    constexpr int EPB = 256 / ((NG > NN) ? NG : NN);
    hipLaunchKernelGGL((solid_stress_kernel<NN, NG, TAB>), dim3(grid_of(E, EPB)), dim3(256), 0, s, xyz, prop_id, props, n_props,
                       u_e, E, rule, tab, ng_tab, extrap, mode, sig, eps, w);
}

}  // namespace

extern "C" {

int fs_shell_stress_batch(int32_t nn, const double *xyz, const int32_t *prop_id, const fs_shell_prop *props,
                          int32_t n_props, const double *u_e, int64_t E, const double *pts_host, int32_t npts,
                          int32_t location, uint32_t parts, int32_t strain, double *out, double *weight,
                          void *stream) {
    if (E < 0 || n_props <= 0) return fs_fail_(FS_EINVAL, "fs_shell_stress_batch: bad sizes", 0);
    if (nn != 3 && nn != 4) return fs_fail_(FS_EINVAL, "fs_shell_stress_batch: nn must be 3 (MITC3) or 4 (MITC4)", 0);
    if (npts < 1 || npts > 4 || !pts_host) return fs_fail_(FS_EINVAL, "fs_shell_stress_batch: 1..4 points (host array [npts][2])", 0);
    if (!strain && !(parts & (FS_STRESS_MEMBRANE | FS_STRESS_BENDING)))
        return fs_fail_(FS_EINVAL, "fs_shell_stress_batch: parts selects neither membrane nor bending", 0);
    if (E == 0) return FS_OK;
    if (!xyz || !prop_id || !props || !u_e || !out) return fs_fail_(FS_EINVAL, "fs_shell_stress_batch: null argument", 0);
    StressPts pts;
    pts.n = npts;
    for (int k = 0; k < 4; k++) { pts.r[k] = (k < npts) ? pts_host[2 * k] : 0.0; pts.s[k] = (k < npts) ? pts_host[2 * k + 1] : 0.0; }
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(grid_of(E * npts, 256)), block(256);
    if (nn == 4)
        hipLaunchKernelGGL(shell_stress_kernel<4>, grid, block, 0, s, xyz, prop_id, props, n_props, u_e, E, pts, location, parts,
                           strain, out, weight);
    else
        hipLaunchKernelGGL(shell_stress_kernel<3>, grid, block, 0, s, xyz, prop_id, props, n_props, u_e, E, pts, location, parts,
                           strain, out, weight);
    ST_LAUNCH_CHECK("shell_stress_kernel");
    return FS_OK;
}

int fs_solid_stress_batch(int32_t nn, int32_t reduced, const double *xyz, const int32_t *prop_id,
                          const double *props, int32_t n_props, const double *u_e, int64_t E,
                          const double *extrap, int32_t mode, double *sigma, double *eps, double *weight,
                          void *stream) {
    if (E < 0 || n_props <= 0) return fs_fail_(FS_EINVAL, "fs_solid_stress_batch: bad sizes", 0);
    if (nn != 4 && nn != 5 && nn != 6 && nn != 8 && nn != 10 && nn != 13 && nn != 15 && nn != 20)
        return fs_fail_(FS_ENOSYS, "fs_solid_stress_batch: nn must be 4, 5, 6, 8, 10, 13, 15 or 20", 0);
    if (mode != FS_SOLID_STRESS_GP && mode != FS_SOLID_STRESS_MEAN && mode != FS_SOLID_STRESS_NODES)
        return fs_fail_(FS_EINVAL, "fs_solid_stress_batch: bad mode", 0);
    if (mode == FS_SOLID_STRESS_NODES && !extrap)
        return fs_fail_(FS_EINVAL, "fs_solid_stress_batch: FS_SOLID_STRESS_NODES needs the extrapolation matrix", 0);
    if (E == 0) return FS_OK;
    if (!xyz || !prop_id || !props || !u_e || (!sigma && !eps))
        return fs_fail_(FS_EINVAL, "fs_solid_stress_batch: null argument", 0);
    SolidRule rule;
    make_solid_rule(nn <= 8 ? nn : 8, reduced, rule);
    const double *tab = nullptr;
    int ng_tab = 0;
    if (nn > 8) {
        const int rc = fs_solid_table_device_(nn, &tab, &ng_tab);
        if (rc != FS_OK) return rc;
    }
    hipStream_t s = (hipStream_t)stream;
    switch (nn) {
        case 20: launch_solid_stress<20, 27, true>(xyz, prop_id, props, n_props, u_e, E, rule, tab, ng_tab, extrap, mode, sigma, eps, weight, s); break;
        case 15: launch_solid_stress<15, 21, true>(xyz, prop_id, props, n_props, u_e, E, rule, tab, ng_tab, extrap, mode, sigma, eps, weight, s); break;
        case 13: launch_solid_stress<13, 18, true>(xyz, prop_id, props, n_props, u_e, E, rule, tab, ng_tab, extrap, mode, sigma, eps, weight, s); break;
        case 10: launch_solid_stress<10, 4, true>(xyz, prop_id, props, n_props, u_e, E, rule, tab, ng_tab, extrap, mode, sigma, eps, weight, s); break;
        case 8: launch_solid_stress<8, 8, false>(xyz, prop_id, props, n_props, u_e, E, rule, tab, ng_tab, extrap, mode, sigma, eps, weight, s); break;
        case 6: launch_solid_stress<6, 8, false>(xyz, prop_id, props, n_props, u_e, E, rule, tab, ng_tab, extrap, mode, sigma, eps, weight, s); break;
        case 5: launch_solid_stress<5, 8, false>(xyz, prop_id, props, n_props, u_e, E, rule, tab, ng_tab, extrap, mode, sigma, eps, weight, s); break;
        default: launch_solid_stress<4, 8, false>(xyz, prop_id, props, n_props, u_e, E, rule, tab, ng_tab, extrap, mode, sigma, eps, weight, s); break;
    }
    ST_LAUNCH_CHECK("solid_stress_kernel");
    return FS_OK;
}

int fs_nodal_average(const double *vals, int32_t ncomp, const double *w_src, const int64_t *seg_ptr,
                     const int64_t *src, int64_t n_nodes, double *out, void *stream) {
    if (n_nodes < 0 || ncomp < 1 || ncomp > 6) return fs_fail_(FS_EINVAL, "fs_nodal_average: bad sizes", 0);
    if (n_nodes == 0) return FS_OK;
    if (!vals || !seg_ptr || !src || !out) return fs_fail_(FS_EINVAL, "fs_nodal_average: null argument", 0);
    hipLaunchKernelGGL(nodal_average_kernel, dim3(grid_of(n_nodes * 6, 256)), dim3(256), 0, (hipStream_t)stream, vals,
                       ncomp, w_src, seg_ptr, src, n_nodes, out);
    ST_LAUNCH_CHECK("nodal_average_kernel");
    return FS_OK;
}

int fs_stress_invariants(const double *voigt, int64_t N, int32_t is_3d, int32_t strain, double *out, void *stream) {
    if (N < 0) return fs_fail_(FS_EINVAL, "fs_stress_invariants: N < 0", 0);
    if (N == 0) return FS_OK;
    if (!voigt || !out) return fs_fail_(FS_EINVAL, "fs_stress_invariants: null argument", 0);
    if (strain)
        hipLaunchKernelGGL(strain_invariants_kernel, dim3(grid_of(N, 256)), dim3(256), 0, (hipStream_t)stream, voigt, N, out);
    else
        hipLaunchKernelGGL(stress_invariants_kernel, dim3(grid_of(N, 256)), dim3(256), 0, (hipStream_t)stream, voigt, N,
                           is_3d, out);
    ST_LAUNCH_CHECK("stress_invariants_kernel");
    return FS_OK;
}

}  // extern "C"
