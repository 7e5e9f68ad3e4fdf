/*
 * femshell_oracle.c -- scalar C restatement (ORACLE) of efirvida/fem-shell's MITC4+ element
 * matrices and CSR assembly.  TEST INFRASTRUCTURE ONLY (see femshell_oracle.h).
 *
 * Build: gcc -O2 -ffp-contract=off -mfma -fopenmp -shared -fPIC (oracle/Makefile).
 * -ffp-contract=off is REQUIRED: the geometry stage reproduces the reference's rounding
 * sequence (NumPy 2.2 + OpenBLAS 0.3.29 as measured in the build container):
 *   np.dot on contiguous length-3/4 vectors = sequential FMA chain          -> dot3(), dotc4()
 *   np.dot(dN, xl[:,k]) (strided operand, n=4) = fma(a0,b0,a2*b2)+fma(a1,b1,a3*b3) -> dots4()
 *   np.cross = rounded products then subtraction (no FMA)                   -> cross3()
 *   np.linalg.norm(v) = sqrt(dot(v,v))
 * Everything downstream of the geometry stage follows the reference's operation structure with
 * plain dense loops (agreement at the eps level is all that is required there).
 *
 * Citations are file:line relative to /root/reference/src/fem_shell/.
 */
#include "femshell_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---------- rounding-pinned primitives ---------- */
static inline double dot3(const double *a, const double *b) {
    double s = a[0] * b[0];
    s = fma(a[1], b[1], s);
    s = fma(a[2], b[2], s);
    return s;
}
static inline double dots4(const double *a, const double *b, int sb) { /* b strided */
    double m2 = a[2] * b[2 * sb], m3 = a[3] * b[3 * sb];
    double t1 = fma(a[0], b[0], m2), t2 = fma(a[1], b[sb], m3);
    return t1 + t2;
}
static inline double dotc4(const double *a, const double *b) { /* both contiguous */
    double s = a[0] * b[0];
    s = fma(a[1], b[1], s);
    s = fma(a[2], b[2], s);
    s = fma(a[3], b[3], s);
    return s;
}
static inline void cross3(const double *a, const double *b, double *c) {
    double p, q;
    p = a[1] * b[2]; q = a[2] * b[1]; c[0] = p - q;
    p = a[2] * b[0]; q = a[0] * b[2]; c[1] = p - q;
    p = a[0] * b[1]; q = a[1] * b[0]; c[2] = p - q;
}
static inline double norm3(const double *a) { return sqrt(dot3(a, a)); }

typedef struct {
    double x[4][3];
    double e1[3], e2[3], e3[3];
    double xl[4][2];
    double aA, aB, aC, aD, aE;
    int status;
} geom_t;

static const double GP = 0.57735026918962584; /* 1/sqrt(3) as computed by 1.0/np.sqrt(3.0) */

static void dshape(double xi, double eta, double *dxi, double *deta) { /* MITC4.py:141-155 */
    dxi[0] = -0.25 * (1.0 - eta); dxi[1] = +0.25 * (1.0 - eta);
    dxi[2] = +0.25 * (1.0 + eta); dxi[3] = -0.25 * (1.0 + eta);
    deta[0] = -0.25 * (1.0 - xi); deta[1] = -0.25 * (1.0 + xi);
    deta[2] = +0.25 * (1.0 + xi); deta[3] = +0.25 * (1.0 - xi);
}
static void shape(double xi, double eta, double *N) { /* MITC4.py:133-139 */
    N[0] = 0.25 * (1.0 - xi) * (1.0 - eta); N[1] = 0.25 * (1.0 + xi) * (1.0 - eta);
    N[2] = 0.25 * (1.0 + xi) * (1.0 + eta); N[3] = 0.25 * (1.0 - xi) * (1.0 + eta);
}

/* MITC4.py:436-476 _compute_local_coordinates */
static void local_frame(geom_t *g) {
    double (*n)[3] = g->x;
    double v1[3], v2[3], c[3], nrm[2][3];
    int nn = 0;
    for (int k = 0; k < 3; k++) { v1[k] = n[1][k] - n[0][k]; v2[k] = n[2][k] - n[0][k]; }
    cross3(v1, v2, c);
    double l = norm3(c);
    if (l > 1e-12) { for (int k = 0; k < 3; k++) nrm[nn][k] = c[k] / l; nn++; }
    for (int k = 0; k < 3; k++) { v1[k] = n[2][k] - n[0][k]; v2[k] = n[3][k] - n[0][k]; }
    cross3(v1, v2, c);
    l = norm3(c);
    if (l > 1e-12) { for (int k = 0; k < 3; k++) nrm[nn][k] = c[k] / l; nn++; }
    double e3[3];
    if (nn == 2) { for (int k = 0; k < 3; k++) e3[k] = (nrm[0][k] + nrm[1][k]) / 2.0; }
    else if (nn == 1) { for (int k = 0; k < 3; k++) e3[k] = nrm[0][k] / 1.0; }
    else { e3[0] = 0; e3[1] = 0; e3[2] = 1; }
    if (nn) { l = norm3(e3); for (int k = 0; k < 3; k++) e3[k] = e3[k] / l; }
    double e1[3];
    for (int k = 0; k < 3; k++) e1[k] = n[1][k] - n[0][k];
    double d = dot3(e1, e3);
    for (int k = 0; k < 3; k++) { double t = d * e3[k]; e1[k] = e1[k] - t; }
    if (norm3(e1) < 1e-12) {
        for (int k = 0; k < 3; k++) e1[k] = n[2][k] - n[0][k];
        d = dot3(e1, e3);
        for (int k = 0; k < 3; k++) { double t = d * e3[k]; e1[k] = e1[k] - t; }
    }
    l = norm3(e1);
    for (int k = 0; k < 3; k++) e1[k] = e1[k] / l;
    double e2[3];
    cross3(e3, e1, e2);
    l = norm3(e2);
    for (int k = 0; k < 3; k++) e2[k] = e2[k] / l;
    memcpy(g->e1, e1, sizeof e1); memcpy(g->e2, e2, sizeof e2); memcpy(g->e3, e3, sizeof e3);
    for (int i = 0; i < 4; i++) { g->xl[i][0] = dot3(n[i], e1); g->xl[i][1] = dot3(n[i], e2); }
    if (!(l == l) || !(g->xl[0][0] == g->xl[0][0])) g->status |= 1; /* NaN frame */
}

static void solve3(const double A[3][3], const double *b, double *x, int *sing) {
    /* np.linalg.solve on 3x3 (MITC4.py:187-188) -> closed-form adjugate; eps-level only */
    double c00 = A[1][1] * A[2][2] - A[1][2] * A[2][1];
    double c01 = A[1][2] * A[2][0] - A[1][0] * A[2][2];
    double c02 = A[1][0] * A[2][1] - A[1][1] * A[2][0];
    double det = A[0][0] * c00 + A[0][1] * c01 + A[0][2] * c02;
    if (det == 0.0) { *sing = 1; return; }
    double c10 = A[0][2] * A[2][1] - A[0][1] * A[2][2];
    double c11 = A[0][0] * A[2][2] - A[0][2] * A[2][0];
    double c12 = A[0][1] * A[2][0] - A[0][0] * A[2][1];
    double c20 = A[0][1] * A[1][2] - A[0][2] * A[1][1];
    double c21 = A[0][2] * A[1][0] - A[0][0] * A[1][2];
    double c22 = A[0][0] * A[1][1] - A[0][1] * A[1][0];
    x[0] = (c00 * b[0] + c10 * b[1] + c20 * b[2]) / det;
    x[1] = (c01 * b[0] + c11 * b[1] + c21 * b[2]) / det;
    x[2] = (c02 * b[0] + c12 * b[1] + c22 * b[2]) / det;
}

/* MITC4.py:157-208 characteristic vectors + membrane coefficients */
static void membrane_coeffs(geom_t *g) {
    double (*n)[3] = g->x;
    double xr[3], xs[3], xd[3], nv[3], nh[3];
    for (int k = 0; k < 3; k++) {
        xr[k] = 0.25 * (((-n[0][k] + n[1][k]) + n[2][k]) - n[3][k]);
        xs[k] = 0.25 * (((-n[0][k] - n[1][k]) + n[2][k]) + n[3][k]);
        xd[k] = 0.25 * (((n[0][k] - n[1][k]) + n[2][k]) - n[3][k]);
    }
    cross3(xr, xs, nv);
    double nn = norm3(nv);
    if (nn < 1e-12) {
        double a[3], b[3];
        for (int k = 0; k < 3; k++) { a[k] = n[1][k] - n[0][k]; b[k] = n[2][k] - n[0][k]; }
        cross3(a, b, nv);
        nn = norm3(nv);
    }
    for (int k = 0; k < 3; k++) nh[k] = nv[k] / nn;
    double A[3][3] = {{dot3(xr, xr), dot3(xr, xs), dot3(xr, nh)},
                      {dot3(xs, xr), dot3(xs, xs), dot3(xs, nh)},
                      {dot3(nh, xr), dot3(nh, xs), dot3(nh, nh)}};
    double b1[3] = {1, 0, 0}, b2[3] = {0, 1, 0}, cr[3], cs[3], mr[3], ms[3];
    int sing = 0;
    solve3(A, b1, cr, &sing);
    solve3(A, b2, cs, &sing);
    if (!sing) {
        for (int k = 0; k < 3; k++) {
            mr[k] = cr[0] * xr[k] + cr[1] * xs[k] + cr[2] * nh[k];
            ms[k] = cs[0] * xr[k] + cs[1] * xs[k] + cs[2] * nh[k];
        }
    } else {
        double drr = dot3(xr, xr), dss = dot3(xs, xs);
        for (int k = 0; k < 3; k++) { mr[k] = xr[k] / drr; ms[k] = xs[k] / dss; }
    }
    double c_r = dot3(xd, mr), c_s = dot3(xd, ms);
    double d = c_r * c_r + c_s * c_s - 1.0;
    if (fabs(d) < 1e-12) d = (d >= 0) ? 1e-12 : -1e-12;
    g->aA = c_r * (c_r - 1.0) / (2.0 * d);
    g->aB = c_r * (c_r + 1.0) / (2.0 * d);
    g->aC = c_s * (c_s - 1.0) / (2.0 * d);
    g->aD = c_s * (c_s + 1.0) / (2.0 * d);
    g->aE = -2.0 * c_r * c_s / d;
    if (!(g->aA == g->aA)) g->status |= 2;
}

static void geom_init(geom_t *g, const double *xyz) {
    memcpy(g->x, xyz, 12 * sizeof(double));
    g->status = 0;
    local_frame(g);
    membrane_coeffs(g);
}

/* MITC4.py:614-632 J */
static double jac(const geom_t *g, double xi, double eta, double J[2][2]) {
    double dxi[4], deta[4];
    dshape(xi, eta, dxi, deta);
    const double *xl = &g->xl[0][0];
    J[0][0] = dots4(dxi, xl, 2); J[0][1] = dots4(dxi, xl + 1, 2);
    J[1][0] = dots4(deta, xl, 2); J[1][1] = dots4(deta, xl + 1, 2);
    double p = J[0][0] * J[1][1], q = J[0][1] * J[1][0];
    return p - q;
}
static void inv2(const double J[2][2], double det, double Ji[2][2]) {
    Ji[0][0] = J[1][1] / det; Ji[0][1] = -J[0][1] / det;
    Ji[1][0] = -J[1][0] / det; Ji[1][1] = J[0][0] / det;
}
/* MITC4.py:642-657 _get_dH */
static double dH(const geom_t *g, double xi, double eta, double *dNdx, double *dNdy) {
    double J[2][2], Ji[2][2], dxi[4], deta[4];
    double det = jac(g, xi, eta, J);
    inv2(J, det, Ji);
    dshape(xi, eta, dxi, deta);
    for (int i = 0; i < 4; i++) {
        dNdx[i] = Ji[0][0] * dxi[i] + Ji[0][1] * deta[i];
        dNdy[i] = Ji[1][0] * dxi[i] + Ji[1][1] * deta[i];
    }
    return det;
}

/* MITC4.py:589-612 J3D + :261-321 covariant rows. comp: 0 rr, 1 ss, 2 rs */
static void cov_row(const geom_t *g, double xi, double eta, int comp, double *row) {
    double dxi[4], deta[4], gr[3] = {0, 0, 0}, gs[3] = {0, 0, 0};
    dshape(xi, eta, dxi, deta);
    for (int i = 0; i < 4; i++)
        for (int k = 0; k < 3; k++) {
            double t = dxi[i] * g->x[i][k]; gr[k] = gr[k] + t;
            t = deta[i] * g->x[i][k]; gs[k] = gs[k] + t;
        }
    double grl[3] = {dot3(gr, g->e1), dot3(gr, g->e2), dot3(gr, g->e3)};
    double gsl[3] = {dot3(gs, g->e1), dot3(gs, g->e2), dot3(gs, g->e3)};
    memset(row, 0, 24 * sizeof(double));
    for (int i = 0; i < 4; i++)
        for (int k = 0; k < 3; k++) {
            if (comp == 0) row[6 * i + k] = grl[k] * dxi[i];
            else if (comp == 1) row[6 * i + k] = gsl[k] * deta[i];
            else row[6 * i + k] = 0.5 * (grl[k] * deta[i] + gsl[k] * dxi[i]);
        }
}

/* MITC4.py:351-392 B_m_MITC4_plus -> Bm[3][24] */
static void Bm_plus(const geom_t *g, double xi, double eta, double Bm[3][24]) {
    double rrA[24], rrB[24], ssC[24], ssD[24], rsE[24];
    cov_row(g, 0.0, 1.0, 0, rrA);
    cov_row(g, 0.0, -1.0, 0, rrB);
    cov_row(g, 1.0, 0.0, 1, ssC);
    cov_row(g, -1.0, 0.0, 1, ssD);
    cov_row(g, 0.0, 0.0, 2, rsE);
    double aA = g->aA, aB = g->aB, aC = g->aC, aD = g->aD, aE = g->aE, r = xi, s = eta;
    double Bc[3][24];
    for (int c = 0; c < 24; c++) {
        Bc[0][c] = (0.5 * (1.0 - 2.0 * aA + s + 2.0 * aA * s * s)) * rrA[c] +
                   (0.5 * (1.0 - 2.0 * aB - s + 2.0 * aB * s * s)) * rrB[c] +
                   aC * (-1.0 + s * s) * ssC[c] + aD * (-1.0 + s * s) * ssD[c] +
                   aE * (-1.0 + s * s) * rsE[c];
        Bc[1][c] = aA * (-1.0 + r * r) * rrA[c] + aB * (-1.0 + r * r) * rrB[c] +
                   (0.5 * (1.0 - 2.0 * aC + r + 2.0 * aC * r * r)) * ssC[c] +
                   (0.5 * (1.0 - 2.0 * aD - r + 2.0 * aD * r * r)) * ssD[c] +
                   aE * (-1.0 + r * r) * rsE[c];
        double brs = 0.25 * (r + 4.0 * aA * r * s) * rrA[c] + 0.25 * (-r + 4.0 * aB * r * s) * rrB[c] +
                     0.25 * (s + 4.0 * aC * r * s) * ssC[c] + 0.25 * (-s + 4.0 * aD * r * s) * ssD[c] +
                     (1.0 + aE * r * s) * rsE[c];
        Bc[2][c] = 2.0 * brs;
    }
    /* MITC4.py:323-349 covariant -> cartesian */
    double J[2][2], Ji[2][2], T[3][3];
    double det = jac(g, xi, eta, J);
    if (fabs(det) < 1e-12) {
        memset(T, 0, sizeof T); T[0][0] = T[1][1] = T[2][2] = 1.0;
    } else {
        inv2(J, det, Ji);
        double a = Ji[0][0], b = Ji[0][1], c = Ji[1][0], d = Ji[1][1];
        T[0][0] = a * a; T[0][1] = c * c; T[0][2] = a * c;
        T[1][0] = b * b; T[1][1] = d * d; T[1][2] = b * d;
        T[2][0] = 2.0 * a * b; T[2][1] = 2.0 * c * d; T[2][2] = a * d + b * c;
    }
    for (int rr = 0; rr < 3; rr++)
        for (int c = 0; c < 24; c++) {
            double sum = 0;
            for (int k = 0; k < 3; k++) sum = fma(T[rr][k], Bc[k][c], sum);
            Bm[rr][c] = sum;
        }
}

/* MITC4.py:756-772 standard B_m */
static void Bm_std(const double *dNdx, const double *dNdy, double Bm[3][24]) {
    memset(Bm, 0, 72 * sizeof(double));
    for (int i = 0; i < 4; i++) {
        Bm[0][6 * i] = dNdx[i]; Bm[1][6 * i + 1] = dNdy[i];
        Bm[2][6 * i] = dNdy[i]; Bm[2][6 * i + 1] = dNdx[i];
    }
}
/* MITC4.py:774-790 B_kappa */
static void Bkappa(const double *dNdx, const double *dNdy, double Bk[3][24]) {
    memset(Bk, 0, 72 * sizeof(double));
    for (int i = 0; i < 4; i++) {
        Bk[0][6 * i + 4] = dNdx[i]; Bk[1][6 * i + 3] = -dNdy[i];
        Bk[2][6 * i + 4] = dNdy[i]; Bk[2][6 * i + 3] = -dNdx[i];
    }
}
/* MITC4.py:792-878 B_gamma_MITC4 */
static void Bgamma(const geom_t *g, double xi, double eta, double Bs[2][24]) {
    const double (*xl)[2] = g->xl;
    double dx34 = xl[2][0] - xl[3][0], dy34 = xl[2][1] - xl[3][1];
    double dx21 = xl[1][0] - xl[0][0], dy21 = xl[1][1] - xl[0][1];
    double dx32 = xl[2][0] - xl[1][0], dy32 = xl[2][1] - xl[1][1];
    double dx41 = xl[3][0] - xl[0][0], dy41 = xl[3][1] - xl[0][1];
    double G[4][12];
    memset(G, 0, sizeof G);
    G[0][0] = -0.5; G[0][1] = -dy41 * 0.25; G[0][2] = dx41 * 0.25;
    G[0][9] = 0.5; G[0][10] = -dy41 * 0.25; G[0][11] = dx41 * 0.25;
    G[1][0] = -0.5; G[1][1] = -dy21 * 0.25; G[1][2] = dx21 * 0.25;
    G[1][3] = 0.5; G[1][4] = -dy21 * 0.25; G[1][5] = dx21 * 0.25;
    G[2][3] = -0.5; G[2][4] = -dy32 * 0.25; G[2][5] = dx32 * 0.25;
    G[2][6] = 0.5; G[2][7] = -dy32 * 0.25; G[2][8] = dx32 * 0.25;
    G[3][6] = 0.5; G[3][7] = -dy34 * 0.25; G[3][8] = dx34 * 0.25;
    G[3][9] = -0.5; G[3][10] = -dy34 * 0.25; G[3][11] = dx34 * 0.25;
    double Ax = ((-xl[0][0] + xl[1][0]) + xl[2][0]) - xl[3][0];
    double Bx = ((xl[0][0] - xl[1][0]) + xl[2][0]) - xl[3][0];
    double Cx = ((-xl[0][0] - xl[1][0]) + xl[2][0]) + xl[3][0];
    double Ay = ((-xl[0][1] + xl[1][1]) + xl[2][1]) - xl[3][1];
    double By = ((xl[0][1] - xl[1][1]) + xl[2][1]) - xl[3][1];
    double Cy = ((-xl[0][1] - xl[1][1]) + xl[2][1]) + xl[3][1];
    double alph = atan2(Ay, Ax);
    double beta = M_PI / 2.0 - atan2(Cx, Cy);
    double Rot[2][2] = {{sin(beta), -sin(alph)}, {-cos(beta), cos(alph)}};
    double Ms[2][4] = {{0, 1.0 - eta, 0, 1.0 + eta}, {1.0 - xi, 0, 1.0 + xi, 0}};
    double Bsv[2][12];
    for (int r = 0; r < 2; r++)
        for (int j = 0; j < 12; j++) {
            double s = 0;
            for (int k = 0; k < 4; k++) s = fma(Ms[r][k], G[k][j], s);
            Bsv[r][j] = s;
        }
    double J[2][2];
    double det = jac(g, xi, eta, J);
    double p0 = Cx + xi * Bx, p1 = Cy + xi * By;
    double r1 = sqrt(fma(p1, p1, p0 * p0));
    double q0 = Ax + eta * Bx, q1 = Ay + eta * By;
    double r2 = sqrt(fma(q1, q1, q0 * q0));
    for (int j = 0; j < 12; j++) {
        Bsv[0][j] = Bsv[0][j] * r1 / (8.0 * det);
        Bsv[1][j] = Bsv[1][j] * r2 / (8.0 * det);
    }
    memset(Bs, 0, 48 * sizeof(double));
    for (int r = 0; r < 2; r++)
        for (int i = 0; i < 4; i++)
            for (int k = 0; k < 3; k++)
                Bs[r][6 * i + 2 + k] = fma(Rot[r][1], Bsv[1][3 * i + k], Rot[r][0] * Bsv[0][3 * i + k]);
}

static void gauss_point(int gp, double *xi, double *eta) { /* MITC4.py:87-88 */
    static const double sx[4] = {-1, +1, +1, -1}, sy[4] = {-1, -1, +1, +1};
    *xi = sx[gp] * GP; *eta = sy[gp] * GP;
}

/* acc[24][24] += s * Ba^T (ra x 24) C (ra x rb) Bb (rb x 24) */
static void btcb(double *acc, double s, const double *Ba, int ra, const double *C, const double *Bb,
                 int rb, int ncb) {
    /* Ba: [ra][24]; Bb: [rb][ncb]; acc: [24][ncb] */
    for (int i = 0; i < 24; i++) {
        double t[6];
        int any = 0;
        for (int b = 0; b < rb; b++) {
            double v = 0;
            for (int a = 0; a < ra; a++) v += (s * Ba[a * 24 + i]) * C[a * rb + b];
            t[b] = v; any |= (v != 0.0);
        }
        if (!any) continue;
        for (int j = 0; j < ncb; j++) {
            double v = 0;
            for (int b = 0; b < rb; b++) v += t[b] * Bb[b * ncb + j];
            acc[i * ncb + j] += v;
        }
    }
}

/* local-axis parts of MITC4.K: k_m (:934-944), _k_bending_shear (:946-1000), k_drill (:1002-1016),
 * composite _coupling_stiffness (MITC4_composite.py:210-248; nodal-only form) */
static void k_parts(const geom_t *g, const fso_shell_prop *p, double *km, double *kbs, double *kd,
                    double *kc) {
    double Knn[576], Knb[48], Kbb[4];
    memset(km, 0, 576 * sizeof(double)); memset(Knn, 0, sizeof Knn);
    memset(Knb, 0, sizeof Knb); memset(Kbb, 0, sizeof Kbb);
    memset(kd, 0, 576 * sizeof(double));
    if (kc) memset(kc, 0, 576 * sizeof(double));
    for (int gp = 0; gp < 4; gp++) {
        double xi, eta, dNdx[4], dNdy[4], N[4];
        gauss_point(gp, &xi, &eta);
        double det = dH(g, xi, eta, dNdx, dNdy);
        double w = 1.0 * det;
        double Bm[3][24], Bk[3][24], Bs[2][24];
        Bm_plus(g, xi, eta, Bm);
        btcb(km, w, &Bm[0][0], 3, p->A, &Bm[0][0], 3, 24);
        Bkappa(dNdx, dNdy, Bk);
        /* bubble: MITC4.py:214-222, 891-907, 880-889 */
        double J[2][2], Ji[2][2];
        jac(g, xi, eta, J); inv2(J, det, Ji);
        double Nb = (1.0 - xi * xi) * (1.0 - eta * eta);
        double dNbxi = -2.0 * xi * (1.0 - eta * eta), dNbeta = -2.0 * eta * (1.0 - xi * xi);
        double dNbx = Ji[0][0] * dNbxi + Ji[0][1] * dNbeta, dNby = Ji[1][0] * dNbxi + Ji[1][1] * dNbeta;
        double Bkb[3][2] = {{0, dNbx}, {-dNby, 0}, {-dNbx, dNby}};
        double Bsb[2][2] = {{0, Nb}, {-Nb, 0}};
        btcb(Knn, w, &Bk[0][0], 3, p->D, &Bk[0][0], 3, 24);
        btcb(Knb, w, &Bk[0][0], 3, p->D, &Bkb[0][0], 3, 2);
        Bgamma(g, xi, eta, Bs);
        btcb(Knn, w, &Bs[0][0], 2, p->Cs, &Bs[0][0], 2, 24);
        btcb(Knb, w, &Bs[0][0], 2, p->Cs, &Bsb[0][0], 2, 2);
        for (int a = 0; a < 2; a++)
            for (int b = 0; b < 2; b++) {
                double v = 0;
                for (int r = 0; r < 3; r++)
                    for (int c = 0; c < 3; c++) v += w * Bkb[r][a] * p->D[r * 3 + c] * Bkb[c][b];
                for (int r = 0; r < 2; r++)
                    for (int c = 0; c < 2; c++) v += w * Bsb[r][a] * p->Cs[r * 2 + c] * Bsb[c][b];
                Kbb[a * 2 + b] += v;
            }
        /* drilling: MITC4.py:909-928, 1002-1016 */
        shape(xi, eta, N);
        double Bd[24];
        memset(Bd, 0, sizeof Bd);
        for (int i = 0; i < 4; i++) {
            Bd[6 * i] = -0.5 * dNdy[i]; Bd[6 * i + 1] = 0.5 * dNdx[i]; Bd[6 * i + 5] = -N[i];
        }
        for (int i = 0; i < 24; i++)
            for (int j = 0; j < 24; j++) kd[i * 24 + j] += w * p->k_drill * (Bd[i] * Bd[j]);
        if (kc && (p->flags & FSO_PROP_COUPLED)) {
            double tmp[576];
            memset(tmp, 0, sizeof tmp);
            btcb(tmp, w, &Bm[0][0], 3, p->B, &Bk[0][0], 3, 24);
            for (int i = 0; i < 24; i++)
                for (int j = 0; j < 24; j++) kc[i * 24 + j] += tmp[i * 24 + j] + tmp[j * 24 + i];
        }
    }
    /* condensation MITC4.py:994-998 */
    double det = Kbb[0] * Kbb[3] - Kbb[1] * Kbb[2];
    memcpy(kbs, Knn, sizeof Knn);
    if (det != 0.0 && det == det) {
        double Ki[4] = {Kbb[3] / det, -Kbb[1] / det, -Kbb[2] / det, Kbb[0] / det};
        for (int i = 0; i < 24; i++) {
            double h0 = Knb[i * 2] * Ki[0] + Knb[i * 2 + 1] * Ki[2];
            double h1 = Knb[i * 2] * Ki[1] + Knb[i * 2 + 1] * Ki[3];
            for (int j = 0; j < 24; j++) kbs[i * 24 + j] -= h0 * Knb[j * 2] + h1 * Knb[j * 2 + 1];
        }
    }
}

/* T^T K T with T = blockdiag_8([e1;e2;e3]) (MITC4.py:634-640, 1032-1033) */
static void rotate_to_global(const geom_t *g, const double *Kl, double *Kg) {
    double R[3][3];
    memcpy(R[0], g->e1, 24); memcpy(R[1], g->e2, 24); memcpy(R[2], g->e3, 24);
    for (int bi = 0; bi < 8; bi++)
        for (int bj = 0; bj < 8; bj++) {
            double X[3][3], Y[3][3];
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++) X[a][b] = Kl[(3 * bi + a) * 24 + 3 * bj + b];
            /* Y = R^T X */
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++) {
                    double s = 0;
                    for (int k = 0; k < 3; k++) s += R[k][a] * X[k][b];
                    Y[a][b] = s;
                }
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++) {
                    double s = 0;
                    for (int k = 0; k < 3; k++) s += Y[a][k] * R[k][b];
                    Kg[(3 * bi + a) * 24 + 3 * bj + b] = s;
                }
        }
}
static void rotate_vec(const geom_t *g, const double *u, double *ul, int to_local) {
    double R[3][3];
    memcpy(R[0], g->e1, 24); memcpy(R[1], g->e2, 24); memcpy(R[2], g->e3, 24);
    for (int b = 0; b < 8; b++)
        for (int a = 0; a < 3; a++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += (to_local ? R[a][k] : R[k][a]) * u[3 * b + k];
            ul[3 * b + a] = s;
        }
}

int fso_mitc4_geom(const double *xyz, double *e, double *xl, double *detJ, double *acoef) {
    geom_t g;
    geom_init(&g, xyz);
    memcpy(e, g.e1, 24); memcpy(e + 3, g.e2, 24); memcpy(e + 6, g.e3, 24);
    memcpy(xl, g.xl, 64);
    for (int gp = 0; gp < 4; gp++) {
        double xi, eta, J[2][2];
        gauss_point(gp, &xi, &eta);
        detJ[gp] = jac(&g, xi, eta, J);
    }
    acoef[0] = g.aA; acoef[1] = g.aB; acoef[2] = g.aC; acoef[3] = g.aD; acoef[4] = g.aE;
    return g.status;
}

int fso_mitc4_K_parts(const double *xyz, const fso_shell_prop *p, double *km, double *kbs,
                      double *kdrill, double *kcoup) {
    geom_t g;
    geom_init(&g, xyz);
    k_parts(&g, p, km, kbs, kdrill, kcoup);
    return g.status;
}

static void k_local(const geom_t *g, const fso_shell_prop *p, double *Kl, double *kbs_out,
                    double *kd_out, double *kc_out) {
    double km[576], kbs[576], kd[576], kc[576];
    k_parts(g, p, km, kbs, kd, kc);
    for (int i = 0; i < 576; i++) {
        Kl[i] = km[i] + kbs[i] + kd[i];
        if (p->flags & FSO_PROP_COUPLED) Kl[i] += kc[i];
    }
    if (kbs_out) memcpy(kbs_out, kbs, sizeof kbs);
    if (kd_out) memcpy(kd_out, kd, sizeof kd);
    if (kc_out) memcpy(kc_out, kc, sizeof kc);
}

int fso_mitc4_K(const double *xyz, const fso_shell_prop *p, double *Ke) {
    geom_t g;
    geom_init(&g, xyz);
    double Kl[576], Kg[576];
    k_local(&g, p, Kl, 0, 0, 0);
    rotate_to_global(&g, Kl, Kg);
    for (int i = 0; i < 24; i++)
        for (int j = 0; j < 24; j++) Ke[i * 24 + j] = 0.5 * (Kg[i * 24 + j] + Kg[j * 24 + i]);
    for (int gp = 0; gp < 4; gp++) {
        double xi, eta, J[2][2];
        gauss_point(gp, &xi, &eta);
        double d = jac(&g, xi, eta, J);
        if (!(d > 0.0)) g.status |= 4;
    }
    return g.status;
}

int fso_mitc4_M(const double *xyz, const fso_shell_prop *p, double *Me) {
    geom_t g;
    geom_init(&g, xyz);
    double Ml[576];
    memset(Ml, 0, sizeof Ml);
    for (int gp = 0; gp < 4; gp++) {
        double xi, eta, J[2][2], N[4];
        gauss_point(gp, &xi, &eta);
        double det = jac(&g, xi, eta, J);
        shape(xi, eta, N);
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) {
                double vt = N[i] * N[j] * p->m_t * 1.0 * det;
                double vr = N[i] * N[j] * p->m_r * 1.0 * det;
                for (int k = 0; k < 3; k++) Ml[(6 * i + k) * 24 + 6 * j + k] += vt;
                for (int k = 3; k < 6; k++) Ml[(6 * i + k) * 24 + 6 * j + k] += vr;
            }
    }
    rotate_to_global(&g, Ml, Me);
    return g.status;
}

/* MITC4.py:1067-1341 (+ composite overrides MITC4_composite.py:499-621) */
int fso_mitc4_NL(const double *xyz, const fso_shell_prop *p, const double *u, double *KT, double *f) {
    geom_t g;
    geom_init(&g, xyz);
    double K0[576], kbs[576], kd[576], kc[576], ul[24];
    k_local(&g, p, K0, kbs, kd, kc);
    rotate_vec(&g, u, ul, 1);
    if (!p->nonlinear) {
        if (KT) rotate_to_global(&g, K0, KT); /* not symmetrised: MITC4.py:1259-1263 */
        if (f) {
            double fl[24];
            for (int i = 0; i < 24; i++) {
                double s = 0;
                for (int j = 0; j < 24; j++) s += K0[i * 24 + j] * ul[j];
                fl[i] = s;
            }
            rotate_vec(&g, fl, f, 0);
        }
        return g.status;
    }
    double KL[576], Ks[576], fl[24];
    memset(KL, 0, sizeof KL); memset(Ks, 0, sizeof Ks);
    /* sigma at (0,0): MITC4.py:1227-1240 ; composite MITC4_composite.py:499-508 */
    double sig[3];
    {
        double dNdx[4], dNdy[4], Bm[3][24], Bk[3][24], eps[3] = {0, 0, 0}, kap[3] = {0, 0, 0};
        dH(&g, 0.0, 0.0, dNdx, dNdy);
        Bm_std(dNdx, dNdy, Bm); Bkappa(dNdx, dNdy, Bk);
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 24; c++) { eps[r] += Bm[r][c] * ul[c]; kap[r] += Bk[r][c] * ul[c]; }
        for (int r = 0; r < 3; r++) {
            double s = 0;
            for (int c = 0; c < 3; c++) s += p->C_raw[r * 3 + c] * eps[c];
            sig[r] = s;
        }
        /* laminate: (A eps + B kappa)/h. For isotropic rows B == 0 and C_raw == A/h up to
         * rounding; the property table sets prop.B != 0 only for laminates, and C_raw*eps is
         * replaced by (A eps + B kappa)/h when flag bit 1 (is_laminate) is set. */
        if (p->flags & FSO_PROP_LAMINATE) {
            for (int r = 0; r < 3; r++) {
                double s = 0;
                for (int c = 0; c < 3; c++) s += p->A[r * 3 + c] * eps[c];
                double s2 = 0;
                for (int c = 0; c < 3; c++) s2 += p->B[r * 3 + c] * kap[c];
                sig[r] = (s + s2) / p->h;
            }
        }
    }
    if (f) {
        for (int i = 0; i < 24; i++) {
            double s = 0, s2 = 0, s3 = 0;
            for (int j = 0; j < 24; j++) {
                s += kbs[i * 24 + j] * ul[j]; s2 += kd[i * 24 + j] * ul[j];
                if (p->flags & FSO_PROP_COUPLED) s3 += kc[i * 24 + j] * ul[j];
            }
            fl[i] = s + s2 + s3;
        }
    }
    for (int gp = 0; gp < 4; gp++) {
        double xi, eta, dNdx[4], dNdy[4];
        gauss_point(gp, &xi, &eta);
        double det = dH(&g, xi, eta, dNdx, dNdy);
        /* H: global displacement components, local derivatives (MITC4.py:1092-1107) */
        double H[3][3];
        memset(H, 0, sizeof H);
        for (int c = 0; c < 3; c++) {
            double un[4] = {u[c], u[6 + c], u[12 + c], u[18 + c]};
            H[c][0] = dotc4(un, dNdx); H[c][1] = dotc4(un, dNdy);
        }
        double BL[3][24], BNL[3][24], BG[6][24];
        memset(BL, 0, sizeof BL); memset(BNL, 0, sizeof BNL); memset(BG, 0, sizeof BG);
        for (int i = 0; i < 4; i++) {
            int c = 6 * i;
            BL[0][c] = dNdx[i]; BL[1][c + 1] = dNdy[i]; BL[2][c] = dNdy[i]; BL[2][c + 1] = dNdx[i];
            for (int k = 0; k < 3; k++) {
                BNL[0][c + k] = H[k][0] * dNdx[i];
                BNL[1][c + k] = H[k][1] * dNdy[i];
                BNL[2][c + k] = H[k][0] * dNdy[i] + H[k][1] * dNdx[i];
                BG[2 * k][c + k] = dNdx[i]; BG[2 * k + 1][c + k] = dNdy[i];
            }
        }
        double w = det * p->h;
        if (KT) {
            btcb(KL, w, &BL[0][0], 3, p->C_raw, &BNL[0][0], 3, 24);
            btcb(KL, w, &BNL[0][0], 3, p->C_raw, &BL[0][0], 3, 24);
            btcb(KL, w, &BNL[0][0], 3, p->C_raw, &BNL[0][0], 3, 24);
            double S[36];
            memset(S, 0, sizeof S);
            for (int k = 0; k < 3; k++) {
                S[(2 * k) * 6 + 2 * k] = sig[0] * p->h; S[(2 * k) * 6 + 2 * k + 1] = sig[2] * p->h;
                S[(2 * k + 1) * 6 + 2 * k] = sig[2] * p->h; S[(2 * k + 1) * 6 + 2 * k + 1] = sig[1] * p->h;
            }
            btcb(Ks, det, &BG[0][0], 6, S, &BG[0][0], 6, 24);
        }
        if (f) {
            double E[3][3];
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++) {
                    double hh = 0;
                    for (int k = 0; k < 3; k++) hh += H[k][a] * H[k][b];
                    E[a][b] = 0.5 * (H[a][b] + H[b][a] + hh);
                }
            double eps[3] = {E[0][0], E[1][1], 2 * E[0][1]}, sm[3];
            for (int r = 0; r < 3; r++) {
                double s = 0;
                for (int c = 0; c < 3; c++) s += p->C_raw[r * 3 + c] * eps[c];
                sm[r] = s;
            }
            for (int i = 0; i < 24; i++) {
                double s = 0;
                for (int r = 0; r < 3; r++) s += (BL[r][i] + BNL[r][i]) * sm[r];
                fl[i] += s * 1.0 * det * p->h;
            }
        }
    }
    if (KT) {
        double Kt[576], Ksym[576];
        for (int i = 0; i < 24; i++)
            for (int j = 0; j < 24; j++) Ksym[i * 24 + j] = 0.5 * (Ks[i * 24 + j] + Ks[j * 24 + i]);
        for (int i = 0; i < 576; i++) Kt[i] = K0[i] + KL[i] + Ksym[i];
        double Kt2[576];
        for (int i = 0; i < 24; i++)
            for (int j = 0; j < 24; j++) Kt2[i * 24 + j] = 0.5 * (Kt[i * 24 + j] + Kt[j * 24 + i]);
        rotate_to_global(&g, Kt2, KT);
    }
    if (f) rotate_vec(&g, fl, f, 0);
    return g.status;
}

int fso_mitc4_K_batch(const double *xyz, const int32_t *prop_id, const fso_shell_prop *props,
                      int64_t E, double *Ke, int32_t *status, int nthreads) {
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(static)
#endif
    for (int64_t e = 0; e < E; e++) {
        int s = fso_mitc4_K(xyz + 12 * e, props + prop_id[e], Ke + 576 * e);
        if (status) status[e] = s;
    }
    return 0;
}
int fso_mitc4_M_batch(const double *xyz, const int32_t *prop_id, const fso_shell_prop *props,
                      int64_t E, double *Me, int32_t *status, int nthreads) {
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(static)
#endif
    for (int64_t e = 0; e < E; e++) {
        int s = fso_mitc4_M(xyz + 12 * e, props + prop_id[e], Me + 576 * e);
        if (status) status[e] = s;
    }
    return 0;
}

/* ---------- assembly restatement (core/assembler.py:222-241, 269-303) ---------- */
static int cmp_i64(const void *a, const void *b) {
    int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
    return (x > y) - (x < y);
}
int64_t fso_csr_symbolic(const int64_t *dofs, int64_t E, int nd, int64_t ndof, int64_t *indptr,
                         int32_t *indices) {
    /* row -> list of elements (CSR of the transpose incidence), then per-row sort/unique */
    int64_t *cnt = (int64_t *)calloc(ndof + 1, sizeof(int64_t));
    for (int64_t e = 0; e < E; e++)
        for (int a = 0; a < nd; a++) cnt[dofs[e * nd + a] + 1]++;
    for (int64_t r = 0; r < ndof; r++) cnt[r + 1] += cnt[r];
    int64_t *el = (int64_t *)malloc((cnt[ndof] > 0 ? cnt[ndof] : 1) * sizeof(int64_t));
    int64_t *pos = (int64_t *)malloc((ndof + 1) * sizeof(int64_t));
    memcpy(pos, cnt, (ndof + 1) * sizeof(int64_t));
    for (int64_t e = 0; e < E; e++)
        for (int a = 0; a < nd; a++) el[pos[dofs[e * nd + a]]++] = e;
    int64_t cap = 4096, nnz = 0;
    int64_t *buf = (int64_t *)malloc(cap * sizeof(int64_t));
    if (!indices) indptr[0] = 0;
    for (int64_t r = 0; r < ndof; r++) {
        int64_t m = (cnt[r + 1] - cnt[r]) * nd;
        if (m > cap) { cap = 2 * m; buf = (int64_t *)realloc(buf, cap * sizeof(int64_t)); }
        int64_t k = 0;
        for (int64_t q = cnt[r]; q < cnt[r + 1]; q++)
            for (int b = 0; b < nd; b++) buf[k++] = dofs[el[q] * nd + b];
        qsort(buf, k, sizeof(int64_t), cmp_i64);
        int64_t u = 0;
        for (int64_t i = 0; i < k; i++)
            if (i == 0 || buf[i] != buf[i - 1]) {
                if (indices) indices[nnz + u] = (int32_t)buf[i];
                u++;
            }
        nnz += u;
        if (!indices) indptr[r + 1] = nnz;
    }
    free(buf); free(pos); free(el); free(cnt);
    return nnz;
}
int fso_csr_numeric(const int64_t *dofs, int64_t E, int nd, const double *vals_e,
                    const int64_t *indptr, const int32_t *indices, double *data) {
    for (int64_t e = 0; e < E; e++) {
        const int64_t *d = dofs + e * nd;
        const double *ke = vals_e + (int64_t)e * nd * nd;
        for (int a = 0; a < nd; a++) {
            int64_t r = d[a], lo0 = indptr[r], hi0 = indptr[r + 1];
            for (int b = 0; b < nd; b++) {
                int64_t c = d[b], lo = lo0, hi = hi0;
                while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (indices[mid] < c) lo = mid + 1; else hi = mid; }
                if (lo >= hi0 || indices[lo] != c) return -1;
                data[lo] += ke[a * nd + b];
            }
        }
    }
    return 0;
}

/* Row-parallel form of fso_csr_numeric for the CPU baseline (bench.py): every row gathers the rows of the element
 * matrices that touch it, elements in ascending order -- the same sums in the same order as the element loop above
 * (core/assembler.py:295-300), so the result is bit-identical, but rows are independent and go to OpenMP threads. */
int fso_csr_numeric_rows(const int64_t *dofs, int64_t E, int nd, const double *vals_e, int64_t ndof,
                         const int64_t *indptr, const int32_t *indices, double *data, int nthreads) {
    int64_t *cnt = (int64_t *)calloc(ndof + 1, sizeof(int64_t));
    for (int64_t e = 0; e < E; e++)
        for (int a = 0; a < nd; a++) cnt[dofs[e * nd + a] + 1]++;
    for (int64_t r = 0; r < ndof; r++) cnt[r + 1] += cnt[r];
    int64_t *el = (int64_t *)malloc((cnt[ndof] > 0 ? cnt[ndof] : 1) * sizeof(int64_t));   /* (element * nd + a) per row */
    int64_t *pos = (int64_t *)malloc((ndof + 1) * sizeof(int64_t));
    memcpy(pos, cnt, (ndof + 1) * sizeof(int64_t));
    for (int64_t e = 0; e < E; e++)
        for (int a = 0; a < nd; a++) el[pos[dofs[e * nd + a]]++] = e * nd + a;
    int bad = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(static) reduction(| : bad)
#endif
    for (int64_t r = 0; r < ndof; r++) {
        const int64_t lo0 = indptr[r], hi0 = indptr[r + 1];
        for (int64_t q = cnt[r]; q < cnt[r + 1]; q++) {
            const int64_t e = el[q] / nd;
            const int a = (int)(el[q] - e * nd);
            const int64_t *d = dofs + e * nd;
            const double *ke = vals_e + ((int64_t)e * nd + a) * nd;
            for (int b = 0; b < nd; b++) {
                int64_t c = d[b], lo = lo0, hi = hi0;
                while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (indices[mid] < c) lo = mid + 1; else hi = mid; }
                if (lo >= hi0 || indices[lo] != c) { bad = 1; continue; }
                data[lo] += ke[b];
            }
        }
    }
    free(pos); free(el); free(cnt);
    return bad ? -1 : 0;
}

/* ====================================================================================
 * MITC3+ (elements/MITC3.py): CST membrane + drilling, enriched-rotation bending, 6-tying-point
 * assumed shear, cubic bubble condensed 20 -> 18.  Literal dense restatement.
 * ==================================================================================== */
typedef struct {
    double x[3][3];
    double e1[3], e2[3], e3[3];
    double x2, x3, y3;      /* local coords: (0,0), (x2,0), (x3,y3) */
    double area, Ji[2][2];
    double dhx[3], dhy[3];
    int status;
} tri_t;

static void tri_init(tri_t *t, const double *xyz) { /* MITC3.py:375-418, 427-469, 527-558 */
    memcpy(t->x, xyz, 9 * sizeof(double));
    t->status = 0;
    double v12[3], v13[3], n[3];
    for (int k = 0; k < 3; k++) { v12[k] = t->x[1][k] - t->x[0][k]; v13[k] = t->x[2][k] - t->x[0][k]; }
    double L12 = norm3(v12);
    for (int k = 0; k < 3; k++) t->e1[k] = v12[k] / L12;
    cross3(v12, v13, n);
    double nl = norm3(n);
    if (!(nl >= 1e-12)) t->status |= 1; /* ValueError("Degenerate triangle") */
    for (int k = 0; k < 3; k++) t->e3[k] = n[k] / nl;
    cross3(t->e3, t->e1, t->e2);
    t->x2 = L12; t->x3 = dot3(v13, t->e1); t->y3 = dot3(v13, t->e2);
    double x1 = 0.0, y1 = 0.0, x2 = L12, y2 = 0.0, x3 = t->x3, y3 = t->y3;
    t->area = 0.5 * fabs(x1 * (y2 - y3) + x2 * (y3 - y1) + x3 * (y1 - y2));
    double J[2][2] = {{x2 - x1, y2 - y1}, {x3 - x1, y3 - y1}};
    double det = J[0][0] * J[1][1] - J[0][1] * J[1][0];
    inv2(J, det, t->Ji);
    const double dr[3] = {-1.0, 1.0, 0.0}, ds[3] = {-1.0, 0.0, 1.0};
    for (int i = 0; i < 3; i++) {
        t->dhx[i] = t->Ji[0][0] * dr[i] + t->Ji[0][1] * ds[i];
        t->dhy[i] = t->Ji[1][0] * dr[i] + t->Ji[1][1] * ds[i];
    }
}

static void tri_shear_point(const tri_t *t, double r, double s, double *ert, double *est) {
    /* MITC3.py:796-851 (_evaluate_covariant_shear_at_point_ext), 20-vectors */
    const double dr[3] = {-1.0, 1.0, 0.0}, ds[3] = {-1.0, 0.0, 1.0};
    double h[3] = {1.0 - r - s, r, s};
    double f4 = 27.0 * r * s * (1.0 - r - s);
    double f[4] = {h[0] - f4 / 3.0, h[1] - f4 / 3.0, h[2] - f4 / 3.0, f4};
    double gr[2] = {t->x2 - 0.0, 0.0 - 0.0}, gs[2] = {t->x3 - 0.0, t->y3 - 0.0};
    memset(ert, 0, 20 * sizeof(double)); memset(est, 0, 20 * sizeof(double));
    for (int i = 0; i < 3; i++) {
        ert[6 * i + 2] = dr[i]; est[6 * i + 2] = ds[i];
        ert[6 * i + 4] = f[i] * gr[0]; ert[6 * i + 3] = -f[i] * gr[1];
        est[6 * i + 4] = f[i] * gs[0]; est[6 * i + 3] = -f[i] * gs[1];
    }
    ert[19] = f[3] * gr[0]; ert[18] = -f[3] * gr[1];
    est[19] = f[3] * gs[0]; est[18] = -f[3] * gs[1];
}

static void tri_Bgamma_ext(const tri_t *t, double r, double s, double Bg[2][20]) { /* MITC3.py:889-933 */
    const double d = 1.0e-4;
    double eA[2][20], eB[2][20], eC[2][20], eD[2][20], eE[2][20], eF[2][20];
    tri_shear_point(t, 1.0 / 6.0, 2.0 / 3.0, eA[0], eA[1]);
    tri_shear_point(t, 2.0 / 3.0, 1.0 / 6.0, eB[0], eB[1]);
    tri_shear_point(t, 1.0 / 6.0, 1.0 / 6.0, eC[0], eC[1]);
    tri_shear_point(t, 1.0 / 3.0 + d, 1.0 / 3.0 - 2.0 * d, eD[0], eD[1]);
    tri_shear_point(t, 1.0 / 3.0 - 2.0 * d, 1.0 / 3.0 + d, eE[0], eE[1]);
    tri_shear_point(t, 1.0 / 3.0 + d, 1.0 / 3.0 + d, eF[0], eF[1]);
    for (int c = 0; c < 20; c++) {
        double ertc = (2.0 / 3.0) * (eB[0][c] - 0.5 * eB[1][c]) + (1.0 / 3.0) * (eC[0][c] + eC[1][c]);
        double estc = (2.0 / 3.0) * (eA[1][c] - 0.5 * eA[0][c]) + (1.0 / 3.0) * (eC[0][c] + eC[1][c]);
        double chat = (eF[0][c] - eD[0][c]) - (eF[1][c] - eE[1][c]);
        double ert = ertc + (1.0 / 3.0) * chat * (3.0 * s - 1.0);
        double est = estc + (1.0 / 3.0) * chat * (1.0 - 3.0 * r);
        Bg[0][c] = t->Ji[0][0] * ert + t->Ji[0][1] * est;
        Bg[1][c] = t->Ji[1][0] * ert + t->Ji[1][1] * est;
    }
}

static void tri_Bkappa_ext(const tri_t *t, double r, double s, double Bk[3][20]) { /* MITC3.py:853-887 */
    double df4r = 27.0 * s * (1.0 - 2.0 * r - s), df4s = 27.0 * r * (1.0 - r - 2.0 * s);
    const double dr[3] = {-1.0, 1.0, 0.0}, ds[3] = {-1.0, 0.0, 1.0};
    double fr[4] = {dr[0] - df4r / 3.0, dr[1] - df4r / 3.0, dr[2] - df4r / 3.0, df4r};
    double fs[4] = {ds[0] - df4s / 3.0, ds[1] - df4s / 3.0, ds[2] - df4s / 3.0, df4s};
    memset(Bk, 0, 60 * sizeof(double));
    for (int i = 0; i < 4; i++) {
        double fx = t->Ji[0][0] * fr[i] + t->Ji[0][1] * fs[i];
        double fy = t->Ji[1][0] * fr[i] + t->Ji[1][1] * fs[i];
        int thx = (i < 3) ? 6 * i + 3 : 18, thy = (i < 3) ? 6 * i + 4 : 19;
        Bk[0][thy] = fx; Bk[1][thx] = -fy; Bk[2][thy] = fy; Bk[2][thx] = -fx;
    }
}

static void btcb_n(double *acc, int n, double s, const double *Ba, int ra, const double *C, const double *Bb, int rb) {
    /* acc[n][n] += s * Ba^T (ra x n) C (ra x rb) Bb (rb x n) */
    for (int i = 0; i < n; i++) {
        double tt[6];
        int any = 0;
        for (int b = 0; b < rb; b++) {
            double v = 0;
            for (int a = 0; a < ra; a++) v += (s * Ba[a * n + i]) * C[a * rb + b];
            tt[b] = v; any |= (v != 0.0);
        }
        if (!any) continue;
        for (int j = 0; j < n; j++) {
            double v = 0;
            for (int b = 0; b < rb; b++) v += tt[b] * Bb[b * n + j];
            acc[i * n + j] += v;
        }
    }
}

static const double TRI_GP[3][2] = {{1.0 / 6.0, 1.0 / 6.0}, {2.0 / 3.0, 1.0 / 6.0}, {1.0 / 6.0, 2.0 / 3.0}};

static void tri_rotate(const tri_t *t, const double *Kl, double *Kg) { /* T^T K T, 6 blocks of 3 */
    double R[3][3];
    memcpy(R[0], t->e1, 24); memcpy(R[1], t->e2, 24); memcpy(R[2], t->e3, 24);
    for (int bi = 0; bi < 6; bi++)
        for (int bj = 0; bj < 6; bj++) {
            double X[3][3], Y[3][3];
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++) X[a][b] = Kl[(3 * bi + a) * 18 + 3 * bj + b];
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++) {
                    double s = 0;
                    for (int k = 0; k < 3; k++) s += R[k][a] * X[k][b];
                    Y[a][b] = s;
                }
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++) {
                    double s = 0;
                    for (int k = 0; k < 3; k++) s += Y[a][k] * R[k][b];
                    Kg[(3 * bi + a) * 18 + 3 * bj + b] = s;
                }
        }
}

/* local parts of MITC3.K: km = membrane CST + drilling (MITC3.py:1009-1037), kbs = condensed and
 * symmetrised bending+shear (:935-1007), kc = membrane-bending coupling (MITC3_composite.py:204-240) */
static void tri_k_local(const tri_t *tp, const fso_shell_prop *p, double *km, double *kbs_sym, double *kc) {
    const tri_t t = *tp;
    double Kext[400];
    memset(km, 0, 324 * sizeof(double)); memset(Kext, 0, sizeof Kext);
    double Bm[3][18];
    memset(Bm, 0, sizeof Bm); memset(kc, 0, 324 * sizeof(double));
    for (int i = 0; i < 3; i++) {
        Bm[0][6 * i] = t.dhx[i]; Bm[1][6 * i + 1] = t.dhy[i];
        Bm[2][6 * i] = t.dhy[i]; Bm[2][6 * i + 1] = t.dhx[i];
    }
    for (int g = 0; g < 3; g++) { /* MITC3.py:1009-1037 */
        double w = (1.0 / 3.0) * t.area;
        btcb_n(km, 18, w, &Bm[0][0], 3, p->A, &Bm[0][0], 3);
    }
    for (int g = 0; g < 3; g++) {
        double r = TRI_GP[g][0], s = TRI_GP[g][1], w = (1.0 / 3.0) * t.area;
        double h[3] = {1.0 - r - s, r, s}, Bd[18];
        memset(Bd, 0, sizeof Bd);
        for (int i = 0; i < 3; i++) { Bd[6 * i] = -0.5 * t.dhy[i]; Bd[6 * i + 1] = 0.5 * t.dhx[i]; Bd[6 * i + 5] = -h[i]; }
        for (int i = 0; i < 18; i++)
            for (int j = 0; j < 18; j++) km[i * 18 + j] += w * p->k_drill * Bd[i] * Bd[j];
        double Bk[3][20], Bg[2][20];
        tri_Bkappa_ext(&t, r, s, Bk);
        tri_Bgamma_ext(&t, r, s, Bg);
        btcb_n(Kext, 20, w, &Bk[0][0], 3, p->D, &Bk[0][0], 3);   /* MITC3.py:935-956 */
        btcb_n(Kext, 20, w, &Bg[0][0], 2, p->Cs, &Bg[0][0], 2);
        if (p->flags & FSO_PROP_COUPLED) { /* MITC3_composite.py:204-240, nodal-only form */
            double tmp[324], Bk18[3][18];
            memset(tmp, 0, sizeof tmp);
            for (int a = 0; a < 3; a++)
                for (int c = 0; c < 18; c++) Bk18[a][c] = Bk[a][c];
            btcb_n(tmp, 18, w, &Bm[0][0], 3, p->B, &Bk18[0][0], 3);
            for (int i = 0; i < 18; i++)
                for (int j = 0; j < 18; j++) kc[i * 18 + j] += tmp[i * 18 + j] + tmp[j * 18 + i];
        }
    }
    /* condensation of the 2 bubble rotations */
    double q00 = Kext[18 * 20 + 18], q01 = Kext[18 * 20 + 19], q10 = Kext[19 * 20 + 18], q11 = Kext[19 * 20 + 19];
    double det = q00 * q11 - q01 * q10;
    double i00 = q11 / det, i01 = -q01 / det, i10 = -q10 / det, i11 = q00 / det;
    double kbs[324];
    for (int i = 0; i < 18; i++) {
        double a0 = Kext[i * 20 + 18], a1 = Kext[i * 20 + 19];
        for (int j = 0; j < 18; j++) {
            double b0 = Kext[18 * 20 + j], b1 = Kext[19 * 20 + j];
            kbs[i * 18 + j] = Kext[i * 20 + j] - (a0 * (i00 * b0 + i01 * b1) + a1 * (i10 * b0 + i11 * b1));
        }
    }
    for (int i = 0; i < 18; i++)
        for (int j = 0; j < 18; j++) kbs_sym[i * 18 + j] = 0.5 * (kbs[i * 18 + j] + kbs[j * 18 + i]);
}

int fso_mitc3_K(const double *xyz, const fso_shell_prop *p, double *Ke) { /* MITC3.py:1077-1103 */
    tri_t t;
    tri_init(&t, xyz);
    double km[324], kbs[324], kc[324], Kl[324], Kg[324];
    tri_k_local(&t, p, km, kbs, kc);
    for (int i = 0; i < 18; i++)
        for (int j = 0; j < 18; j++) {
            Kl[i * 18 + j] = km[i * 18 + j] + kbs[i * 18 + j];
            if (p->flags & FSO_PROP_COUPLED) Kl[i * 18 + j] += kc[i * 18 + j];
        }
    tri_rotate(&t, Kl, Kg);
    for (int i = 0; i < 18; i++)
        for (int j = 0; j < 18; j++) Ke[i * 18 + j] = 0.5 * (Kg[i * 18 + j] + Kg[j * 18 + i]);
    return t.status;
}

/* rotation with an explicit frame R (rows e1,e2,e3): Kg = T^T Kl T ; vectors: to_local ? T u : T^T u */
static void tri_rotate_R(const double R[3][3], const double *Kl, double *Kg) {
    tri_t t;
    memcpy(t.e1, R[0], 24); memcpy(t.e2, R[1], 24); memcpy(t.e3, R[2], 24);
    tri_rotate(&t, Kl, Kg);
}
static void tri_rotate_vec(const double R[3][3], const double *u, double *out, int to_local) {
    for (int b = 0; b < 6; b++)
        for (int a = 0; a < 3; a++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += (to_local ? R[a][k] : R[k][a]) * u[3 * b + k];
            out[3 * b + a] = s;
        }
}

/* MITC3 total-Lagrangian tangent / internal force after update_configuration(u)
 * (MITC3.py:1144-1187, 1198-1258, 1334-1447, 1449-1580, 1582-1631, 1717-1832; composite overrides
 * MITC3_composite.py:246-265, 496-621).  Quirks kept: every derivative, the area and K0 come from the
 * INITIAL geometry (cached _local_coordinates / _dH_cache), while T() is rebuilt from the CURRENT
 * coordinates x + u (MITC3.py:471-494); H mixes global displacement components with local
 * derivatives (:1198-1225). */
int fso_mitc3_NL(const double *xyz, const fso_shell_prop *p, const double *u, double *KT, double *f) {
    tri_t t;
    tri_init(&t, xyz);
    double km[324], kbs[324], kc[324], K0[324];
    tri_k_local(&t, p, km, kbs, kc);
    const int coupled = (p->flags & FSO_PROP_COUPLED) != 0;
    for (int i = 0; i < 324; i++) K0[i] = km[i] + kbs[i] + (coupled ? kc[i] : 0.0);
    /* T() from the current coordinates */
    double R[3][3];
    {
        double xc[3][3], v12[3], v13[3], n[3];
        for (int i = 0; i < 3; i++)
            for (int k = 0; k < 3; k++) xc[i][k] = t.x[i][k] + u[6 * i + k];
        for (int k = 0; k < 3; k++) { v12[k] = xc[1][k] - xc[0][k]; v13[k] = xc[2][k] - xc[0][k]; }
        double l = norm3(v12);
        for (int k = 0; k < 3; k++) R[0][k] = v12[k] / l;
        cross3(v12, v13, n);
        l = norm3(n);
        for (int k = 0; k < 3; k++) R[2][k] = n[k] / l;
        cross3(R[2], R[0], R[1]);
    }
    double ul[18];
    tri_rotate_vec(R, u, ul, 1);
    if (!p->nonlinear) {
        if (KT) tri_rotate_R(R, K0, KT); /* MITC3.py:1728-1732, not symmetrised */
        if (f) { /* MITC3.py:1786-1832 with the linear membrane branch */
            double fl[18];
            for (int i = 0; i < 18; i++) {
                double s = 0;
                for (int j = 0; j < 18; j++) s += (kbs[i * 18 + j] + (coupled ? kc[i * 18 + j] : 0.0)) * ul[j];
                fl[i] = s;
            }
            for (int g = 0; g < 3; g++) {
                double r = TRI_GP[g][0], s = TRI_GP[g][1], w = 1.0 / 3.0;
                double h[3] = {1.0 - r - s, r, s}, Bd[18], bu = 0;
                memset(Bd, 0, sizeof Bd);
                for (int i = 0; i < 3; i++) { Bd[6 * i] = -0.5 * t.dhy[i]; Bd[6 * i + 1] = 0.5 * t.dhx[i]; Bd[6 * i + 5] = -h[i]; }
                for (int i = 0; i < 18; i++) bu += Bd[i] * ul[i];
                for (int i = 0; i < 18; i++) fl[i] += p->k_drill * (Bd[i] * bu) * w * t.area;
            }
            for (int g = 0; g < 3; g++) {
                double eps[3] = {0, 0, 0}, sm[3], w = 1.0 / 3.0;
                for (int i = 0; i < 3; i++) {
                    eps[0] += t.dhx[i] * ul[6 * i]; eps[1] += t.dhy[i] * ul[6 * i + 1];
                    eps[2] += t.dhy[i] * ul[6 * i] + t.dhx[i] * ul[6 * i + 1];
                }
                for (int a = 0; a < 3; a++) sm[a] = p->C_raw[a * 3] * eps[0] + p->C_raw[a * 3 + 1] * eps[1] + p->C_raw[a * 3 + 2] * eps[2];
                for (int i = 0; i < 3; i++) {
                    fl[6 * i] += (t.dhx[i] * sm[0] + t.dhy[i] * sm[2]) * w * t.area * p->h;
                    fl[6 * i + 1] += (t.dhy[i] * sm[1] + t.dhx[i] * sm[2]) * w * t.area * p->h;
                }
            }
            tri_rotate_vec(R, fl, f, 0);
        }
        return t.status;
    }
    /* sigma at the centroid (MITC3.py:1582-1631; laminate MITC3_composite.py:496-505) */
    double sig[3];
    {
        double eps[3] = {0, 0, 0}, kap[3] = {0, 0, 0};
        for (int i = 0; i < 3; i++) {
            eps[0] += t.dhx[i] * ul[6 * i]; eps[1] += t.dhy[i] * ul[6 * i + 1];
            eps[2] += t.dhy[i] * ul[6 * i] + t.dhx[i] * ul[6 * i + 1];
            kap[0] += t.dhx[i] * ul[6 * i + 4]; kap[1] += -t.dhy[i] * ul[6 * i + 3];
            kap[2] += t.dhy[i] * ul[6 * i + 4] - t.dhx[i] * ul[6 * i + 3];
        }
        for (int a = 0; a < 3; a++) {
            if (p->flags & FSO_PROP_LAMINATE) {
                double s1 = 0, s2 = 0;
                for (int c = 0; c < 3; c++) { s1 += p->A[a * 3 + c] * eps[c]; s2 += p->B[a * 3 + c] * kap[c]; }
                sig[a] = (s1 + s2) / p->h;
            } else {
                double s1 = 0;
                for (int c = 0; c < 3; c++) s1 += p->C_raw[a * 3 + c] * eps[c];
                sig[a] = s1;
            }
        }
    }
    /* H (constant over the CST): global components, local derivatives */
    double H[3][3];
    memset(H, 0, sizeof H);
    for (int c = 0; c < 3; c++)
        for (int i = 0; i < 3; i++) { H[c][0] += u[6 * i + c] * t.dhx[i]; H[c][1] += u[6 * i + c] * t.dhy[i]; }
    double BL[3][18], BNL[3][18], BG[6][18];
    memset(BL, 0, sizeof BL); memset(BNL, 0, sizeof BNL); memset(BG, 0, sizeof BG);
    for (int i = 0; i < 3; i++) {
        int c = 6 * i;
        BL[0][c] = t.dhx[i]; BL[1][c + 1] = t.dhy[i]; BL[2][c] = t.dhy[i]; BL[2][c + 1] = t.dhx[i];
        for (int k = 0; k < 3; k++) {
            BNL[0][c + k] = H[k][0] * t.dhx[i];
            BNL[1][c + k] = H[k][1] * t.dhy[i];
            BNL[2][c + k] = H[k][0] * t.dhy[i] + H[k][1] * t.dhx[i];
            BG[2 * k][c + k] = t.dhx[i]; BG[2 * k + 1][c + k] = t.dhy[i];
        }
    }
    if (KT) {
        double KL[324], Ks[324], S[36], Kt[324], Kt2[324];
        memset(KL, 0, sizeof KL); memset(Ks, 0, sizeof Ks); memset(S, 0, sizeof S);
        for (int k = 0; k < 3; k++) {
            S[(2 * k) * 6 + 2 * k] = sig[0] * p->h; S[(2 * k) * 6 + 2 * k + 1] = sig[2] * p->h;
            S[(2 * k + 1) * 6 + 2 * k] = sig[2] * p->h; S[(2 * k + 1) * 6 + 2 * k + 1] = sig[1] * p->h;
        }
        for (int g = 0; g < 3; g++) {
            double w = (1.0 / 3.0) * t.area;
            btcb_n(KL, 18, w * p->h, &BL[0][0], 3, p->C_raw, &BNL[0][0], 3);
            btcb_n(KL, 18, w * p->h, &BNL[0][0], 3, p->C_raw, &BL[0][0], 3);
            btcb_n(KL, 18, w * p->h, &BNL[0][0], 3, p->C_raw, &BNL[0][0], 3);
            btcb_n(Ks, 18, w, &BG[0][0], 6, S, &BG[0][0], 6);
        }
        for (int i = 0; i < 18; i++)
            for (int j = 0; j < 18; j++)
                Kt[i * 18 + j] = K0[i * 18 + j] + KL[i * 18 + j] + 0.5 * (Ks[i * 18 + j] + Ks[j * 18 + i]);
        for (int i = 0; i < 18; i++)
            for (int j = 0; j < 18; j++) Kt2[i * 18 + j] = 0.5 * (Kt[i * 18 + j] + Kt[j * 18 + i]);
        tri_rotate_R(R, Kt2, KT);
    }
    if (f) {
        double fl[18];
        for (int i = 0; i < 18; i++) {
            double s = 0;
            for (int j = 0; j < 18; j++) s += (kbs[i * 18 + j] + (coupled ? kc[i * 18 + j] : 0.0)) * ul[j];
            fl[i] = s;
        }
        for (int g = 0; g < 3; g++) {
            double r = TRI_GP[g][0], s = TRI_GP[g][1], w = 1.0 / 3.0;
            double h[3] = {1.0 - r - s, r, s}, Bd[18], bu = 0;
            memset(Bd, 0, sizeof Bd);
            for (int i = 0; i < 3; i++) { Bd[6 * i] = -0.5 * t.dhy[i]; Bd[6 * i + 1] = 0.5 * t.dhx[i]; Bd[6 * i + 5] = -h[i]; }
            for (int i = 0; i < 18; i++) bu += Bd[i] * ul[i];
            for (int i = 0; i < 18; i++) fl[i] += p->k_drill * (Bd[i] * bu) * w * t.area;
        }
        double E[3][3];
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) {
                double hh = 0;
                for (int k = 0; k < 3; k++) hh += H[k][a] * H[k][b];
                E[a][b] = 0.5 * (H[a][b] + H[b][a] + hh);
            }
        double eps[3] = {E[0][0], E[1][1], 2 * E[0][1]}, sm[3];
        for (int a = 0; a < 3; a++) sm[a] = p->C_raw[a * 3] * eps[0] + p->C_raw[a * 3 + 1] * eps[1] + p->C_raw[a * 3 + 2] * eps[2];
        for (int g = 0; g < 3; g++) {
            double w = 1.0 / 3.0;
            for (int i = 0; i < 18; i++) {
                double s = 0;
                for (int r = 0; r < 3; r++) s += (BL[r][i] + BNL[r][i]) * sm[r];
                fl[i] += s * w * t.area * p->h;
            }
        }
        tri_rotate_vec(R, fl, f, 0);
    }
    return t.status;
}

/* shell geometric stiffness from a given local membrane stress [sxx, syy, sxy]
 * (MITC4.py:1191-1225, MITC3.py:1501-1580), nn = 3 or 4; KG [6nn][6nn] global axes.  Returns 1 and a
 * zero matrix when max|sigma| < 1e-20 (the assembler skips the element, core/assembler.py:498-500). */
int fso_shell_KG(int nn, const double *xyz, const fso_shell_prop *p, const double *sigma, double *KG) {
    const int nd = 6 * nn;
    memset(KG, 0, (size_t)nd * nd * sizeof(double));
    double mx = fmax(fabs(sigma[0]), fmax(fabs(sigma[1]), fabs(sigma[2])));
    if (mx < 1e-20) return 1;
    double S[36];
    memset(S, 0, sizeof S);
    for (int k = 0; k < 3; k++) {
        S[(2 * k) * 6 + 2 * k] = sigma[0] * p->h; S[(2 * k) * 6 + 2 * k + 1] = sigma[2] * p->h;
        S[(2 * k + 1) * 6 + 2 * k] = sigma[2] * p->h; S[(2 * k + 1) * 6 + 2 * k + 1] = sigma[1] * p->h;
    }
    if (nn == 4) {
        geom_t g;
        geom_init(&g, xyz);
        double Ks[576], Ksym[576];
        memset(Ks, 0, sizeof Ks);
        for (int gp = 0; gp < 4; gp++) {
            double xi, eta, dNdx[4], dNdy[4], BG[6][24];
            gauss_point(gp, &xi, &eta);
            double det = dH(&g, xi, eta, dNdx, dNdy);
            memset(BG, 0, sizeof BG);
            for (int i = 0; i < 4; i++)
                for (int k = 0; k < 3; k++) { BG[2 * k][6 * i + k] = dNdx[i]; BG[2 * k + 1][6 * i + k] = dNdy[i]; }
            btcb(Ks, det, &BG[0][0], 6, S, &BG[0][0], 6, 24);
        }
        for (int i = 0; i < 24; i++)
            for (int j = 0; j < 24; j++) Ksym[i * 24 + j] = 0.5 * (Ks[i * 24 + j] + Ks[j * 24 + i]);
        rotate_to_global(&g, Ksym, KG);
        return 0;
    }
    tri_t t;
    tri_init(&t, xyz);
    double Ks[324], Ksym[324], BG[6][18];
    memset(Ks, 0, sizeof Ks); memset(BG, 0, sizeof BG);
    for (int i = 0; i < 3; i++)
        for (int k = 0; k < 3; k++) { BG[2 * k][6 * i + k] = t.dhx[i]; BG[2 * k + 1][6 * i + k] = t.dhy[i]; }
    for (int g = 0; g < 3; g++) btcb_n(Ks, 18, (1.0 / 3.0) * t.area, &BG[0][0], 6, S, &BG[0][0], 6);
    for (int i = 0; i < 18; i++)
        for (int j = 0; j < 18; j++) Ksym[i * 18 + j] = 0.5 * (Ks[i * 18 + j] + Ks[j * 18 + i]);
    tri_rotate(&t, Ksym, KG);
    return 0;
}

/* centrifugal prestress: MITC3 (MITC3.py:1633-1715) -> local [sxx, syy, sxy];
 * solids (SOLID.py:268-320) -> global Voigt [xx,yy,zz,xy,yz,zx].  `axis` is normalised here. */
static int centrifugal_radial(int nn, const double *xyz, const double *axis_in, const double *center,
                              double *axis, double *rdir, double *r_radial) {
    double la = norm3(axis_in), c[3] = {0, 0, 0}, rv[3], rp[3], rr[3];
    for (int k = 0; k < 3; k++) axis[k] = axis_in[k] / la;
    for (int k = 0; k < 3; k++) { /* np.mean(axis=0): pairwise == sequential for n <= 8 */
        double s = 0;
        for (int i = 0; i < nn; i++) s += xyz[3 * i + k];
        c[k] = s / nn;
    }
    for (int k = 0; k < 3; k++) rv[k] = c[k] - center[k];
    double d = dot3(rv, axis);
    for (int k = 0; k < 3; k++) { rp[k] = d * axis[k]; rr[k] = rv[k] - rp[k]; }
    *r_radial = norm3(rr);
    if (*r_radial < 1e-10) return 0;
    for (int k = 0; k < 3; k++) rdir[k] = rr[k] / *r_radial;
    return 1;
}
int fso_mitc3_centrifugal(const double *xyz, double rho, double omega, const double *axis_in,
                          const double *center, double *sigma) {
    tri_t t;
    tri_init(&t, xyz);
    double axis[3], rd[3], rr;
    sigma[0] = sigma[1] = sigma[2] = 0.0;
    if (!centrifugal_radial(3, xyz, axis_in, center, axis, rd, &rr)) return t.status;
    double L = sqrt(t.area), s = rho * (omega * omega) * rr * L;
    double c = dot3(t.e1, rd), sn = dot3(t.e2, rd);
    sigma[0] = s * (c * c); sigma[1] = s * (sn * sn); sigma[2] = s * c * sn;
    return t.status;
}
int fso_solid_centrifugal(int nn, const double *xyz, double rho, double omega, const double *axis_in,
                          const double *center, double *sigma) {
    double axis[3], rd[3], rr;
    for (int k = 0; k < 6; k++) sigma[k] = 0.0;
    if (!centrifugal_radial(nn, xyz, axis_in, center, axis, rd, &rr)) return 0;
    double d[3];
    for (int k = 0; k < 3; k++) {
        double mn = xyz[k], mx = xyz[k];
        for (int i = 1; i < nn; i++) { mn = fmin(mn, xyz[3 * i + k]); mx = fmax(mx, xyz[3 * i + k]); }
        d[k] = mx - mn;
    }
    double L = norm3(d) / 1.732, s = rho * (omega * omega) * rr * L;
    sigma[0] = s * rd[0] * rd[0]; sigma[1] = s * rd[1] * rd[1]; sigma[2] = s * rd[2] * rd[2];
    sigma[3] = s * rd[0] * rd[1]; sigma[4] = s * rd[1] * rd[2]; sigma[5] = s * rd[0] * rd[2];
    return 0;
}

int fso_mitc3_M(const double *xyz, const fso_shell_prop *p, double *Me) { /* MITC3.py:1105-1138 */
    tri_t t;
    tri_init(&t, xyz);
    double Ml[324];
    memset(Ml, 0, sizeof Ml);
    for (int g = 0; g < 3; g++) {
        double r = TRI_GP[g][0], s = TRI_GP[g][1], w = 1.0 / 3.0;
        double h[3] = {1.0 - r - s, r, s};
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                double vt = h[i] * h[j] * p->m_t * w * t.area, vr = h[i] * h[j] * p->m_r * w * t.area;
                for (int k = 0; k < 3; k++) Ml[(6 * i + k) * 18 + 6 * j + k] += vt;
                for (int k = 3; k < 6; k++) Ml[(6 * i + k) * 18 + 6 * j + k] += vr;
            }
    }
    tri_rotate(&t, Ml, Me);
    return t.status;
}

int fso_mitc3_batch(const double *xyz, const int32_t *prop_id, const fso_shell_prop *props, int64_t E,
                    double *Ke, double *Me, int32_t *status, int nthreads) {
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(static)
#endif
    for (int64_t e = 0; e < E; e++) {
        int s = 0;
        if (Ke) s |= fso_mitc3_K(xyz + 9 * e, props + prop_id[e], Ke + 324 * e);
        if (Me) s |= fso_mitc3_M(xyz + 9 * e, props + prop_id[e], Me + 324 * e);
        if (status) status[e] = s;
    }
    return 0;
}

/* ====================================================================================
 * Solids (elements/SOLID.py): TETRA4, PYRAMID5, WEDGE6, HEXA8, TETRA10, PYRAMID13, WEDGE15, HEXA20.  type == node count.
 * ==================================================================================== */
static int solid_rule(int nn, int reduced, double pts[][3], double *w) {
    const double gp = 1.0 / sqrt(3.0);
    if (nn == 8) { /* SOLID.py:1744-1766: order k-j-i, weights 1; reduced: 1 point, w = 8 */
        if (reduced) { pts[0][0] = pts[0][1] = pts[0][2] = 0.0; w[0] = 8.0; return 1; }
        int n = 0;
        for (int k = 0; k < 2; k++)
            for (int j = 0; j < 2; j++)
                for (int i = 0; i < 2; i++) {
                    pts[n][0] = i ? gp : -gp; pts[n][1] = j ? gp : -gp; pts[n][2] = k ? gp : -gp;
                    w[n++] = 1.0;
                }
        return 8;
    }
    if (nn == 4) { pts[0][0] = pts[0][1] = pts[0][2] = 0.25; w[0] = 1.0 / 6.0; return 1; } /* :1575-1579 */
    if (nn == 6) { /* SOLID.py:731-754: tri_w * lin_w * 2 (the documented x2 quirk) */
        const double tp[3][2] = {{1.0 / 6, 1.0 / 6}, {2.0 / 3, 1.0 / 6}, {1.0 / 6, 2.0 / 3}};
        int n = 0;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 2; j++) {
                pts[n][0] = tp[i][0]; pts[n][1] = tp[i][1]; pts[n][2] = j ? gp : -gp;
                w[n++] = (1.0 / 6) * 1.0 * 2;
            }
        return 6;
    }
    if (nn == 5) { /* SOLID.py:1099-1127 Felippa 8-point */
        const double a = 0.26318405556971359557, za = 0.54415184401122528880, wa = 0.10078588207982543059;
        const double b = 0.50661630334978742377, zb = 0.12251482265544137787, wb = 0.23254745125350790274;
        const double sx[4] = {-1, 1, 1, -1}, sy[4] = {-1, -1, 1, 1};
        for (int q = 0; q < 4; q++) {
            pts[q][0] = sx[q] * a; pts[q][1] = sy[q] * a; pts[q][2] = za; w[q] = wa;
            pts[4 + q][0] = sx[q] * b; pts[4 + q][1] = sy[q] * b; pts[4 + q][2] = zb; w[4 + q] = wb;
        }
        return 8;
    }
    if (nn == 13) { /* SOLID.py:1268-1321: 18-point Felippa rule, two 3 x 3 layers */
        const double u[2] = {0.35309846330877704481, 0.67969709567986745790};
        const double z[2] = {0.54415184401122528880, 0.12251482265544137787};
        const double ww[2][3] = {{0.023330065296255886709, 0.037328104474009418735, 0.059724967158415069975},
                                 {0.053830428530904607120, 0.086128685649447371390, 0.137805897039115794220}};
        int n = 0;
        for (int l = 0; l < 2; l++)
            for (int j = -1; j <= 1; j++)
                for (int i = -1; i <= 1; i++) {
                    pts[n][0] = i * u[l]; pts[n][1] = j * u[l]; pts[n][2] = z[l];
                    w[n++] = ww[l][(i == 0) + (j == 0)];
                }
        return 18;
    }
    if (nn == 10) { /* SOLID.py:1642-1655 */
        const double a = (5 - sqrt(5.0)) / 20, b = (5 + 3 * sqrt(5.0)) / 20;
        for (int q = 0; q < 4; q++) {
            pts[q][0] = pts[q][1] = pts[q][2] = a;
            if (q) pts[q][q - 1] = b;
            w[q] = 1.0 / 24;
        }
        return 4;
    }
    if (nn == 15) { /* SOLID.py:880-919: 7-point triangle x 3-point Gauss, tri_w * lin_w * 2 */
        const double tp[7][2] = {{1.0 / 3, 1.0 / 3}, {0.797426985353087, 0.101286507323456}, {0.101286507323456, 0.797426985353087},
                                 {0.101286507323456, 0.101286507323456}, {0.470142064105115, 0.059715871789770},
                                 {0.470142064105115, 0.470142064105115}, {0.059715871789770, 0.470142064105115}};
        const double tw[7] = {0.225, 0.125939180544827, 0.125939180544827, 0.125939180544827, 0.132394152788506,
                              0.132394152788506, 0.132394152788506};
        const double s35 = sqrt(3.0 / 5.0), lp[3] = {-s35, 0, s35}, lw[3] = {5.0 / 9, 8.0 / 9, 5.0 / 9};
        int n = 0;
        for (int i = 0; i < 7; i++)
            for (int j = 0; j < 3; j++) {
                pts[n][0] = tp[i][0]; pts[n][1] = tp[i][1]; pts[n][2] = lp[j];
                w[n++] = tw[i] * 0.5 * lw[j] * 2;
            }
        return 21;
    }
    if (nn == 20) { /* SOLID.py:1862-1877: 3x3x3, order k-j-i */
        const double s35 = sqrt(3.0 / 5.0), p1[3] = {-s35, 0, s35}, w1[3] = {5.0 / 9, 8.0 / 9, 5.0 / 9};
        int n = 0;
        for (int k = 0; k < 3; k++)
            for (int j = 0; j < 3; j++)
                for (int i = 0; i < 3; i++) {
                    pts[n][0] = p1[i]; pts[n][1] = p1[j]; pts[n][2] = p1[k];
                    w[n++] = w1[i] * w1[j] * w1[k];
                }
        return 27;
    }
    return -1;
}

/* HEXA20 serendipity shape functions (SOLID.py:1879-1932); x**2 of a NumPy scalar is libm pow */
static void hexa20_shape(double xi, double eta, double zeta, double *N) {
    static const double c[8][3] = {{-1, -1, -1}, {1, -1, -1}, {1, 1, -1}, {-1, 1, -1}, {-1, -1, 1}, {1, -1, 1}, {1, 1, 1}, {-1, 1, 1}};
    for (int i = 0; i < 8; i++)
        N[i] = 0.125 * (1 + xi * c[i][0]) * (1 + eta * c[i][1]) * (1 + zeta * c[i][2]) *
               (xi * c[i][0] + eta * c[i][1] + zeta * c[i][2] - 2);
    N[8] = 0.25 * (1 - pow(xi, 2)) * (1 - eta) * (1 - zeta);
    N[9] = 0.25 * (1 - xi) * (1 - pow(eta, 2)) * (1 - zeta);
    N[10] = 0.25 * (1 - xi) * (1 - eta) * (1 - pow(zeta, 2));
    N[11] = 0.25 * (1 + xi) * (1 - pow(eta, 2)) * (1 - zeta);
    N[12] = 0.25 * (1 + xi) * (1 - eta) * (1 - pow(zeta, 2));
    N[13] = 0.25 * (1 - pow(xi, 2)) * (1 + eta) * (1 - zeta);
    N[14] = 0.25 * (1 + xi) * (1 + eta) * (1 - pow(zeta, 2));
    N[15] = 0.25 * (1 - xi) * (1 + eta) * (1 - pow(zeta, 2));
    N[16] = 0.25 * (1 - pow(xi, 2)) * (1 - eta) * (1 + zeta);
    N[17] = 0.25 * (1 - xi) * (1 - pow(eta, 2)) * (1 + zeta);
    N[18] = 0.25 * (1 + xi) * (1 - pow(eta, 2)) * (1 + zeta);
    N[19] = 0.25 * (1 - pow(xi, 2)) * (1 + eta) * (1 + zeta);
}

static void solid_shape(int nn, double xi, double eta, double zeta, double *N, double *dxi, double *deta,
                        double *dzeta) {
    if (nn == 20) { /* SOLID.py:1934-1946: FORWARD DIFFERENCES, h = 1e-8 (reference quirk, kept) */
        const double h = 1e-8;
        double Na[20], Nb[20], Nc[20];
        hexa20_shape(xi, eta, zeta, N);
        hexa20_shape(xi + h, eta, zeta, Na);
        hexa20_shape(xi, eta + h, zeta, Nb);
        hexa20_shape(xi, eta, zeta + h, Nc);
        for (int i = 0; i < 20; i++) { dxi[i] = (Na[i] - N[i]) / h; deta[i] = (Nb[i] - N[i]) / h; dzeta[i] = (Nc[i] - N[i]) / h; }
        return;
    }
    if (nn == 13) { /* PYRAMID13, SOLID.py:1323-1537: serendipity functions of the scaled base x vertical blends */
        double r = 1 - zeta; if (r < 1e-14) r = 1e-14;
        const double r2 = r * r;
        double xs = xi / r, es = eta / r;
        xs = xs < -1 ? -1 : (xs > 1 ? 1 : xs);
        es = es < -1 ? -1 : (es > 1 ? 1 : es);
        const double hb = r * (1 - 2 * zeta), hl = 4 * zeta * r, dhb = -3 + 4 * zeta, dhl = 4 * (1 - 2 * zeta);
        /* function of (xs, es), its two derivatives, per node; blend h and dh/dzeta */
        double fn[13], fx[13], fe[13];
        const double sx[4] = {-1, 1, 1, -1}, sy[4] = {-1, -1, 1, 1};
        for (int i = 0; i < 4; i++) { /* corners 0..3 and lateral mid-edges 7, 9, 11, 12 */
            const double a = 1 + sx[i] * xs, b = 1 + sy[i] * es, c = sx[i] * xs + sy[i] * es - 1;
            fn[i] = 0.25 * a * b * c;
            fx[i] = 0.25 * b * (2 * xs + sx[i] * sy[i] * es);
            fe[i] = 0.25 * a * (sx[i] * sy[i] * xs + 2 * es);
            const int l = i == 0 ? 7 : (i == 1 ? 9 : (i == 2 ? 11 : 12));
            fn[l] = 0.25 * a * b; fx[l] = 0.25 * sx[i] * b; fe[l] = 0.25 * sy[i] * a;
        }
        fn[5] = 0.5 * (1 - xs * xs) * (1 - es); fx[5] = -xs * (1 - es); fe[5] = -0.5 * (1 - xs * xs);
        fn[6] = 0.5 * (1 - xs) * (1 - es * es); fx[6] = -0.5 * (1 - es * es); fe[6] = -(1 - xs) * es;
        fn[8] = 0.5 * (1 + xs) * (1 - es * es); fx[8] = 0.5 * (1 - es * es); fe[8] = -(1 + xs) * es;
        fn[10] = 0.5 * (1 - xs * xs) * (1 + es); fx[10] = -xs * (1 + es); fe[10] = 0.5 * (1 - xs * xs);
        for (int n = 0; n < 13; n++) {
            if (n == 4) { N[4] = zeta * (2 * zeta - 1); dxi[4] = deta[4] = 0; dzeta[4] = 4 * zeta - 1; continue; }
            const int lat = n == 7 || n == 9 || n == 11 || n == 12;
            const double h = lat ? hl : hb, dh = lat ? dhl : dhb;
            N[n] = fn[n] * h;
            dxi[n] = fx[n] * h / r; deta[n] = fe[n] * h / r;
            dzeta[n] = (fx[n] * xi + fe[n] * eta) / r2 * h + fn[n] * dh;
        }
        return;
    }
    if (nn == 10) { /* SOLID.py:1657-1708 (Gmsh edge order 0-1, 1-2, 0-2, 0-3, 2-3, 1-3) */
        const double L[4] = {1 - xi - eta - zeta, xi, eta, zeta};
        const double dL[4][3] = {{-1, -1, -1}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
        const int ea[6] = {0, 1, 2, 0, 2, 1}, eb[6] = {1, 2, 0, 3, 3, 3};
        double *d[3] = {dxi, deta, dzeta};
        for (int i = 0; i < 4; i++) {
            N[i] = L[i] * (2 * L[i] - 1);
            for (int k = 0; k < 3; k++) d[k][i] = (4 * L[i] - 1) * dL[i][k];
        }
        for (int m = 0; m < 6; m++) {
            N[4 + m] = 4 * L[ea[m]] * L[eb[m]];
            for (int k = 0; k < 3; k++) d[k][4 + m] = 4 * (L[eb[m]] * dL[ea[m]][k] + L[ea[m]] * dL[eb[m]][k]);
        }
        return;
    }
    if (nn == 15) { /* SOLID.py:921-1059 */
        const double L[3] = {1 - xi - eta, xi, eta}, dLx[3] = {-1, 1, 0}, dLe[3] = {-1, 0, 1};
        const double Lz[2] = {0.5 * zeta * (zeta - 1), 0.5 * zeta * (zeta + 1)}, L0 = 1 - zeta * zeta;
        const double dLz[2] = {zeta - 0.5, zeta + 0.5}, dL0 = -2 * zeta;
        const int vert[3] = {8, 10, 11};                         /* vertical mid-edges at corners 0, 1, 2 */
        const int edge[2][3] = {{6, 7, 9}, {12, 13, 14}};        /* edges 0-1, 0-2, 1-2 of the bottom / top face */
        const int ea[3] = {0, 0, 1}, eb[3] = {1, 2, 2};
        for (int c = 0; c < 2; c++) {
            for (int i = 0; i < 3; i++) {
                const int n = 3 * c + i;
                N[n] = L[i] * (2 * L[i] - 1) * Lz[c];
                dxi[n] = (4 * L[i] - 1) * dLx[i] * Lz[c]; deta[n] = (4 * L[i] - 1) * dLe[i] * Lz[c];
                dzeta[n] = L[i] * (2 * L[i] - 1) * dLz[c];
            }
            for (int m = 0; m < 3; m++) {
                const int n = edge[c][m], a = ea[m], b = eb[m];
                N[n] = 4 * L[a] * L[b] * Lz[c];
                dxi[n] = 4 * (dLx[a] * L[b] + L[a] * dLx[b]) * Lz[c];
                deta[n] = 4 * (dLe[a] * L[b] + L[a] * dLe[b]) * Lz[c];
                dzeta[n] = 4 * L[a] * L[b] * dLz[c];
            }
        }
        for (int i = 0; i < 3; i++) {
            const int n = vert[i];
            N[n] = L[i] * L0; dxi[n] = dLx[i] * L0; deta[n] = dLe[i] * L0; dzeta[n] = L[i] * dL0;
        }
        return;
    }
    if (nn == 8) { /* SOLID.py:1768-1818 */
        const double sx[8] = {-1, 1, 1, -1, -1, 1, 1, -1}, sy[8] = {-1, -1, 1, 1, -1, -1, 1, 1},
                     sz[8] = {-1, -1, -1, -1, 1, 1, 1, 1};
        for (int i = 0; i < 8; i++) {
            double fx = 1 + sx[i] * xi, fy = 1 + sy[i] * eta, fz = 1 + sz[i] * zeta;
            N[i] = 0.125 * (fx * fy * fz);
            dxi[i] = 0.125 * (sx[i] * fy * fz); deta[i] = 0.125 * (sy[i] * fx * fz); dzeta[i] = 0.125 * (sz[i] * fx * fy);
        }
    } else if (nn == 4) { /* :1581-1597 */
        N[0] = 1 - xi - eta - zeta; N[1] = xi; N[2] = eta; N[3] = zeta;
        const double a[4] = {-1, 1, 0, 0}, b[4] = {-1, 0, 1, 0}, c[4] = {-1, 0, 0, 1};
        for (int i = 0; i < 4; i++) { dxi[i] = a[i]; deta[i] = b[i]; dzeta[i] = c[i]; }
    } else if (nn == 6) { /* :756-829 */
        double L[3] = {1 - xi - eta, xi, eta}, Lm = 0.5 * (1 - zeta), Lp = 0.5 * (1 + zeta);
        const double dLx[3] = {-1, 1, 0}, dLe[3] = {-1, 0, 1};
        for (int i = 0; i < 3; i++) {
            N[i] = L[i] * Lm; N[3 + i] = L[i] * Lp;
            dxi[i] = dLx[i] * Lm; dxi[3 + i] = dLx[i] * Lp;
            deta[i] = dLe[i] * Lm; deta[3 + i] = dLe[i] * Lp;
            dzeta[i] = L[i] * -0.5; dzeta[3 + i] = L[i] * 0.5;
        }
    } else { /* PYRAMID5 :1129-1218 */
        double r = 1 - zeta; if (r < 1e-14) r = 1e-14;
        double r2 = r * r, rmx = r - xi, rpx = r + xi, rme = r - eta, rpe = r + eta;
        N[0] = 0.25 * rmx * rme / r; N[1] = 0.25 * rpx * rme / r; N[2] = 0.25 * rpx * rpe / r; N[3] = 0.25 * rmx * rpe / r; N[4] = zeta;
        dxi[0] = -0.25 * rme / r; dxi[1] = 0.25 * rme / r; dxi[2] = 0.25 * rpe / r; dxi[3] = -0.25 * rpe / r; dxi[4] = 0;
        deta[0] = -0.25 * rmx / r; deta[1] = -0.25 * rpx / r; deta[2] = 0.25 * rpx / r; deta[3] = 0.25 * rmx / r; deta[4] = 0;
        dzeta[0] = -0.25 * (rme + rmx) / r + 0.25 * rmx * rme / r2;
        dzeta[1] = -0.25 * (rme + rpx) / r + 0.25 * rpx * rme / r2;
        dzeta[2] = -0.25 * (rpe + rpx) / r + 0.25 * rpx * rpe / r2;
        dzeta[3] = -0.25 * (rpe + rmx) / r + 0.25 * rmx * rpe / r2;
        dzeta[4] = 1.0;
    }
}

/* J, det, gradients (SOLID.py:164-216, 242-248). returns 0, or 4 (FS_ST_BAD_JAC) if det <= 1e-14 */
static int solid_grad(int nn, const double *xyz, const double *dxi, const double *deta, const double *dzeta,
                      double *detJ, double g[][3], double gT[][3]) {
    double J[3][3] = {{0}};
    for (int c = 0; c < 3; c++)
        for (int i = 0; i < nn; i++) {
            J[0][c] += dxi[i] * xyz[3 * i + c]; J[1][c] += deta[i] * xyz[3 * i + c]; J[2][c] += dzeta[i] * xyz[3 * i + c];
        }
    double det = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) - J[0][1] * (J[1][0] * J[2][2] - J[1][2] * J[2][0]) +
                 J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
    *detJ = det;
    double inv[3][3];
    inv[0][0] = (J[1][1] * J[2][2] - J[1][2] * J[2][1]) / det; inv[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) / det;
    inv[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) / det; inv[1][0] = (J[1][2] * J[2][0] - J[1][0] * J[2][2]) / det;
    inv[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) / det; inv[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) / det;
    inv[2][0] = (J[1][0] * J[2][1] - J[1][1] * J[2][0]) / det; inv[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) / det;
    inv[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) / det;
    for (int i = 0; i < nn; i++)
        for (int a = 0; a < 3; a++) {
            g[i][a] = inv[a][0] * dxi[i] + inv[a][1] * deta[i] + inv[a][2] * dzeta[i];
            /* quirk: compute_geometric_stiffness uses inv_J.T @ dN_local (SOLID.py:352-354) */
            gT[i][a] = inv[0][a] * dxi[i] + inv[1][a] * deta[i] + inv[2][a] * dzeta[i];
        }
    return (det <= 1e-14) ? 4 : 0;
}

/* K, M, body load, geometric stiffness of one solid element; any output may be NULL.
 * C[36] constitutive (global axes), rho, b[3] body force, sigma[6] Voigt stress (xx,yy,zz,xy,yz,zx).
 * SOLID.py:543-567 (K), :569-597 (M), :599-631 (body_load), :322-374 (K_G). */
int fso_solid(int nn, int reduced, const double *xyz, const double *C, double rho, const double *b,
              const double *sigma, double *Ke, double *Me, double *fe, double *KGe) {
    double pts[27][3], w[27];
    int ng = solid_rule(nn, reduced, pts, w), nd = 3 * nn, status = 0;
    if (ng < 0) return -1;
    if (Ke) memset(Ke, 0, sizeof(double) * nd * nd);
    if (Me) memset(Me, 0, sizeof(double) * nd * nd);
    if (fe) memset(fe, 0, sizeof(double) * nd);
    if (KGe) memset(KGe, 0, sizeof(double) * nd * nd);
    for (int q = 0; q < ng; q++) {
        double N[20], dxi[20], deta[20], dzeta[20], g[20][3], gT[20][3], det;
        solid_shape(nn, pts[q][0], pts[q][1], pts[q][2], N, dxi, deta, dzeta);
        status |= solid_grad(nn, xyz, dxi, deta, dzeta, &det, g, gT);
        if (Ke) {
            double B[6][60];
            memset(B, 0, sizeof B);
            for (int i = 0; i < nn; i++) {
                B[0][3 * i] = g[i][0]; B[1][3 * i + 1] = g[i][1]; B[2][3 * i + 2] = g[i][2];
                B[3][3 * i] = g[i][1]; B[3][3 * i + 1] = g[i][0];
                B[4][3 * i + 1] = g[i][2]; B[4][3 * i + 2] = g[i][1];
                B[5][3 * i] = g[i][2]; B[5][3 * i + 2] = g[i][0];
            }
            for (int r = 0; r < nd; r++) {
                double t6[6];
                for (int c = 0; c < 6; c++) {
                    double s = 0;
                    for (int a = 0; a < 6; a++) s += B[a][r] * C[a * 6 + c];
                    t6[c] = s;
                }
                for (int c = 0; c < nd; c++) {
                    double s = 0;
                    for (int a = 0; a < 6; a++) s += t6[a] * B[a][c];
                    Ke[r * nd + c] += s * det * w[q];
                }
            }
        }
        for (int i = 0; i < nn; i++) {
            if (fe) for (int k = 0; k < 3; k++) fe[3 * i + k] += (N[i] * b[k]) * det * w[q];
            for (int j = 0; j < nn; j++) {
                if (Me) for (int k = 0; k < 3; k++) Me[(3 * i + k) * nd + 3 * j + k] += rho * (N[i] * N[j]) * det * w[q];
                if (KGe) {
                    const double S[3][3] = {{sigma[0], sigma[3], sigma[5]}, {sigma[3], sigma[1], sigma[4]}, {sigma[5], sigma[4], sigma[2]}};
                    double gv = 0;
                    for (int a = 0; a < 3; a++) {
                        double t = 0;
                        for (int c = 0; c < 3; c++) t += gT[i][c] * S[c][a];
                        gv += t * gT[j][a];
                    }
                    for (int k = 0; k < 3; k++) KGe[(3 * i + k) * nd + 3 * j + k] += gv * (w[q] * det);
                }
            }
        }
    }
    if (Ke)
        for (int r = 0; r < nd; r++)
            for (int c = r + 1; c < nd; c++) { double m = 0.5 * (Ke[r * nd + c] + Ke[c * nd + r]); Ke[r * nd + c] = Ke[c * nd + r] = m; }
    if (Me)
        for (int r = 0; r < nd; r++)
            for (int c = r + 1; c < nd; c++) { double m = 0.5 * (Me[r * nd + c] + Me[c * nd + r]); Me[r * nd + c] = Me[c * nd + r] = m; }
    return status;
}

int fso_solid_batch(int nn, int reduced, const double *xyz, const int32_t *prop_id, const double *props /* [P][37]: C, rho */,
                    const double *b, const double *sigma /* [E][6] or NULL */, int64_t E, double *Ke, double *Me,
                    double *fe, double *KGe, int32_t *status, int nthreads) {
    const int nd = 3 * nn;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(static)
#endif
    for (int64_t e = 0; e < E; e++) {
        const double *p = props + 37 * prop_id[e];
        int s = fso_solid(nn, reduced, xyz + 3 * nn * e, p, p[36], b, sigma ? sigma + 6 * e : NULL,
                          Ke ? Ke + (int64_t)nd * nd * e : NULL, Me ? Me + (int64_t)nd * nd * e : NULL,
                          fe ? fe + (int64_t)nd * e : NULL, (KGe && sigma) ? KGe + (int64_t)nd * nd * e : NULL);
        if (status) status[e] = s;
    }
    return 0;
}
