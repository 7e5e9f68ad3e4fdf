/*
 * femshell_hip.h -- C ABI of libfemshell_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the element-matrix + CSR-assembly hot path of efirvida/fem-shell.
 * The reference has no FFI for this path (it is pure Python); each entry point below cites the
 * reference routine (file:line relative to /root/reference/src/fem_shell/) whose per-element /
 * per-mesh work it replaces.  INTEGRATION.md shows the ctypes stub a maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (hipMalloc / torch tensor) unless a
 *     parameter is documented as host memory;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default);
 *   - return value: 0 on success, negative errno-style code on failure (never throws);
 *     fs_last_error() returns a thread-local message for the last failure;
 *   - per-element geometric failures (degenerate frame, non-positive Jacobian) are reported
 *     through an int32 status[E] array (FS_ST_* bits), not through the return value.
 */
#ifndef FEMSHELL_HIP_H
#define FEMSHELL_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
#define FS_STATIC_ASSERT(cond, msg) static_assert(cond, msg)
extern "C" {
#else
#define FS_STATIC_ASSERT(cond, msg) _Static_assert(cond, msg)
#endif

#define FS_OK 0
#define FS_EINVAL (-22)
#define FS_ENOSYS (-38)
#define FS_EHIP (-5)

/* `want` bits */
#define FS_WANT_K 1u
#define FS_WANT_M 2u
#define FS_WANT_KT 4u
#define FS_WANT_FINT 8u
#define FS_WANT_FBODY 16u /* solids: body-load vectors */
#define FS_WANT_KG 32u    /* solids: geometric stiffness from a per-element stress */

/* element-matrix layouts */
#define FS_LAYOUT_DENSE 0u  /* [E][nd][nd] C-order == reference `_ke_array` (core/assembler.py:213-215) */
#define FS_LAYOUT_SYMBLK 1u /* [E][npair][d*d]: node-pair blocks (i<=j) only, row-major d x d each;
                               block (j,i) is the transpose.  pair index p(i,j)=i*nn-i*(i-1)/2+(j-i) */
#define FS_LAYOUT_SYMBLK_T16 2u /* quads (nn = 4, d = 6) only: the same 10 blocks per element, grouped in
                               tiles of 16 elements (160 blocks, buffer = ceil(E/16)*160 blocks) and
                               stored in the order the stiffness kernel produces them, so that it can
                               write whole 128-byte lines.  Block (e, p) has index
                                 (e/16)*160 + (r < 2 ? r*64 + (Q/8)*32 + g*8 + Q%8 : 128 + g*16 + Q),
                               Q = e%16, (r, g) = (0,0) (1,0) (2,0) (1,3) (0,1) (1,1) (2,1) (0,2) (1,2) (0,3)
                               for p = 0..9.  Accepted by fs_shell_quad_batch, fs_shell_kg_batch (nn = 4),
                               fs_blk_expand and fs_csr_symbolic_begin; consumers that take block
                               indices (fs_csr_numeric_*) are layout-agnostic. */

/* status[E] bits */
#define FS_ST_NAN_FRAME 1   /* zero-area / zero-norm local frame (MITC4.py:436-476 silently NaNs) */
#define FS_ST_BAD_COEFF 2   /* characteristic-vector solve failed (MITC4.py:186-193) */
#define FS_ST_BAD_JAC 4     /* det J <= 0 at a Gauss point */
#define FS_ST_BAD_PROP 8    /* prop_id outside [0, n_props): NO entry point ever indexes the table with it -- row 0 is
                               used instead and the entry points that take a status array set this bit (those
                               without one just compute with row 0); also: a coupled property row in a call that
                               declared the table free of them (fs_fused_assemble_quad) */

/* property-table flags */
#define FS_PROP_COUPLED 1   /* add membrane-bending coupling (MITC4_composite.py:98-99,206-248) */
#define FS_PROP_LAMINATE 2  /* composite subclass (NL membrane stress uses (A e + B k)/h) */

/* One row per distinct shell property (isotropic ShellProperty or CompositeShellProperty),
 * built on the host by fem_shell_amd.properties.build_prop_table with the reference's own
 * NumPy expressions (MITC4.py:671-734, MITC4_composite.py:172-204, core/laminate.py:336-452). */
typedef struct fs_shell_prop {
    double A[9];      /* Cm : membrane 3x3            */
    double B[9];      /* laminate coupling 3x3 (0 iso) */
    double D[9];      /* Cb : bending 3x3             */
    double Cs[4];     /* transverse shear 2x2         */
    double h;         /* total thickness              */
    double m_t;       /* mass / area                  */
    double m_r;       /* rotational inertia / area    */
    double k_drill;   /* drilling stabilisation factor */
    double C_raw[9];  /* membrane stress-strain (Cm/h) */
    double rho;       /* density used by the centrifugal prestress (laminates: first ply) */
    int32_t flags;    /* FS_PROP_*                    */
    int32_t nonlinear;
} fs_shell_prop;      /* 368 bytes */
FS_STATIC_ASSERT(sizeof(fs_shell_prop) == 368, "fs_shell_prop is 368 bytes");
FS_STATIC_ASSERT(offsetof(fs_shell_prop, Cs) == 216 && offsetof(fs_shell_prop, h) == 248 &&
                 offsetof(fs_shell_prop, C_raw) == 280 && offsetof(fs_shell_prop, rho) == 352 &&
                 offsetof(fs_shell_prop, flags) == 360, "fs_shell_prop field offsets");

const char *fs_last_error(void);
/* library / device probe: returns FS_OK when a gfx950 device is usable; fills name (<=len).  Also the library's set-up
 * call for fs_shell_quad_fint_noalloc (see there) when `device` is the current device. */
int fs_device_info(int device, char *name, int len, int *cu_count);

/*
 * MITC4+ batch.  Replaces, per element, MITC4.__init__/_compute_local_coordinates
 * (elements/MITC4.py:52-127,436-476), K (:1018-1036 with k_m :934-944, _k_bending_shear
 * :946-1000, k_drill :1002-1016, MITC4Composite._coupling_stiffness MITC4_composite.py:210-248),
 * M (:1038-1061) and, with u_e, compute_tangent_stiffness / compute_internal_forces
 * (:1246-1341, MITC4_composite.py:521-621) -- i.e. the body of the element loop of
 * MeshAssembler._precompute_elements (core/assembler.py:121-193).
 *   xyz     [E][4][3]  nodal coordinates (global axes)
 *   prop_id [E]        row of `props`
 *   props   [n_props]  property table (device)
 *   u_e     [E][24]    element displacement vectors (global axes) or NULL (KT/FINT only)
 *   Ke/Me/KTe          outputs in `layout` (NULL if not wanted); fe [E][24]
 *   status  [E]        FS_ST_* bits (may be NULL)
 * KT / FINT need the linear blocks K of the same elements.  They do not depend on u_e (the reference reads its cached
 * MITC4.K, a cached_property), so a caller that kept them passes them back: FS_LAYOUT_SYMBLK_T16, FS_WANT_K clear and
 * Ke != NULL make Ke an INPUT and skip the stiffness kernel (KTe may then be NULL for FINT alone).  Otherwise the
 * stiffness kernel runs first, into KTe.
 */
int fs_shell_quad_batch(const double *xyz, const int32_t *prop_id, const fs_shell_prop *props,
                        int32_t n_props, const double *u_e, int64_t E, uint32_t want,
                        uint32_t layout, double *Ke, double *Me, double *KTe, double *fe,
                        int32_t *status, void *stream);

/*
 * MITC3+ batch (3-node shells).  Replaces, per element, MITC3.__init__/_compute_local_coordinates
 * (elements/MITC3.py:103-210,375-418), K (:1077-1103 with k_m :1009-1037, _k_bs_condensed :935-1007,
 * tying :796-933, MITC3Composite._coupling_stiffness MITC3_composite.py:204-240) and M (:1105-1138).
 * Same arguments as fs_shell_quad_batch with xyz [E][3][3]; FS_LAYOUT_SYMBLK has 6 pair blocks per
 * element.  With u_e [E][18]: FS_WANT_KT / FS_WANT_FINT = update_configuration(u_e) +
 * compute_tangent_stiffness() / compute_internal_forces() (MITC3.py:1144-1832, MITC3_composite.py:496-621;
 * the reference rebuilds T() from the CURRENT coordinates x + u, :471-494 -- kept); KTe is also the
 * FINT workspace, fe [E][18].  As for the quads, Ke != NULL with FS_WANT_K clear passes the cached linear blocks in
 * (input) and skips the stiffness kernel.
 * status: FS_ST_NAN_FRAME where the reference raises ValueError("Degenerate triangle").
 */
int fs_shell_tri_batch(const double *xyz, const int32_t *prop_id, const fs_shell_prop *props,
                       int32_t n_props, const double *u_e, int64_t E, uint32_t want,
                       uint32_t layout, double *Ke, double *Me, double *KTe, double *fe,
                       int32_t *status, void *stream);

/*
 * Linear solids, 3 dofs/node: nn = 4 TETRA4, 5 PYRAMID5, 6 WEDGE6, 8 HEXA8 (elements/SOLID.py).
 * Replaces, per element, SolidElement._compute_jacobian / compute_B_matrix (:164-266), K (:543-567),
 * M (:569-597), body_load (:599-631), compute_geometric_stiffness (:322-374) with the quadrature
 * rules and shape functions of the concrete classes (:731-829, :1099-1218, :1575-1597, :1744-1818;
 * `reduced` != 0 selects HEXA8's 1-point rule).  Reference quirks kept: WEDGE6 weights sum to 2;
 * K_G uses inv_J.T where B uses inv_J.
 * nn = 10 / 13 / 15 / 20 (TETRA10 / PYRAMID13 / WEDGE15 / HEXA20, SOLID.py:832-1059, 1221-1537, 1600-1708, 1821-1946) run the same kernel
 * from a host-built shape table (fs_solid2_table); the first call per device uploads it (one hipMalloc).
 *   xyz     [E][nn][3]
 *   props   [n_props][37]  C (6x6, global axes, Voigt xx,yy,zz,xy,yz,zx; SOLID.py:377-541 evaluated
 *                          on the host by fem_shell_amd.properties.solid_row) followed by rho
 *   body_host  3 doubles in HOST memory (FS_WANT_FBODY);  sigma [E][6] device (FS_WANT_KG)
 *   Ke/Me/KGe  FS_LAYOUT_SYMBLK with d = 3: [E][nn(nn+1)/2][9];  fe [E][3 nn]
 *   status  FS_ST_BAD_JAC where the reference raises ValueError (det J <= 1e-14); must be zeroed
 *           by the caller (bits are OR-ed in).
 */
int fs_solid_batch(int32_t nn, int32_t reduced, const double *xyz, const int32_t *prop_id,
                   const double *props, int32_t n_props, int64_t E, uint32_t want, uint32_t layout,
                   const double *body_host, const double *sigma, double *Ke, double *Me, double *fe,
                   double *KGe, int32_t *status, void *stream);

/*
 * Shape data of the quadratic solids at their integration points as the reference evaluates them
 * (SOLID.py:880-1059 WEDGE15, :1642-1708 TETRA10, :1862-1946 HEXA20 -- HEXA20 derivatives are the
 * reference's forward differences, h = 1e-8).  Host arrays: pts [ng][3], w [ng], vals [ng][nn][4] =
 * N, dN/dxi, dN/deta, dN/dzeta; any may be NULL.  ng <= 27.  The kernels of fs_solid_batch use the same table.
 */
int fs_solid2_table(int32_t nn, double *pts, double *w, double *vals, int32_t *ng);

/*
 * Shell geometric stiffness from a given membrane stress: per element
 * compute_geometric_stiffness(sigma, transform_to_global=True) (MITC4.py:1191-1225, MITC3.py:1501-1580),
 * the element call of MeshAssembler.assemble_geometric_stiffness (core/assembler.py:483-513).
 *   nn      3 (MITC3) or 4 (MITC4);  xyz [E][nn][3]
 *   sigma   [E][3] device: [sxx, syy, sxy] in the element's LOCAL axes; elements with
 *           max|sigma| < 1e-20 yield zero blocks (skipped by the reference, :498-500)
 *   KGe     FS_LAYOUT_SYMBLK [E][nn(nn+1)/2][36]
 */
int fs_shell_kg_batch(int32_t nn, const double *xyz, const int32_t *prop_id, const fs_shell_prop *props,
                      int32_t n_props, const double *sigma, int64_t E, uint32_t layout, double *KGe,
                      void *stream);

/*
 * Tangent stiffness with a GIVEN membrane stress: per element compute_tangent_stiffness(sigma=sigma)
 * (MITC4.py:1246-1298, MITC3.py:1717-1777, composite overrides): K_T = T^T sym(K0 + K_L(u) + K_sigma(sigma)) T, where
 * the default (sigma=None, FS_WANT_KT of fs_shell_quad_batch / _tri_batch) derives sigma from u.  Elements whose
 * property row is not `nonlinear` return K (the reference ignores sigma there too).
 *   nn 3 / 4;  xyz [E][nn][3];  u_e [E][6 nn];  sigma [E][3] = [sxx, syy, sxy] in the element's LOCAL axes
 *   KTe     FS_LAYOUT_SYMBLK [E][nn(nn+1)/2][36];  status [E] FS_ST_* bits or NULL
 */
int fs_shell_kt_sigma_batch(int32_t nn, const double *xyz, const int32_t *prop_id, const fs_shell_prop *props,
                            int32_t n_props, const double *u_e, const double *sigma, int64_t E, double *KTe,
                            int32_t *status, void *stream);

/*
 * Local-axes views of global-axes element results: the transform_to_global=False outputs of
 * compute_tangent_stiffness / compute_internal_forces / compute_geometric_stiffness (MITC4.py:1191-1341,
 * MITC3.py:1501-1832): K_local = T K_global T^T, f_local = T f_global, T = blockdiag(e1; e2; e3) (MITC4.py:634-640,
 * MITC3.py:471-492).  The frame is computed from `xyz` by the kernels' own geometry stage: pass the coordinates the
 * reference's T() uses -- initial ones for quads, CURRENT ones for triangles.
 *   K_global / K_local  FS_LAYOUT_DENSE [E][6 nn][6 nn] (either pair may be NULL);  f_global / f_local [E][6 nn]
 */
int fs_shell_rotate_local(int32_t nn, const double *xyz, int64_t E, const double *K_global, const double *f_global,
                          double *K_local, double *f_local, void *stream);

/*
 * Centrifugal prestress of every element (the `omega` mode of assemble_geometric_stiffness,
 * core/assembler.py:487-493): compute_centrifugal_prestress of MITC3 (MITC3.py:1633-1715 ->
 * sigma [E][3] local [sxx,syy,sxy]) or of the linear solids (SOLID.py:268-320 -> sigma [E][6]
 * global Voigt [xx,yy,zz,xy,yz,zx]).
 *   solid   0: shells (nn = 3; nn = 4 returns FS_ENOSYS -- the reference MITC4 has no such method),
 *           1: solids (nn = 4, 5, 6, 8, 10, 13, 15, 20)
 *   props   fs_shell_prop rows (rho field) for shells, [n_props][37] rows for solids (rho = column 36)
 *   axis_host, center_host   3 doubles each in HOST memory; the axis is normalised here
 */
int fs_centrifugal_prestress(int32_t solid, int32_t nn, const double *xyz, const int32_t *prop_id,
                             const void *props, int32_t n_props, double omega, const double *axis_host,
                             const double *center_host, int64_t E, double *sigma, void *stream);

/*
 * Batched stress / strain recovery (postprocess/stress_recovery.py:280-1402).
 *
 * fs_shell_stress_batch: StressRecovery._compute_shell_stress / _compute_shell_strain (:529-650) of
 * every MITC4 (nn = 4) / MITC3 (nn = 3) element at `npts` parametric points.
 *   u_e       [E][6 nn] element displacement vectors (slices of the global vector, global axes -- the
 *             reference applies B_m / B_kappa to them without a T-transformation; kept)
 *   pts_host  [npts][2] (r, s) in HOST memory, 1 <= npts <= 4 (element centre, or the nn parametric node
 *             positions of compute_nodal_stresses, :355-358)
 *   location  +1 TOP (z = +h/2), 0 MIDDLE, -1 BOTTOM (z = -h/2)
 *   parts     FS_STRESS_MEMBRANE | FS_STRESS_BENDING (StressType MEMBRANE / BENDING / TOTAL); ignored for
 *             strains (always eps_m + z kappa)
 *   strain    0: stress (Cm / h)(...), 1: strain
 *   out       [E][npts][3]  [xx, yy, xy];   weight [E][npts] or NULL: element area (:1372-1388) repeated
 *             per point (the weight of the area-weighted smoothing)
 */
#define FS_STRESS_MEMBRANE 1u
#define FS_STRESS_BENDING 2u
int fs_shell_stress_batch(int32_t nn, const double *xyz, const int32_t *prop_id, const fs_shell_prop *props,
                          int32_t n_props, const double *u_e, int64_t E, const double *pts_host, int32_t npts,
                          int32_t location, uint32_t parts, int32_t strain, double *out, double *weight,
                          void *stream);

/*
 * fs_solid_stress_batch: _compute_solid_gauss_stresses (:655-705) of TETRA4 / PYRAMID5 / WEDGE6 / HEXA8.
 *   u_e [E][3 nn];  props as for fs_solid_batch (C is used as stored);  mode:
 *   FS_SOLID_STRESS_GP     sigma / eps [E][ngp][6] at the Gauss points (Voigt xx,yy,zz,xy,yz,zx)
 *   FS_SOLID_STRESS_MEAN   [E][6] mean over the Gauss points (compute_element_stresses, :761-764)
 *   FS_SOLID_STRESS_NODES  [E][nn][6] = extrap @ Gauss values, extrap [nn][ngp] on the device
 *                          (E = inv / pinv of N_gp, :437-492, evaluated on the host like the reference)
 *   sigma, eps  either may be NULL;  weight [E] or NULL: element volume sum |det J| w (:1391-1402)
 */
#define FS_SOLID_STRESS_GP 0
#define FS_SOLID_STRESS_MEAN 1
#define FS_SOLID_STRESS_NODES 2
int fs_solid_stress_batch(int32_t nn, int32_t reduced, const double *xyz, const int32_t *prop_id,
                          const double *props, int32_t n_props, const double *u_e, int64_t E,
                          const double *extrap, int32_t mode, double *sigma, double *eps, double *weight,
                          void *stream);

/*
 * fs_nodal_average: the smoothing loop of compute_nodal_stresses / compute_nodal_strains (:844-864):
 * out[n][c] = sum_q w_q vals[src[q]][c] / sum_q w_q over q in [seg_ptr[n], seg_ptr[n+1]) (element
 * order), 0 where a node has no source; c < ncomp, the remaining columns of out [n_nodes][6] are 0.
 *   vals [n_src][ncomp];  w_src [n_src] or NULL (weights 1)
 */
int fs_nodal_average(const double *vals, int32_t ncomp, const double *w_src, const int64_t *seg_ptr,
                     const int64_t *src, int64_t n_nodes, double *out, void *stream);

/*
 * fs_stress_invariants: _build_stress_result / _build_strain_result (:1087-1341) of N Voigt rows
 * [xx, yy, zz, xy, yz, zx].
 *   strain = 0: out [6][N] = von Mises, s1, s2, s3, tau_max, principal angle (rad); is_3d selects the
 *               Cardano form (s3 valid, angle 0) or the Mohr circle on (xx, yy, xy) (s3 = 0)
 *   strain = 1: out [3][N] = e1, e2, gamma_max
 */
int fs_stress_invariants(const double *voigt, int64_t N, int32_t is_3d, int32_t strain, double *out, void *stream);

/* Geometry stage only, for bit-exactness tests against the oracle:
 * frame[E][9] (e1,e2,e3), xl[E][8], detJ[E][4], acoef[E][5] (a_A..a_E). */
int fs_shell_quad_geom(const double *xyz, int64_t E, double *frame, double *xl, double *detJ,
                       double *acoef, void *stream);

/* Status words alone: the FS_ST_* bits fs_shell_quad_batch(FS_WANT_K) would report (degenerate frame, failed
 * coefficient solve, det J <= 0 at a Gauss point, prop_id outside [0, n_props) when prop_id != NULL) without evaluating
 * element matrices -- the geometry checks of the reference's constructor loop (core/assembler.py:160-164 ->
 * MITC4.__init__, elements/MITC4.py:52-99,436-476) for callers that assemble through fs_fused_assemble_quad only. */
int fs_shell_quad_geom_status(const double *xyz, const int32_t *prop_id, int32_t n_props, int64_t E, int32_t *status,
                              void *stream);

/* FS_LAYOUT_SYMBLK / FS_LAYOUT_SYMBLK_T16 (`layout`) -> FS_LAYOUT_DENSE ([E][nn*d][nn*d]);
 * nn nodes/element, d dofs/node. */
int fs_blk_expand(const double *packed, uint32_t layout, int64_t E, int32_t nn, int32_t d, double *dense,
                  void *stream);

/*
 * Numeric CSR assembly, generic form (any element mix).  Replaces the setValuesLocal(ADD_VALUES)
 * loop + assemble() of MeshAssembler.assemble_stiffness_matrix / assemble_mass_matrix
 * (core/assembler.py:269-336): data[s] = sum over q in [seg_ptr[s], seg_ptr[s+1]) of
 * vals[src[q]], contributions ordered by element index (== the reference's insertion order).
 */
int fs_csr_numeric(const double *vals, const int64_t *src, const int64_t *seg_ptr, int64_t nnz,
                   double *data, void *stream);

/*
 * Numeric CSR assembly, node-pair block form for meshes whose nodes all carry d dofs
 * (d = 6 shells, 3 solids) and whose element matrices are in FS_LAYOUT_SYMBLK / _T16.  Same replacement as
 * fs_csr_numeric (core/assembler.py:287-336).  One descriptor per node-pair block (I,J), blocks in any order:
 *   dst     index in `data` of entry (k=0,l=0) of the block: data is CSR with rows = d*I+k, row-major, sorted
 *           columns, so dst = d*d*nbr_ptr[I] + d*r with r = rank of J among I's neighbours
 *   stride  distance between consecutive rows k of the block in `data` (= d * m_I, m_I = neighbours of I)
 *   cnt     number of contributions n; n <= 4: c[0..n) hold them inline
 *           ((index of the d*d source block in `vals` << 1) | transposed, element order); n > 4: c[0] is the
 *           offset of the n contributions in `contrib` (same encoding, sorted by element index) -- unless
 *           FS_BLK_PACKED is set in cnt (cnt = n | FS_BLK_PACKED, 5 <= n <= 8, every code below 2^32): then the 32
 *           bytes of c hold the n codes inline as uint32[8] (the diagonal block of a node shared by 6 triangles or 8
 *           bricks then costs no second, dependent read of `contrib`)
 * The sum runs in element order (== the reference's ADD_VALUES insertion order): n <= 4 as ((x0 + x1) + x2) + x3 with
 * +0.0 for the missing ones, n > 4 as 0.0 + x0 + x1 + ... */
#define FS_BLK_PACKED 0x40000000
typedef struct fs_blk_desc {
    int64_t dst;
    int32_t stride;
    int32_t cnt;
    int64_t c[4];
} fs_blk_desc; /* 48 bytes */
FS_STATIC_ASSERT(sizeof(fs_blk_desc) == 48 && offsetof(fs_blk_desc, stride) == 8 && offsetof(fs_blk_desc, c) == 16,
                 "fs_blk_desc layout (the d = 6 assembly kernel reads it as 3 x 16 bytes)");
int fs_csr_numeric_desc(const double *vals, const fs_blk_desc *desc, const int64_t *contrib,
                        int64_t nblk, int32_t d, double *data, void *stream);

/*
 * Fused evaluate-and-assemble for uniform MITC4+ quad meshes, gather form: the CSR values of K straight from the
 * coordinates, without element matrices in memory.  Replaces the whole loop of MeshAssembler.assemble_stiffness_matrix
 * over the cached element matrices (core/assembler.py:287-303) TOGETHER with the element evaluations that fill the
 * cache (core/assembler.py:160-193 -> MITC4.K, elements/MITC4.py:1018-1036).  Same block function and the same
 * element-order sums as fs_shell_quad_batch(FS_WANT_K) followed by fs_csr_numeric_desc; the values agree with that route to
 * rounding (about 4e-16 of the largest entry: a contribution its block receives transposed is formed as the block of the
 * swapped node pair, the Gauss-point sums of stage 1 run in another order), not to the bit.  The result is EXACTLY
 * symmetric (block (J, I) is written from the registers of block (I, J)), the same bits in every run, and an entry that
 * is exactly zero may carry either sign.  FS_FUSED_MASS is bit-identical to the two-kernel mass route.  An element whose
 * matrix is not finite (status reports it) can reach the blocks summed on the neighbouring lanes of its wave as well: check
 * status before using the values.
 *
 * The mesh is cut into PATCHES of nodes (plan built once per mesh by fs_fused_plan_build below from a map node ->
 * patch: fem_shell_amd/fused.py::partition_nodes, or any map whose patches touch at most max_patch_elems elements;
 * tests/cabi/driver.c drives the whole path from plain C).  A workgroup evaluates the elements touching its patch into LDS and
 * forms every node-pair block (I, J), I in the patch and I <= J (plus pattern-only blocks without a transposed
 * partner), writing it to (I, J) and transposed to (J, I).
 *   pat_eptr  [n_patches+1]  ranges into pat_elems; a patch holds <= max_patch_elems <= FS_FUSED_MAX_PATCH_ELEMS
 *   pat_elems               element indices (rows of xyz / prop_id) of each patch; position in the range = SLOT
 *   pat_rptr  [n_patches+1]  ranges into recs: one record per LANE, range starts multiples of 64 (whole waves)
 *   recs      fs_fused_rec.  A lane evaluates up to two contributions ev[0], ev[1] (step 0, step 1):
 *               ev = slot | ti << 8 | tj << 10 | transposed << 12 | active << 13 -- block (ti <= tj) of the element in
 *               `slot` (transposed: it contributes as its transpose).
 *             The contributions x0, x1, ... of a CSR block, in ELEMENT ORDER, occupy consecutive (lane, step)
 *             positions: n = 2: one lane (x0 + x1); n = 3, 4: two adjacent lanes, the second one FS_FL_CHAIN at
 *             position 1: ((x0 + x1) + x2) + x3; n > 4: more chained lanes (positions 1, 2, ...) inside one 16-lane
 *             row, the first one FS_FL_ZERO_START (0 + x0 + x1 + ..., the order of fs_csr_numeric_desc); n <= 1: the
 *             block takes one step of a lane, and the lane's other step may carry another such block
 *             (FS_FL_OUT0 -> step 0 is output block 0, FS_FL_OUT1 -> step 1 is output block 1, kept apart).
 *             FS_FL_PAD: the block has fewer than 4 contributions (its sum ends in + 0.0, as in the descriptor kernel).
 *             The LAST lane of a block carries FS_FL_OUT0 (or OUT1) and the block's destination: dst / stride = entry
 *             (0,0) and row distance of block (I,J) in `data` (as in fs_blk_desc), dst_t / stride_t those of block
 *             (J,I) (-1: none).  flags bits 8..15: length of the longest chain of the lane's wave (uniform over the
 *             wave, >= 1), bits 16..19: position of the lane in its chain.  slots: four bytes (output 0 direct,
 *             output 0 transposed, output 1 direct, output 1 transposed; 255 = none).  The "direct" bytes number the
 *             blocks a wave emits (outputs 0 first, then outputs 1, each by ascending dst): the block's staging slot;
 *             the "transposed" bytes rank the wave's transposed copies by ascending dst_t: the order in which they
 *             are read back from those slots, so that both sweeps of the wave's stores run through memory in order.
 *   flags     FS_FUSED_HAS_COUPLED: some property row has FS_PROP_COUPLED (0 selects the kernel without the
 *             coupling terms; a coupled row met anyway is reported as FS_ST_BAD_PROP)
 *             FS_FUSED_MASS: the same launch forms the consistent MASS matrix instead (MITC4.M, elements/MITC4.py:1038-1061,
 *             summed by the loop of MeshAssembler.assemble_mass_matrix, core/assembler.py:305-336): bit-identical to
 *             fs_shell_quad_batch(FS_WANT_M) followed by fs_csr_numeric_desc, same plan, no element matrices in memory
 *   status    [E] FS_ST_* bits or NULL
 */
typedef struct fs_fused_rec {
    int64_t dst[2];
    int64_t dst_t[2];
    int32_t stride[2];
    int32_t stride_t[2];
    uint32_t ev[2];
    uint32_t flags;
    uint32_t slots;
} fs_fused_rec; /* 64 bytes */
FS_STATIC_ASSERT(sizeof(fs_fused_rec) == 64 && offsetof(fs_fused_rec, stride) == 32 && offsetof(fs_fused_rec, ev) == 48 &&
                 offsetof(fs_fused_rec, flags) == 56, "fs_fused_rec layout");
#define FS_FL_CHAIN 1u
#define FS_FL_OUT0 2u
#define FS_FL_OUT1 4u
#define FS_FL_ZERO_START 8u
#define FS_FL_PAD 16u
#define FS_FUSED_MAX_PATCH_ELEMS 64
#define FS_FUSED_HAS_COUPLED 1u
#define FS_FUSED_MASS 2u   /* assemble the consistent mass matrix M instead of K (see above) */
int fs_fused_assemble_quad(const double *xyz, const int32_t *prop_id, const fs_shell_prop *props, int32_t n_props,
                           int64_t E, int64_t n_patches, const int64_t *pat_eptr, const int32_t *pat_elems,
                           const int64_t *pat_rptr, const fs_fused_rec *recs, int32_t max_patch_elems,
                           uint32_t flags, double *data, int32_t *status, void *stream);

/*
 * fs_fused_assemble_quad TRUSTS its plan: the kernel indexes LDS with the slots, ranks and chain positions of the
 * records and `data` with their destinations.  Plans of fs_fused_plan_build satisfy every invariant by construction; a
 * plan from anywhere else must pass fs_fused_plan_check ONCE before its first launch (device pointers; synchronises
 * `stream`, allocates 32 bytes).  Checked: every patch holds 1..max_patch_elems elements with indices in [0, E) and a
 * non-empty record range of whole waves; every active contribution names a slot of its patch with ti <= tj; chain
 * length (1..16, uniform over a wave) and chain positions are consistent; staging slots and transposed ranks are below
 * 128; every destination block (and transposed copy) lies inside [0, nnz) with even offsets and strides (the stores
 * are 16 bytes wide).  Returns FS_EINVAL (fs_last_error names the first violated rule) instead of a GPU fault.
 * No counterpart in the reference.
 */
int fs_fused_plan_check(int64_t n_patches, const int64_t *pat_eptr, const int32_t *pat_elems, const int64_t *pat_rptr,
                        const fs_fused_rec *recs, int64_t E, int32_t max_patch_elems, int64_t nnz, void *stream);

/*
 * The plan of fs_fused_assemble_quad, built on the HOST in native code (OpenMP over patches; all pointers of these
 * three calls are host memory, the caller uploads the exported arrays).  No counterpart in the reference (its loops need
 * no plan).
 *   conn        [E][4] node indices;  nbr_ptr [n_nodes+1], nbr [nblk]: the node graph of the CSR pattern, neighbours
 *               ascending (fs_csr_symbolic_finish's arrays copied to the host; ghost columns of a distributed run included)
 *   node_patch  [n_nodes] patch id of every node (any integers; nodes without a block row are ignored)
 * fs_fused_plan_build fails with FS_EINVAL when a patch touches more than max_patch_elems elements, when a node pair is
 * shared by more than 32 elements, or when a block with contributions has no transposed partner in the pattern.
 * fs_fused_plan_export copies the plan into caller buffers of n_patches+1, n_elem_slots, n_patches+1, n_recs entries
 * (n_evals, optional: number of contributions, for fill statistics).
 */
typedef struct fs_fused_plan fs_fused_plan;
int fs_fused_plan_build(const int64_t *conn, int64_t E, int64_t n_nodes, const int64_t *nbr_ptr, const int32_t *nbr,
                        const int64_t *node_patch, int32_t max_patch_elems, fs_fused_plan **plan, int64_t *n_patches,
                        int64_t *n_elem_slots, int64_t *n_recs);
int fs_fused_plan_export(const fs_fused_plan *plan, int64_t *pat_eptr, int32_t *pat_elems, int64_t *pat_rptr,
                         fs_fused_rec *recs, int64_t *n_evals);
void fs_fused_plan_free(fs_fused_plan *plan);

/*
 * The same plan built ON THE DEVICE (csrc/femshell_fused_plan_dev.hip): every pointer is DEVICE memory, the four plan
 * arrays come out bit for bit as fs_fused_plan_build / _export would give them, without the node graph travelling to the
 * host and gigabytes of records travelling back.  Two calls, like fs_csr_symbolic_begin / _finish:
 *   begin   sorts the (node, element) incidences and the lead nodes by patch (rocPRIM), counts elements and lanes of
 *           every patch with one workgroup per patch and returns the sizes the caller allocates: n_patches + 1,
 *           n_elem_slots, n_patches + 1, n_recs entries (n_evals, optional: number of contributions).  Synchronises
 *           `stream`; allocates and frees its own work space.  conn / nbr_ptr / nbr must stay alive until finish.
 *   finish  fills the caller's arrays (the same kernel, now writing) and frees the handle, also on failure.
 * node_patch: patch ids >= 0 (nodes without a block row are ignored).  Errors (FS_EINVAL) as fs_fused_plan_build; in
 * addition "device plan capacity exceeded" when a patch is beyond what one workgroup holds in LDS (more than 640 blocks
 * or 1024 lanes: pattern-only blocks of a distributed plan, or many chains of more than 4 contributions) -- such a mesh
 * goes through the host builder.  No counterpart in the reference (its loops need no plan).
 */
typedef struct fs_fused_plan_dev fs_fused_plan_dev;
int fs_fused_plan_device_begin(const int64_t *conn, int64_t E, int64_t n_nodes, const int64_t *nbr_ptr, const int32_t *nbr,
                               const int64_t *node_patch, int32_t max_patch_elems, fs_fused_plan_dev **plan,
                               int64_t *n_patches, int64_t *n_elem_slots, int64_t *n_recs, int64_t *n_evals, void *stream);
int fs_fused_plan_device_finish(fs_fused_plan_dev *plan, int64_t *pat_eptr, int32_t *pat_elems, int64_t *pat_rptr,
                                fs_fused_rec *recs, void *stream);
void fs_fused_plan_device_free(fs_fused_plan_dev *plan);

/*
 * The Newton path on the ASSEMBLED matrix: K_T(u) = K + sum_e dK_e(u) in one launch (csrc/femshell_fused_nl.hip).
 * MITC4.compute_tangent_stiffness (elements/MITC4.py:1246-1298, after update_configuration :1067-1084; laminates
 * MITC4_composite.py:499-621) returns the linear K plus the total-Lagrangian membrane correction T^T sym(K_L + K_sigma) T,
 * which lives in the translational 3 x 3 sub-blocks only; the loop of core/assembler.py:287-303 then adds the element
 * matrices.  K does not depend on u, so per Newton iterate fs_fused_assemble_quad_tangent reads the blocks (I,J), I <= J,
 * of the assembled linear matrix, forms the corrections of every element from xyz, u and the property rows (rows with
 * nonlinear == 0 add +0.0), adds them to the blocks IN ELEMENT ORDER and writes block (I,J) and its transpose (J,I) of
 * k_t -- no element matrices, 288 B read and 576 B written per node pair.
 *
 * The linear blocks come from a PACKED copy (36 contiguous doubles per block, in the order the plan's waves emit them),
 * made once per linear matrix:
 *   fs_fused_tangent_offsets  wave_off [n_recs / 64 + 1] (device, out): first packed block of every wave of records;
 *                             n_blocks (host, out): number of blocks I <= J.  Once per plan; synchronises `stream`,
 *                             allocates and frees scan work space.
 *   fs_fused_tangent_pack     packed [36 * n_blocks] (device, out) from the CSR values k_lin (fs_fused_assemble_quad or
 *                             fs_csr_numeric_desc; K is symmetric, only its blocks I <= J are read).
 *   fs_fused_assemble_quad_tangent
 *     u [n_nodes][6] nodal displacements; plan arrays, max_patch_elems: the plan of fs_fused_assemble_quad
 *     (fs_fused_plan_build / fs_fused_plan_device_*; fs_fused_plan_check applies);  pat_conn [n_elem_slots][4] int32: the
 *     connectivity rows of the plan's element slots, pat_conn[s] = conn[pat_elems[s]] (the node ids of a patch arrive with
 *     its element indices: one dependent load less in front of the gather of u);  k_t [nnz] out;
 *     f_nl [n_nodes][6] out, or NULL: the element-order sums of the NONLINEAR nodal forces of compute_internal_forces
 *     (what fs_shell_quad_fint_nl + fs_vec_gather_sum give: translational rows, zeros in the rotational rows), formed in
 *     the same launch -- stage 1 evaluates them, they ride on the sums of the diagonal blocks; every node with a block
 *     row is written once, other entries are left alone.  f_int = fs_csr_blk_matvec(K, u, y0 = f_nl);
 *     flags: FS_TANGENT_TWO_WGS_PER_CU selects the 32-slot variant of the kernel (two workgroups per CU; measured slower;
 *     ignored when f_nl is requested).
 * Agreement with fs_shell_quad_batch(FS_WANT_KT) + fs_csr_numeric_desc: rounding level (the correction is summed apart
 * from the linear blocks and evaluated without the cancellation of B~^T C B~ - B_L^T C B_L); k_t is exactly symmetric; the
 * same bits in every run.  All three: no counterpart in the reference beyond the lines cited (its loop needs no plan).
 */
#define FS_TANGENT_TWO_WGS_PER_CU 1u
int fs_fused_tangent_offsets(int64_t n_recs, const fs_fused_rec *recs, int64_t *wave_off, int64_t *n_blocks, void *stream);
int fs_fused_tangent_pack(int64_t n_recs, const fs_fused_rec *recs, const int64_t *wave_off, const double *k_lin,
                          double *packed, void *stream);
int fs_fused_assemble_quad_tangent(const double *xyz, const int32_t *pat_conn, const double *u, const int32_t *prop_id,
                                   const fs_shell_prop *props, int32_t n_props, int64_t E, int64_t n_patches,
                                   const int64_t *pat_eptr, const int32_t *pat_elems, const int64_t *pat_rptr,
                                   const fs_fused_rec *recs, int32_t max_patch_elems, const int64_t *wave_off,
                                   const double *k_lin_packed, double *k_t, double *f_nl, uint32_t flags, void *stream);

/*
 * The NONLINEAR part of the internal forces of MITC4.compute_internal_forces (elements/MITC4.py:1300-1341 after
 * update_configuration :1067-1084): fe [E][24] = T^T ( sum_g detJ h (B_L + B_NL)^T C e_GL  -  K_m u_loc ) in the
 * translational rows of each node, zeros in the rotational rows and for property rows with nonlinear == 0 -- what
 * fs_shell_quad_batch(FS_WANT_FINT) adds to K_e u_e.  With it  f_int = K u + sum_e fe  needs no element blocks:
 * fs_csr_blk_matvec on the assembled K (y0 = the element-order sums of fe, fs_vec_gather_sum).  u [n_nodes][6] is
 * gathered through conn [E][4] (device).  Same kernel as the FS_WANT_FINT path, same bits for these terms.
 */
int fs_shell_quad_fint_nl(const double *xyz, const int64_t *conn, const double *u, const int32_t *prop_id,
                          const fs_shell_prop *props, int32_t n_props, int64_t E, double *fe, void *stream);

/*
 * Fused evaluate-and-assemble, HEXA8 (3 dofs per node, full 2 x 2 x 2 rule): the CSR values of K straight from the
 * coordinates, no element matrices in memory.  Replaces the element evaluations HEXA8.K (elements/SOLID.py:543-567 with
 * _compute_jacobian / compute_B_matrix :164-266 and the rule / shape functions :1744-1818) TOGETHER with the
 * setValuesLocal(ADD_VALUES) loop over them (core/assembler.py:287-303).
 *
 * Gather form over PATCHES of nodes (plan built once per mesh by fs_fused_hex_plan_build from a map node -> patch).  A
 * workgroup (512 lanes) evaluates, for every element touching its patch, the physical shape-function gradients at the 8
 * Gauss points (scaled by sqrt(w detJ)) into LDS, then forms the COMPLETE 3-row CSR slabs of the patch's nodes -- every
 * node-pair block (I, J) with I in the patch, 27 of them for an interior node of a structured mesh -- stages them in LDS
 * and stores them as contiguous runs (the three rows of consecutively numbered nodes are adjacent in CSR).
 * C is constant over an element, so a contribution is contracted over the Gauss points FIRST:
 *     K_e[i][j] = sum_q w detJ B_i^T C B_j = C : G,   G = sum_q (w detJ) grad N_i grad N_j^T   (3 x 3),
 * 8 x 9 + 81 multiply-adds instead of 8 x 90 (the same sum of products re-associated: eps-level differences).  A
 * block's contributions are added in element order, 0 + x0 + x1 + ... (the reference's insertion order); when the model
 * has ONE property row the G of the block's elements are added in that order and contracted once per block
 * (sum_e C : G_e = C : sum_e G_e).  Every contribution is formed for the node pair ordered by GLOBAL index and transposed
 * where the block is (larger, smaller), the diagonal ones are symmetrised (SOLID.py:565), so the assembled matrix is
 * EXACTLY symmetric.
 *   pat_eptr / pat_elems   elements of each patch, position = LDS slot, <= FS_HEX_MAX_PATCH_ELEMS per patch (ascending; on
 *                          structured patches the builder picks a slot order that frees phase 2's LDS gathers of bank
 *                          conflicts, in which a few slots may repeat an element: any order is valid)
 *   pat_nptr / nodes       per patch node (ascending node index): offset of its slab in the patch's staging image
 *                          (doubles) and its number of neighbours m (slab = 3 rows x 3 m values)
 *   pat_rptr / runs        copy-out runs of the staging image: `len` doubles from staging offset `sbase` to data[gdst...]
 *   pat_lptr / steps       lanes of each patch (multiples of 64, <= FS_HEX_MAX_PATCH_LANES), 8 step words per lane:
 *                          slot | la << 7 | lb << 10 | active << 13 | flush << 14 | node_local << 15 | rank << 21 |
 *                          transposed << 28 -- contribution of the element in `slot` to block (I, J): la / lb = local
 *                          index of the smaller / larger of the global nodes I, J in that element, `transposed` (on
 *                          every step of the block): I > J, the block is the transpose of what (la, lb) gives; `flush`:
 *                          the block (node_local, rank-th neighbour) is complete after this step.  Blocks start at EVEN steps and
 *                          flush at ODD ones (a block with an odd number of contributions ends in an idle step; a
 *                          pattern-only block is two idle steps, the second flushing zeros)
 *   props [n_props][37] as for fs_solid_batch; status [E] (FS_ST_BAD_JAC where det J <= 1e-14, OR-ed in: zero it) or NULL
 * No counterpart in the reference beyond the lines above.
 */
typedef struct fs_hex_node { int32_t sbase; int32_t m; } fs_hex_node;                    /* 8 bytes */
typedef struct fs_hex_run { int64_t gdst; int32_t sbase; int32_t len; } fs_hex_run;      /* 16 bytes */
FS_STATIC_ASSERT(sizeof(fs_hex_node) == 8 && sizeof(fs_hex_run) == 16, "fused HEXA8 plan records");
#define FS_HEX_MAX_PATCH_ELEMS 100
#define FS_HEX_MAX_PATCH_NODES 64
#define FS_HEX_MAX_PATCH_LANES 512
#define FS_HEX_STAGE_DOUBLES 19300      /* staging image of a patch's CSR slabs (overlays the gradient image, 19 600 doubles) */
int fs_fused_assemble_hex8(const double *xyz, const int32_t *prop_id, const double *props, int32_t n_props, int64_t E,
                           int64_t n_patches, const int64_t *pat_eptr, const int32_t *pat_elems, const int64_t *pat_nptr,
                           const fs_hex_node *nodes, const int64_t *pat_rptr, const fs_hex_run *runs,
                           const int64_t *pat_lptr, const uint32_t *steps, double *data, int32_t *status, void *stream);
/* The consistent MASS matrix through the same plan and skeleton (SolidElement.M, elements/SOLID.py:569-597: rho sum_q
 * N^T N detJ w, blocks m I_3; and the loop of MeshAssembler.assemble_mass_matrix, core/assembler.py:305-336): the values
 * of M straight from the coordinates, <= 1e-15 from fs_solid_batch(FS_WANT_M) + fs_csr_numeric_desc, exactly symmetric. */
int fs_fused_assemble_hex8_mass(const double *xyz, const int32_t *prop_id, const double *props, int32_t n_props, int64_t E,
                                int64_t n_patches, const int64_t *pat_eptr, const int32_t *pat_elems, const int64_t *pat_nptr,
                                const fs_hex_node *nodes, const int64_t *pat_rptr, const fs_hex_run *runs,
                                const int64_t *pat_lptr, const uint32_t *steps, double *data, void *stream);
/* The kernel trusts its plan.  A plan that does not come from fs_fused_hex_plan_build must pass this check once before
 * its first launch (device pointers; synchronises `stream`, allocates 24 bytes): patch sizes within the limits above,
 * element indices in [0, E), node slabs and copy-out runs inside the staging image and inside [0, nnz), active steps
 * naming a slot of their patch, flushes on odd steps with node and neighbour rank in range.  FS_EINVAL otherwise
 * (fs_last_error names the first violated rule) -- never a GPU fault. */
int fs_fused_hex_plan_check(int64_t n_patches, const int64_t *pat_eptr, const int32_t *pat_elems, const int64_t *pat_nptr,
                            const fs_hex_node *nodes, const int64_t *pat_rptr, const fs_hex_run *runs,
                            const int64_t *pat_lptr, const uint32_t *steps, int64_t E, int64_t nnz, void *stream);
/* Host-side plan builder (all pointers host memory; OpenMP over patches).  conn [E][8]; nbr_ptr / nbr: node graph of the
 * CSR pattern (neighbours ascending); node_patch [n_nodes].  FS_EINVAL when a patch exceeds a limit above, when a node
 * pair is shared by more than 8 elements or when an element lists a node twice -- callers then keep the two-kernel route. */
typedef struct fs_fused_hex_plan fs_fused_hex_plan;
int fs_fused_hex_plan_build(const int64_t *conn, int64_t E, int64_t n_nodes, const int64_t *nbr_ptr, const int32_t *nbr,
                            const int64_t *node_patch, fs_fused_hex_plan **plan, int64_t *n_patches, int64_t *n_elem_slots,
                            int64_t *n_pat_nodes, int64_t *n_runs, int64_t *n_lanes);
int fs_fused_hex_plan_export(const fs_fused_hex_plan *plan, int64_t *pat_eptr, int32_t *pat_elems, int64_t *pat_nptr,
                             fs_hex_node *nodes, int64_t *pat_rptr, fs_hex_run *runs, int64_t *pat_lptr, uint32_t *steps,
                             int64_t *n_evals);
void fs_fused_hex_plan_free(fs_fused_hex_plan *plan);

/*
 * Fused evaluate-and-assemble, MITC3+ (3-node shells, 6 dofs per node): the CSR values of K straight from the coordinates,
 * no element matrices in memory.  Replaces MITC3.K / MITC3Composite.K of every element (elements/MITC3.py:935-1103,
 * MITC3_composite.py:204-240) TOGETHER with the setValuesLocal(ADD_VALUES) loop over them (core/assembler.py:287-303).
 * Gather form over PATCHES of nodes in the step layout of fs_fused_assemble_hex8: a workgroup (256 lanes) evaluates, for
 * every element touching its patch, frame, area and the shape / tying data of the three nodal functions and the bubble
 * at the 3 Hammer points into LDS, then forms the COMPLETE 6-row CSR slabs of the patch's nodes.  A lane works through
 * eight steps, one contribution (element, local node pair) each: the 6 x 6 block of the ordered local pair with the
 * device function of the element kernel (fs_shell_tri_batch forms the same bits), transposed where the block is
 * (larger, smaller); the contributions of a CSR block follow each other on one lane and are added in ELEMENT ORDER,
 * 0 + x0 + x1 + ... (the reference's insertion order); finished blocks leave from registers as 16-byte stores.
 *   xyz [E][3][3], prop_id [E], props / n_props as for fs_shell_tri_batch
 *   pat_eptr / pat_elems   elements of each patch (ascending; position = LDS slot), <= FS_TRI_MAX_PATCH_ELEMS per patch
 *   pat_nptr / nodes       per patch node (ascending node index): gbase = offset of entry (0, 0) of its first block in
 *                          `data` (36 x the node's first block index) and its number of neighbours m (row stride 6 m)
 *   pat_lptr / steps       lanes of each patch (multiples of 64, <= FS_TRI_MAX_PATCH_LANES), 8 step words per lane:
 *                          slot | li << 8 | lj << 10 | active << 13 | flush << 14 | node_local << 15 | rank << 22 --
 *                          contribution of the element in `slot` to block (I, J), li / lj = local index of I / J in that
 *                          element; `flush`: the block (node_local, rank-th neighbour) is complete after this step and is
 *                          stored (a pattern-only block is one idle step that flushes zeros)
 *   status [E] FS_ST_* bits (frame / area checks of the element kernel, FS_ST_BAD_PROP) or NULL
 * No counterpart in the reference beyond the lines above.
 */
typedef struct fs_tri_node { int64_t gbase; int32_t m; int32_t pad; } fs_tri_node;       /* 16 bytes */
FS_STATIC_ASSERT(sizeof(fs_tri_node) == 16, "fused MITC3+ plan records");
#define FS_TRI_MAX_PATCH_ELEMS 176
#define FS_TRI_MAX_PATCH_NODES 96
#define FS_TRI_MAX_PATCH_LANES 256
int fs_fused_assemble_tri(const double *xyz, const int32_t *prop_id, const fs_shell_prop *props, int32_t n_props, int64_t E,
                          int64_t n_patches, const int64_t *pat_eptr, const int32_t *pat_elems, const int64_t *pat_nptr,
                          const fs_tri_node *nodes, const int64_t *pat_lptr, const uint32_t *steps, double *data,
                          int32_t *status, void *stream);
/* The kernel trusts its plan; a plan that does not come from fs_fused_tri_plan_build must pass this check once before its
 * first launch (device pointers; synchronises `stream`, allocates 24 bytes): FS_EINVAL (fs_last_error names the first
 * violated rule) instead of a GPU fault. */
/* The consistent MASS matrix through the plan of fs_fused_assemble_tri: MITC3.M of every element (elements/MITC3.py:1105-1138)
 * and the loop of MeshAssembler.assemble_mass_matrix (core/assembler.py:305-336) in one launch, the bits of
 * fs_shell_tri_batch(FS_WANT_M) + fs_csr_numeric_desc (same block arithmetic, same element-order sums), no element mass
 * matrices in memory. */
int fs_fused_assemble_tri_mass(const double *xyz, const int32_t *prop_id, const fs_shell_prop *props, int32_t n_props, int64_t E,
                               int64_t n_patches, const int64_t *pat_eptr, const int32_t *pat_elems, const int64_t *pat_nptr,
                               const fs_tri_node *nodes, const int64_t *pat_lptr, const uint32_t *steps, double *data,
                               void *stream);
int fs_fused_tri_plan_check(int64_t n_patches, const int64_t *pat_eptr, const int32_t *pat_elems, const int64_t *pat_nptr,
                            const fs_tri_node *nodes, const int64_t *pat_lptr, const uint32_t *steps, int64_t E, int64_t nnz,
                            void *stream);
/* Host-side plan builder (all pointers host memory; OpenMP over patches).  conn [E][3]; nbr_ptr / nbr: node graph of the
 * CSR pattern (neighbours ascending); node_patch [n_nodes].  FS_EINVAL when a patch exceeds a limit above, when a node
 * pair is shared by more than 8 elements (a node of valence > 8) or when an element lists a node twice -- callers then
 * keep the two-kernel route. */
typedef struct fs_fused_tri_plan fs_fused_tri_plan;
int fs_fused_tri_plan_build(const int64_t *conn, int64_t E, int64_t n_nodes, const int64_t *nbr_ptr, const int32_t *nbr,
                            const int64_t *node_patch, fs_fused_tri_plan **plan, int64_t *n_patches, int64_t *n_elem_slots,
                            int64_t *n_pat_nodes, int64_t *n_lanes);
int fs_fused_tri_plan_export(const fs_fused_tri_plan *plan, int64_t *pat_eptr, int32_t *pat_elems, int64_t *pat_nptr,
                             fs_tri_node *nodes, int64_t *pat_lptr, uint32_t *steps, int64_t *n_evals);
void fs_fused_tri_plan_free(fs_fused_tri_plan *plan);

/* Symbolic companion of fs_csr_numeric_desc (pattern of core/assembler.py:222-241 for uniform-d
 * meshes): fills indptr[N*d+1] (int64) and indices[nnz] (int32) from the node graph:
 *   nbr_ptr [N+1] row pointer of the node graph (block rows), blk_row [nblk] node I and nbr [nblk] node J of
 *   every node-pair block, blocks sorted by (I,J). */
int fs_csr_pattern_blk(const int64_t *nbr_ptr, const int32_t *blk_row, const int32_t *nbr,
                       int64_t n_nodes, int64_t nblk, int32_t d, int64_t *indptr,
                       int32_t *indices, void *stream);

/*
 * Device symbolic phase: everything fs_csr_numeric_desc / fs_csr_pattern_blk need, built on the GPU
 * from the connectivity.  Replaces MeshAssembler._precompute_elements' DOF map and
 * _compute_sparsity_pattern (core/assembler.py:166-187, 222-241: per-row Python sets) by a stable
 * radix sort of node-pair keys I * n_nodes + J generated in element order, so each block's
 * contributions keep the reference's insertion order (core/assembler.py:295-300).
 *
 * fs_csr_symbolic_begin (synchronises `stream`; allocates its own workspace):
 *   conn       [E][nnmax] int64 device, node indices; rows of shorter elements padded with trailing -1
 *   entry_ptr  [E+1] exclusive prefix sum of nn_e^2, or NULL for uniform meshes (nn_e == nnmax)
 *   src_base   [E] index of element e's first FS_LAYOUT_SYMBLK block in the value buffer, or NULL
 *              for a uniform mesh stored in element order (e * nn(nn+1)/2)
 *   layout     layout of the element-matrix buffers the plan will index: FS_LAYOUT_SYMBLK, or
 *              FS_LAYOUT_SYMBLK_T16 (uniform quad meshes, entry_ptr == src_base == NULL)
 *   n_entries  sum of nn_e^2 (E * nnmax^2 for uniform meshes)
 *   extra_keys [n_extra] keys I * n_nodes + J of pattern-only blocks (ghost columns of a distributed
 *              run), or NULL
 *   nblk, ncontrib (host, out): number of node-pair blocks / of contributions
 * fs_csr_symbolic_finish (consumes the handle): nbr_ptr [n_nodes+1], nbr [nblk], blk_row [nblk],
 *   as documented at fs_csr_pattern_blk; blk_ptr [nblk+1] = range of each block's contributions in contrib
 *   [ncontrib] ((source block index << 1) | transposed, element order inside a block); desc [nblk] (may be NULL)
 *   as documented at fs_blk_desc for d dofs per node.
 * fs_csr_symbolic_free: drop a handle without finishing.
 */
typedef struct fs_symbolic fs_symbolic;
int fs_csr_symbolic_begin(const int64_t *conn, const int64_t *entry_ptr, const int64_t *src_base, int64_t E,
                          int32_t nnmax, uint32_t layout, int64_t n_entries, int64_t n_nodes, const int64_t *extra_keys,
                          int64_t n_extra, fs_symbolic **handle, int64_t *nblk, int64_t *ncontrib,
                          void *stream);
int fs_csr_symbolic_finish(fs_symbolic *handle, int32_t d, int64_t *nbr_ptr, int32_t *nbr, int32_t *blk_row,
                           int64_t *blk_ptr, int64_t *contrib, fs_blk_desc *desc, void *stream);
void fs_csr_symbolic_free(fs_symbolic *handle);

/*
 * Dirichlet reduction and mass lumping on the device CSR (the step right after the assembly path).
 * The reference does this through PETSc (core/bc.py:116-239, solvers/fsi/linear_dynamic.py:227-234);
 * no reference test pins these PETSc calls, so their parity is pinned through the reduced solve of
 * the Scordelis-Lo fixture (tests/golden/asm_scordelis_16.npz).
 *
 * fs_csr_zero_rows: Mat.zeroRows(rows, diag, x, b) -- every entry of the listed rows becomes 0, the
 *   diagonal entry `diag`, and (x, b non-NULL) b[r] = diag * x[r]; the pattern is kept (explicit zeros).
 * fs_csr_reduce_symbolic (synchronises `stream`): free_rank[ndof] (rank among the free dofs, -1 for
 *   fixed ones) and the row pointer red_indptr[n_free+1] (caller provides ndof+1 entries) of the
 *   free x free sub-matrix, createSubMatrix(free_is, free_is); n_free / nnz_red returned on the host.
 * fs_csr_reduce_fill: column indices (renumbered) and values of the sub-matrix -- either may be NULL,
 *   so K and M share one symbolic pass -- and, with f_red, the adjusted load
 *   f_red = f[free] - K[free, fixed] u_fixed (u_fixed: full-length vector read at fixed dofs; NULL = 0).
 * fs_expand_solution: u_full[free] = u_red, u_full[fixed] = u_fixed (NULL = 0)  (core/bc.py:241-290).
 * fs_csr_row_sums: out[r] = sum of row r (M.getRowSum: lumped mass diagonal).
 */
int fs_csr_zero_rows(const int64_t *indptr, const int32_t *indices, double *data, const int64_t *rows,
                     int64_t n_rows, double diag, const double *x, double *b, void *stream);
int fs_csr_reduce_symbolic(const int64_t *indptr, const int32_t *indices, int64_t ndof,
                           const uint8_t *fixed_mask, int32_t *free_rank, int64_t *red_indptr,
                           int64_t *n_free, int64_t *nnz_red, void *stream);
int fs_csr_reduce_fill(const int64_t *indptr, const int32_t *indices, const double *data, int64_t ndof,
                       const int32_t *free_rank, const int64_t *red_indptr, int32_t *red_indices,
                       double *red_data, const double *u_fixed, const double *f, double *f_red, void *stream);
int fs_expand_solution(const double *u_red, const int32_t *free_rank, const double *u_fixed, int64_t ndof,
                       double *u_full, void *stream);
int fs_csr_row_sums(const int64_t *indptr, const double *data, int64_t nrows, double *out, void *stream);

/*
 * y = A x (+ y0, NULL = none) for a matrix in the block layout of fs_csr_pattern_blk (d = 3 or 6 dofs per node; the
 * arrays of fs_csr_symbolic_finish).  New with the Newton path: the linear part K u of the internal forces
 * (compute_internal_forces, elements/MITC4.py:1300-1341: f = K u_e + nonlinear membrane terms, summed by the loop of
 * core/assembler.py) taken from the ASSEMBLED K instead of from 2 880 B of element blocks per element.  One writer per
 * row, partial products added from left to right: the same bits in every run.  y must not alias x.
 */
int fs_csr_blk_matvec(const int64_t *nbr_ptr, const int32_t *nbr, const int32_t *blk_row, int64_t n_nodes, int32_t d,
                      const double *data, const double *x, const double *y0, double *y, void *stream);

/* f[dof] = sum of fe contributions in element order (core/assembler.py:338-384 and the new
 * assemble_internal_forces): same segmented form as fs_csr_numeric. */
int fs_vec_gather_sum(const double *vals, const int64_t *src, const int64_t *seg_ptr,
                      int64_t n, double *out, void *stream);

/* Multi-GPU interface exchange (new; the reference is serial, core/assembler.py:37,257-262 only
 * stores a communicator).  Ranks own block rows; a rank packs the partial sums of the rows it
 * does not own (fs_halo_pack: buf[t] = data[idx[t]]), ships them to the owner over RCCL, and the
 * owner adds them in ascending neighbour-rank order (fs_halo_add: data[idx[t]] += buf[t]; idx
 * unique within one message). */
int fs_halo_pack(const double *data, const int64_t *idx, int64_t n, double *buf, void *stream);
int fs_halo_add(const double *buf, const int64_t *idx, int64_t n, double *data, void *stream);

/* u_e[e][a] = u[dofs[e][a]]  (gather for KT / f_int re-assembly) */
int fs_gather_dofs(const double *u, const int64_t *dofs, int64_t n, double *u_e, void *stream);

/*
 * Inertial nodal forces of a rotating frame: InertialForcesCalculator.compute_centrifugal_force / compute_coriolis_force
 * / compute_euler_force / compute_all_inertial_forces (solvers/fsi/corotational.py:347-552), the rotor load that goes
 * with the lumped mass of the assembled M.
 *   coords [N][3], vel [N][3] (NULL unless FS_INERTIAL_CORIOLIS), mass [N]: device;  axis_host (unit vector),
 *   center_host (NULL = origin): 3 doubles each in HOST memory;  parts: FS_INERTIAL_* bits;  F [N][3] = their sum.
 */
#define FS_INERTIAL_CENTRIFUGAL 1u
#define FS_INERTIAL_CORIOLIS 2u
#define FS_INERTIAL_EULER 4u
int fs_inertial_forces(const double *coords, const double *vel, const double *mass, int64_t N, double omega,
                       double alpha, const double *axis_host, const double *center_host, uint32_t parts, double *F,
                       void *stream);

/*
 * fs_shell_quad_fint_noalloc: NO counterpart in the reference.  It is for callers that cannot allocate device memory
 * (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc): the internal forces of
 * fs_shell_quad_batch(want = FS_WANT_FINT, layout = FS_LAYOUT_SYMBLK_T16, KTe = workspace) -- update_configuration(u_e)
 * + compute_internal_forces() per element, fe [E][24] -- without the KTe workspace argument.  The workspace (the
 * elements' linear blocks, 2 880 B per element) lives in a static device array of the library, 47 185 920 B on every
 * GPU the entry point is used on, held until the process ends.  The call allocates nothing (no hipMalloc / hipFree).
 *   E > FS_FINT_NOALLOC_MAX_E  ->  FS_EINVAL (use fs_shell_quad_batch with your own workspace)
 * Set-up: the array's device address is resolved once per GPU -- by fs_device_info(dev, ...) when `dev` is the
 * current device (call it once per GPU, with that GPU current, BEFORE a graph capture that contains this entry point),
 * else by the first call on that GPU (which must then happen outside a capture).  `stream` must belong to the current
 * device.
 * All calls on one GPU share that GPU's array: calls that overlap on one GPU (different streams or threads) race;
 * serialise them.  Nothing has to be cleared between calls (every workspace block that is read is rewritten first).
 */
#define FS_FINT_NOALLOC_MAX_E 16384
int fs_shell_quad_fint_noalloc(const double *xyz, const int32_t *prop_id, const fs_shell_prop *props,
                               int32_t n_props, const double *u_e, int64_t E, double *fe, int32_t *status,
                               void *stream);

/* Diagnostic, no counterpart in the reference: `waves` 64-lane workgroups each run iters * 64 fp64 FMAs per lane in
 * 8 independent chains (2 * 64 * 64 * iters * waves flop per launch); out [waves * 64].  bench.py times it to report
 * the box's own fp64 ceiling beside the vendor peak. */
int fs_bench_fp64_fma(int32_t waves, int32_t iters, double *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif
