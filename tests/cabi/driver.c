/*
 * Plain-C host for libfemshell_hip.so: proves that include/femshell_hip.h is a C header and that the hot
 * path (MITC4+ element matrices -> device symbolic phase -> CSR assembly -> lumped mass) runs from a
 * process that contains neither Python nor torch -- the way a cgo / JNI / ctypes binding would drive it.
 * Test infrastructure (tests/test_gpu_cabi.py writes the input file and checks the output file against the
 * oracle); built by fem_shell_amd/csrc/Makefile with gcc, linked against libamdhip64 for hipMalloc/hipMemcpy.
 *
 *   usage: cabi_driver in.bin out.bin
 *   in :  int64 E, n_nodes, n_props; f64 xyz[E][4][3]; i32 prop_id[E]; fs_shell_prop props[n_props];
 *         i64 conn[E][4]; i64 node_patch[n_nodes] (the patch of every node, for the fused path)
 *   out:  int64 nblk, nnz, ndof; i64 indptr[ndof+1]; i32 indices[nnz]; f64 K[nnz]; f64 M[nnz];
 *         f64 lumped[ndof]; i32 status[E]
 * Besides, the fused route (fs_fused_plan_build on the host, fs_fused_assemble_quad on the device) must reproduce the K of
 * the two-kernel route bit for bit: exit code 8 otherwise.
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "femshell_hip.h"

#define HIPCHK(x)                                                                          \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e_));     \
            return 2;                                                                      \
        }                                                                                  \
    } while (0)
#define FSCHK(x)                                                                           \
    do {                                                                                   \
        int r_ = (x);                                                                      \
        if (r_ != FS_OK) {                                                                 \
            fprintf(stderr, "%s:%d: %d %s\n", __FILE__, __LINE__, r_, fs_last_error());    \
            return 3;                                                                      \
        }                                                                                  \
    } while (0)

static void *rd(FILE *f, size_t n) {
    void *p = malloc(n ? n : 1);
    if (!p || fread(p, 1, n, f) != n) {
        fprintf(stderr, "short read (%zu bytes)\n", n);
        exit(4);
    }
    return p;
}

static int to_dev(void **d, const void *h, size_t n) {
    HIPCHK(hipMalloc(d, n ? n : 1));
    if (h) HIPCHK(hipMemcpy(*d, h, n, hipMemcpyHostToDevice));
    return 0;
}

static int wr_dev(FILE *f, const void *d, size_t n) {
    void *h = malloc(n ? n : 1);
    HIPCHK(hipMemcpy(h, d, n, hipMemcpyDeviceToHost));
    fwrite(h, 1, n, f);
    free(h);
    return 0;
}

int main(int argc, char **argv) {
    if (argc != 3) return 1;
    char name[128];
    int cus = 0;
    FSCHK(fs_device_info(0, name, (int)sizeof name, &cus));
    FILE *in = fopen(argv[1], "rb");
    if (!in) return 1;
    int64_t hdr[3];
    if (fread(hdr, 8, 3, in) != 3) return 4;
    const int64_t E = hdr[0], N = hdr[1], P = hdr[2], ndof = 6 * N;
    if (sizeof(fs_shell_prop) != 368 || sizeof(fs_blk_desc) != 48) return 5;
    double *xyz = rd(in, (size_t)E * 96);
    int32_t *pid = rd(in, (size_t)E * 4);
    fs_shell_prop *props = rd(in, (size_t)P * sizeof(fs_shell_prop));
    int64_t *conn = rd(in, (size_t)E * 32);
    int64_t *node_patch = rd(in, (size_t)N * 8);
    fclose(in);

    hipStream_t s;
    HIPCHK(hipStreamCreate(&s));
    void *d_KT = NULL, *d_fnl = NULL, *d_f = NULL;       /* Newton iterate (below) */
    double *h_u_out = NULL;
    void *d_xyz, *d_pid, *d_props, *d_conn, *d_Ke, *d_Me, *d_st;
    const size_t tile_bytes = (size_t)((E + 15) / 16) * 160 * 36 * 8;
    if (to_dev(&d_xyz, xyz, (size_t)E * 96) || to_dev(&d_pid, pid, (size_t)E * 4) ||
        to_dev(&d_props, props, (size_t)P * sizeof(fs_shell_prop)) || to_dev(&d_conn, conn, (size_t)E * 32) ||
        to_dev(&d_Ke, NULL, tile_bytes) || to_dev(&d_Me, NULL, tile_bytes) || to_dev(&d_st, NULL, (size_t)E * 4))
        return 2;
    HIPCHK(hipMemsetAsync(d_st, 0, (size_t)E * 4, s));

    /* element matrices (the body of MeshAssembler._precompute_elements) */
    FSCHK(fs_shell_quad_batch(d_xyz, d_pid, d_props, (int32_t)P, NULL, E, FS_WANT_K | FS_WANT_M, FS_LAYOUT_SYMBLK_T16,
                              d_Ke, d_Me, NULL, NULL, d_st, s));

    /* symbolic phase (_compute_sparsity_pattern) */
    fs_symbolic *h = NULL;
    int64_t nblk = 0, ncontrib = 0;
    FSCHK(fs_csr_symbolic_begin(d_conn, NULL, NULL, E, 4, FS_LAYOUT_SYMBLK_T16, E * 16, N, NULL, 0, &h, &nblk, &ncontrib, s));
    const int64_t nnz = nblk * 36;
    void *d_nbr_ptr, *d_nbr, *d_blk_row, *d_blk_ptr, *d_contrib, *d_desc, *d_indptr, *d_indices, *d_K, *d_M, *d_lump;
    if (to_dev(&d_nbr_ptr, NULL, (size_t)(N + 1) * 8) || to_dev(&d_nbr, NULL, (size_t)nblk * 4) ||
        to_dev(&d_blk_row, NULL, (size_t)nblk * 4) || to_dev(&d_blk_ptr, NULL, (size_t)(nblk + 1) * 8) ||
        to_dev(&d_contrib, NULL, (size_t)ncontrib * 8) || to_dev(&d_desc, NULL, (size_t)nblk * sizeof(fs_blk_desc)) ||
        to_dev(&d_indptr, NULL, (size_t)(ndof + 1) * 8) || to_dev(&d_indices, NULL, (size_t)nnz * 4) ||
        to_dev(&d_K, NULL, (size_t)nnz * 8) || to_dev(&d_M, NULL, (size_t)nnz * 8) || to_dev(&d_lump, NULL, (size_t)ndof * 8))
        return 2;
    FSCHK(fs_csr_symbolic_finish(h, 6, d_nbr_ptr, d_nbr, d_blk_row, d_blk_ptr, d_contrib, d_desc, s));
    FSCHK(fs_csr_pattern_blk(d_nbr_ptr, d_blk_row, d_nbr, N, nblk, 6, d_indptr, d_indices, s));

    /* numeric phase (assemble_stiffness_matrix / assemble_mass_matrix) + lumping */
    FSCHK(fs_csr_numeric_desc(d_Ke, d_desc, d_contrib, nblk, 6, d_K, s));
    FSCHK(fs_csr_numeric_desc(d_Me, d_desc, d_contrib, nblk, 6, d_M, s));
    FSCHK(fs_csr_row_sums(d_indptr, d_M, ndof, d_lump, s));
    HIPCHK(hipStreamSynchronize(s));

    /* the fused route: plan on the host from the pattern's node graph, one launch, no element matrices -- same bits */
    {
        int64_t *h_nbr_ptr = malloc((size_t)(N + 1) * 8);
        int32_t *h_nbr = malloc((size_t)nblk * 4);
        HIPCHK(hipMemcpy(h_nbr_ptr, d_nbr_ptr, (size_t)(N + 1) * 8, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(h_nbr, d_nbr, (size_t)nblk * 4, hipMemcpyDeviceToHost));
        fs_fused_plan *fp = NULL;
        int64_t npat = 0, nslots = 0, nrecs = 0;
        FSCHK(fs_fused_plan_build(conn, E, N, h_nbr_ptr, h_nbr, node_patch, FS_FUSED_MAX_PATCH_ELEMS, &fp, &npat, &nslots, &nrecs));
        int64_t *h_eptr = malloc((size_t)(npat + 1) * 8), *h_rptr = malloc((size_t)(npat + 1) * 8);
        int32_t *h_elems = malloc((size_t)nslots * 4 + 4);
        fs_fused_rec *h_recs = malloc((size_t)nrecs * sizeof(fs_fused_rec) + 1);
        FSCHK(fs_fused_plan_export(fp, h_eptr, h_elems, h_rptr, h_recs, NULL));
        fs_fused_plan_free(fp);
        void *d_eptr, *d_rptr, *d_elems, *d_recs, *d_Kf, *d_stf;
        if (to_dev(&d_eptr, h_eptr, (size_t)(npat + 1) * 8) || to_dev(&d_rptr, h_rptr, (size_t)(npat + 1) * 8) ||
            to_dev(&d_elems, h_elems, (size_t)nslots * 4) || to_dev(&d_recs, h_recs, (size_t)nrecs * sizeof(fs_fused_rec)) ||
            to_dev(&d_Kf, NULL, (size_t)nnz * 8) || to_dev(&d_stf, NULL, (size_t)E * 4))
            return 2;
        /* a plan from anywhere else than fs_fused_plan_build must pass fs_fused_plan_check before its first launch: the
         * built plan passes; a corrupted copy (slot beyond the patch, destination past the value array, empty patch)
         * comes back as FS_EINVAL -- it is never launched */
        FSCHK(fs_fused_plan_check(npat, d_eptr, d_elems, d_rptr, d_recs, E, FS_FUSED_MAX_PATCH_ELEMS, nnz, s));
        {
            fs_fused_rec *bad = malloc((size_t)nrecs * sizeof(fs_fused_rec) + 1);
            void *d_bad;
            int64_t first_out = -1, first_act = -1;
            for (int64_t i = 0; i < nrecs; i++) {
                if (first_out < 0 && (h_recs[i].flags & FS_FL_OUT0)) first_out = i;
                if (first_act < 0 && (h_recs[i].ev[0] & (1u << 13))) first_act = i;
            }
            if (first_out < 0 || first_act < 0) return 10;
            memcpy(bad, h_recs, (size_t)nrecs * sizeof(fs_fused_rec));
            bad[first_act].ev[0] |= 255u;
            if (to_dev(&d_bad, bad, (size_t)nrecs * sizeof(fs_fused_rec))) return 2;
            if (fs_fused_plan_check(npat, d_eptr, d_elems, d_rptr, d_bad, E, FS_FUSED_MAX_PATCH_ELEMS, nnz, s) != FS_EINVAL ||
                !strstr(fs_last_error(), "slot"))
                return 10;
            memcpy(bad, h_recs, (size_t)nrecs * sizeof(fs_fused_rec));
            bad[first_out].dst[0] = nnz - 4;
            HIPCHK(hipMemcpy(d_bad, bad, (size_t)nrecs * sizeof(fs_fused_rec), hipMemcpyHostToDevice));
            if (fs_fused_plan_check(npat, d_eptr, d_elems, d_rptr, d_bad, E, FS_FUSED_MAX_PATCH_ELEMS, nnz, s) != FS_EINVAL ||
                !strstr(fs_last_error(), "destination"))
                return 10;
            if (npat >= 2) {
                int64_t *be = malloc((size_t)(npat + 1) * 8);
                void *d_be;
                memcpy(be, h_eptr, (size_t)(npat + 1) * 8);
                be[1] = be[0];                                   /* patch 0 without elements */
                if (to_dev(&d_be, be, (size_t)(npat + 1) * 8)) return 2;
                if (fs_fused_plan_check(npat, d_be, d_elems, d_rptr, d_recs, E, FS_FUSED_MAX_PATCH_ELEMS, nnz, s) != FS_EINVAL)
                    return 10;
            }
        }
        uint32_t fl = 0;
        for (int64_t i = 0; i < P; i++)
            if (props[i].flags & FS_PROP_COUPLED) fl = FS_FUSED_HAS_COUPLED;
        FSCHK(fs_fused_assemble_quad(d_xyz, d_pid, d_props, (int32_t)P, E, npat, d_eptr, d_elems, d_rptr, d_recs,
                                     FS_FUSED_MAX_PATCH_ELEMS, fl, d_Kf, d_stf, s));
        HIPCHK(hipStreamSynchronize(s));
        double *k2 = malloc((size_t)nnz * 8), *kf = malloc((size_t)nnz * 8);
        int32_t *st2 = malloc((size_t)E * 4), *stf = malloc((size_t)E * 4);
        HIPCHK(hipMemcpy(k2, d_K, (size_t)nnz * 8, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(kf, d_Kf, (size_t)nnz * 8, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(st2, d_st, (size_t)E * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(stf, d_stf, (size_t)E * 4, hipMemcpyDeviceToHost));
        /* the two routes sum the same blocks in the same order; a contribution received transposed is formed by the fused
         * kernel as the block of the swapped node pair, so the values agree to rounding, the status words exactly */
        {
            double kmax = 0.0, dmax = 0.0;
            for (int64_t i = 0; i < nnz; i++) {
                if (fabs(k2[i]) > kmax) kmax = fabs(k2[i]);
                if (!(fabs(kf[i] - k2[i]) <= dmax)) dmax = fabs(kf[i] - k2[i]);
            }
            if (!(dmax <= 2e-14 * kmax) || memcmp(st2, stf, (size_t)E * 4) != 0) return 8;
        }
        /* one Newton iterate on the assembled matrix (update_configuration + compute_tangent_stiffness /
         * compute_internal_forces of every element + the assembly loops): packed linear blocks once, then K_T and the
         * nonlinear nodal forces in one launch, f_int = K u + f_nl */
        {
            int32_t *h_pc = malloc((size_t)nslots * 16 + 16);
            for (int64_t q = 0; q < nslots; q++)
                for (int c = 0; c < 4; c++) h_pc[4 * q + c] = (int32_t)conn[4 * (int64_t)h_elems[q] + c];
            double *h_u = malloc((size_t)ndof * 8);
            for (int64_t n = 0; n < N; n++)
                for (int c = 0; c < 6; c++) h_u[6 * n + c] = 2e-5 * (double)((n * 7 + c * 13) % 23 - 11) * (c < 3 ? 1.0 : 10.0);
            void *d_pc, *d_u, *d_woff, *d_packed;
            int64_t nup = 0;
            if (to_dev(&d_pc, h_pc, (size_t)nslots * 16) || to_dev(&d_u, h_u, (size_t)ndof * 8) ||
                to_dev(&d_woff, NULL, (size_t)(nrecs / 64 + 1) * 8) || to_dev(&d_KT, NULL, (size_t)nnz * 8) ||
                to_dev(&d_fnl, NULL, (size_t)ndof * 8) || to_dev(&d_f, NULL, (size_t)ndof * 8))
                return 2;
            FSCHK(fs_fused_tangent_offsets(nrecs, d_recs, d_woff, &nup, s));
            if (nup != (nblk + N) / 2) return 10;                  /* the blocks I <= J of a symmetric pattern */
            if (to_dev(&d_packed, NULL, (size_t)nup * 288)) return 2;
            FSCHK(fs_fused_tangent_pack(nrecs, d_recs, d_woff, d_Kf, d_packed, s));
            HIPCHK(hipMemsetAsync(d_fnl, 0, (size_t)ndof * 8, s));
            FSCHK(fs_fused_assemble_quad_tangent(d_xyz, d_pc, d_u, d_pid, d_props, (int32_t)P, E, npat, d_eptr, d_elems,
                                                 d_rptr, d_recs, FS_FUSED_MAX_PATCH_ELEMS, d_woff, d_packed, d_KT, d_fnl,
                                                 0, s));
            FSCHK(fs_csr_blk_matvec(d_nbr_ptr, d_nbr, d_blk_row, N, 6, d_Kf, d_u, d_fnl, d_f, s));
            HIPCHK(hipStreamSynchronize(s));
            if (fs_fused_tangent_offsets(nrecs + 1, d_recs, d_woff, &nup, s) != FS_EINVAL ||
                fs_fused_tangent_pack(nrecs, d_recs, NULL, d_Kf, d_packed, s) != FS_EINVAL ||
                fs_fused_assemble_quad_tangent(d_xyz, NULL, d_u, d_pid, d_props, (int32_t)P, E, npat, d_eptr, d_elems, d_rptr,
                                               d_recs, FS_FUSED_MAX_PATCH_ELEMS, d_woff, d_packed, d_KT, NULL, 0, s) != FS_EINVAL ||
                fs_csr_blk_matvec(d_nbr_ptr, d_nbr, d_blk_row, N, 5, d_Kf, d_u, NULL, d_f, s) != FS_EINVAL ||
                fs_shell_quad_fint_nl(d_xyz, NULL, d_u, d_pid, d_props, (int32_t)P, E, d_f, s) != FS_EINVAL)
                return 6;
            h_u_out = h_u;
        }
        /* the mass matrix through the same plan (FS_FUSED_MASS): the bits of fs_shell_quad_batch(FS_WANT_M) + fs_csr_numeric_desc */
        FSCHK(fs_fused_assemble_quad(d_xyz, d_pid, d_props, (int32_t)P, E, npat, d_eptr, d_elems, d_rptr, d_recs,
                                     FS_FUSED_MAX_PATCH_ELEMS, FS_FUSED_MASS, d_Kf, NULL, s));
        HIPCHK(hipStreamSynchronize(s));
        HIPCHK(hipMemcpy(k2, d_M, (size_t)nnz * 8, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(kf, d_Kf, (size_t)nnz * 8, hipMemcpyDeviceToHost));
        if (memcmp(k2, kf, (size_t)nnz * 8) != 0) return 9;
        if (fs_fused_plan_build(conn, E, N, h_nbr_ptr, h_nbr, NULL, 64, &fp, &npat, &nslots, &nrecs) != FS_EINVAL) return 6;
    }

    /* error behaviour: bad arguments come back as codes + message, never as a crash */
    if (fs_shell_quad_batch(NULL, d_pid, d_props, (int32_t)P, NULL, E, FS_WANT_K, FS_LAYOUT_SYMBLK, d_Ke, NULL, NULL,
                            NULL, NULL, s) != FS_EINVAL || !strstr(fs_last_error(), "null input"))
        return 6;
    if (fs_shell_quad_batch(d_xyz, d_pid, d_props, (int32_t)P, NULL, E, FS_WANT_K, FS_LAYOUT_DENSE, d_Ke, NULL, NULL,
                            NULL, NULL, s) != FS_EINVAL)
        return 6;

    if (fs_csr_numeric_desc(NULL, d_desc, d_contrib, nblk, 6, d_K, s) != FS_EINVAL || !strstr(fs_last_error(), "null") ||
        fs_halo_pack(NULL, NULL, 5, NULL, s) != FS_EINVAL || fs_blk_expand(NULL, FS_LAYOUT_SYMBLK, E, 4, 6, NULL, s) != FS_EINVAL ||
        fs_csr_pattern_blk(NULL, d_blk_row, d_nbr, N, nblk, 6, d_indptr, d_indices, s) != FS_EINVAL ||
        fs_shell_quad_geom(NULL, E, NULL, NULL, NULL, NULL, s) != FS_EINVAL)
        return 6;
    /* property ids outside the table: never an out-of-bounds read -- row 0 is used and FS_ST_BAD_PROP is reported */
    if (E >= 3) {
        int32_t *bad = malloc((size_t)E * 4), *st = malloc((size_t)E * 4);
        double *k_ref = malloc(3 * 2880), *k_bad = malloc(3 * 2880);
        void *d_bad, *d_st2, *d_K2;
        memcpy(bad, pid, (size_t)E * 4);
        bad[1] = (int32_t)P + 1000000;
        bad[2] = -7;
        const int32_t zero_id[3] = {pid[0], 0, 0};      /* what the library must have used instead */
        void *d_zero;
        if (to_dev(&d_bad, bad, (size_t)E * 4) || to_dev(&d_st2, NULL, (size_t)E * 4) || to_dev(&d_zero, zero_id, 12) ||
            to_dev(&d_K2, NULL, (size_t)E * 2880))
            return 2;
        FSCHK(fs_shell_quad_batch(d_xyz, d_bad, d_props, (int32_t)P, NULL, E, FS_WANT_K, FS_LAYOUT_SYMBLK, d_K2, NULL,
                                  NULL, NULL, d_st2, s));
        HIPCHK(hipStreamSynchronize(s));
        HIPCHK(hipMemcpy(st, d_st2, (size_t)E * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(k_bad, d_K2, 3 * 2880, hipMemcpyDeviceToHost));
        FSCHK(fs_shell_quad_batch(d_xyz, d_zero, d_props, (int32_t)P, NULL, 3, FS_WANT_K, FS_LAYOUT_SYMBLK, d_K2, NULL,
                                  NULL, NULL, NULL, s));
        HIPCHK(hipStreamSynchronize(s));
        HIPCHK(hipMemcpy(k_ref, d_K2, 3 * 2880, hipMemcpyDeviceToHost));
        if (!(st[1] & FS_ST_BAD_PROP) || !(st[2] & FS_ST_BAD_PROP) || (st[0] & FS_ST_BAD_PROP) ||
            memcmp(k_ref, k_bad, 3 * 2880) != 0)
            return 7;
        for (int64_t e = 3; e < E; e++)
            if (st[e] & FS_ST_BAD_PROP) return 7;
    }

    FILE *out = fopen(argv[2], "wb");
    if (!out) return 1;
    int64_t oh[3] = {nblk, nnz, ndof};
    fwrite(oh, 8, 3, out);
    if (wr_dev(out, d_indptr, (size_t)(ndof + 1) * 8) || wr_dev(out, d_indices, (size_t)nnz * 4) ||
        wr_dev(out, d_K, (size_t)nnz * 8) || wr_dev(out, d_M, (size_t)nnz * 8) || wr_dev(out, d_lump, (size_t)ndof * 8) ||
        wr_dev(out, d_st, (size_t)E * 4) || wr_dev(out, d_KT, (size_t)nnz * 8) || wr_dev(out, d_f, (size_t)ndof * 8))
        return 2;
    fwrite(h_u_out, 8, (size_t)ndof, out);
    fclose(out);
    printf("cabi_driver ok: %s (%d CUs), E=%lld nblk=%lld nnz=%lld\n", name, cus, (long long)E, (long long)nblk,
           (long long)nnz);
    return 0;
}
