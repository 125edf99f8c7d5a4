// Nystrom executor: the launch sequence of ker_svd_sparsify (SOBER/_rchq.py:34-39) from the Gram matrix on -- PSD
// repair (make_cov_psd, SOBER/_utils.py:131-157) and the range finder of torch.svd_lowrank -- behind ONE C call, plus
// the projection P = [U, -U T] and the copy of every flag to pinned memory.  Pure host code: it only sequences the
// library's own entry points on the caller's stream (what sober_amd/_ops_hip.py issued one ctypes call at a time:
// ~40 calls and a dozen allocations per step, "at the host's pace").  No host decision inside: the caller reads the
// flags once, behind whatever it enqueues next.
#include <hip/hip_runtime.h>

#include "../../include/sober_hip.h"
#include "internal.hpp"

extern "C" int sober_nystrom_job_size(void) { return (int)sizeof(sober_nystrom_job); }

#define NX_TRY(call)                 \
    do {                             \
        const int rc_ = (call);      \
        if (rc_ != 0) return rc_;    \
    } while (0)
#define NX_HIP(call)                                   \
    do {                                               \
        const hipError_t e_ = (call);                  \
        if (e_ != hipSuccess) return (int)e_;          \
    } while (0)

extern "C" int64_t sober_nystrom_flags_bytes(int n_rungs, int niter) {
    const int64_t n_orth2 = 2 * (1 + 2 * (int64_t)niter);
    return 8 * (n_rungs + 1 + n_orth2) + 4 * (2 + n_rungs + n_orth2);
}

// CholeskyQR of the M x s block in *Y, in place over the two buffers (`passes` = 2: CholeskyQR2; 1: the intermediate
// blocks of the power iteration, with min pivot / max diagonal of the Gram matrix in pivs[slot + 1]).  On return *Y
// holds Q and *other is free.
// (Round 5 built and measured dropping the intermediate passes behind a device-side guard -- -0.2 ms at cfg-2, a subspace
//  error of 4e-11 that one hypersensitive pool of the fuzz slice turns into other indices: profiles/r05_ab_nystrom_skip.txt.
//  Round 6 took that opt-in path and its guarded launches out of the library; the passes stay.)
// ... of a block whose rank s is beyond one workgroup's Cholesky panel (sober_chol_max_n() < s <= 2 x that; round 6): the
// columns in two halves, block Gram-Schmidt with the same three kernels --
//     Q1 = Y1 R11^-1,   W = Q1^T Y2,   Y2 -= Q1 W,   Q2 = Y2 R22^-1
// -- which IS the QR factorisation of [Y1 | Y2] (unique: positive diagonals), so the same Q to rounding; the verdicts of the two
// factorisations are merged into the pass's slot (first failure, smallest pivot, smallest pivot ratio).
static int nx_orth_halves(const sober_nystrom_job* j, double* Y, double* O, int32_t* info, double* piv, double* ratio, void* stream) {
    const int M = j->M, s = j->s, s1 = (s + 1) / 2, s2 = s - s1;
    double* G1 = j->Gm;                                       // s1 x s1 (then s2 x s2), compact
    double* W = j->Gm + (size_t)s1 * s1;                      // s1 x s2: s1 s1 + s1 s2 = s1 s <= s s doubles in all
    double* tail = j->xinv + (size_t)((s + 31) / 32 - 1) * 1024;          // (a half uses fewer 32-blocks of xinv than that)
    double *piv2 = tail, *ratio2 = tail + 1;
    int32_t* info2 = (int32_t*)(tail + 2);
    NX_TRY(sober_dgemm(1, 0, s1, s1, M, 1.0, Y, s, Y, s, 0.0, G1, s1, stream));
    NX_TRY(sober_cholesky_inv_ratio(G1, s1, s1, 0.0, info, piv, j->xinv, ratio, stream));
    NX_TRY(sober_trsm_blocks(Y, M, s1, s, G1, s1, j->xinv, O, s, stream));
    NX_TRY(sober_dgemm(1, 0, s1, s2, M, 1.0, O, s, Y + s1, s, 0.0, W, s2, stream));
    NX_TRY(sober_dgemm(0, 0, M, s2, s1, -1.0, O, s, W, s2, 1.0, Y + s1, s, stream));
    NX_TRY(sober_dgemm(1, 0, s2, s2, M, 1.0, Y + s1, s, Y + s1, s, 0.0, G1, s2, stream));
    NX_TRY(sober_cholesky_inv_ratio(G1, s2, s2, 0.0, info2, piv2, j->xinv, ratio ? ratio2 : nullptr, stream));
    NX_TRY(sober_trsm_blocks(Y + s1, M, s2, s, G1, s2, j->xinv, O + s1, s, stream));
    return sober::orth_merge(info, piv, ratio, info2, piv2, ratio ? ratio2 : nullptr, stream);
}

static int nx_orth(const sober_nystrom_job* j, double** Y, double** other, int32_t* infos, double* pivs, int slot,
                   int passes, void* stream) {
    const int M = j->M, s = j->s;
    for (int it = 0; it < passes; ++it) {
        if (s > sober_chol_max_n()) {
            NX_TRY(nx_orth_halves(j, *Y, *other, infos + slot + it, pivs + slot + it, passes == 1 ? pivs + slot + 1 : nullptr, stream));
        } else {
            NX_TRY(sober_dgemm(1, 0, s, s, M, 1.0, *Y, s, *Y, s, 0.0, j->Gm, s, stream));                         // Y^T Y
            NX_TRY(sober_cholesky_inv_ratio(j->Gm, s, s, 0.0, infos + slot + it, pivs + slot + it, j->xinv,
                                            passes == 1 ? pivs + slot + 1 : nullptr, stream));
            NX_TRY(sober_trsm_blocks(*Y, M, s, s, j->Gm, s, j->xinv, *other, s, stream));                         // Q = Y R^-1
        }
        double* t = *Y; *Y = *other; *other = t;
    }
    return 0;
}

extern "C" int sober_nystrom_basis(const sober_nystrom_job* j, int phase, void* stream) {
    if (phase < 0 || phase > 2) return SOBER_E_ARG;
    if (!j || !j->G || !j->shifts || (!j->R && phase != 1) || !j->C || !j->chol_work || !j->Y[0] || !j->Y[1] || !j->Gm ||
        !j->xinv || !j->flags_block || !j->h_flags_block || !j->Ut)
        return SOBER_E_ARG;
    const int M = j->M, s = j->s, n_r = j->n_rungs, niter = j->niter;
    if (M <= 0 || s <= 0 || s >= M || s > 2 * sober_chol_max_n() || n_r <= 0 || n_r > 64 || niter < 0 || niter > 8) return SOBER_E_ARG;
    if (M > sober_nystrom_max_n()) return SOBER_E_DIM;
    if (j->flags_bytes < sober_nystrom_flags_bytes(n_r, niter)) return SOBER_E_WS;
    hipStream_t st = (hipStream_t)stream;
    const int n_orth2 = 2 * (1 + 2 * niter);
    // the flag block: pivots[n_r + 1] f64 | pivs_rf[n_orth2] f64 | flags[2 + n_r] i32 | infos_rf[n_orth2] i32
    double* pivots = (double*)j->flags_block;
    double* pivs_rf = pivots + n_r + 1;
    int32_t* flags = (int32_t*)(pivs_rf + n_orth2);
    int32_t* infos_rf = flags + 2 + n_r;
    if (phase != 2) {
        // (the multi-workgroup probe's own set-up rides in the launch that zeroes the flag block)
        const bool mc = M <= sober_chol_max_n() && j->probe_mc && j->probe_ws && (j->flags_bytes & 3) == 0 &&
                        j->probe_ws_bytes >= sober_cholesky_probe_mc_ws_bytes(M, n_r);
        if (mc) NX_TRY(sober::nystrom_flags_init(j->flags_block, j->flags_bytes, j->probe_ws, sober::probe_mc_flag_bytes(n_r),
                                                 flags + 2, n_r, stream));
        else NX_HIP(hipMemsetAsync(j->flags_block, 0, (size_t)j->flags_bytes, st));
        // ---- make_cov_psd: |cov| + symmetry flag + largest diagonal entry, every rung of the jitter ladder probed at once,
        //      the first positive definite rung (or the diagonal fallback) applied with the reference's own additions
        NX_TRY(sober_abs_sym_dmax(j->G, M, M, j->C, M, flags, pivots + n_r, stream));
        if (M > sober_chol_max_n()) {               // beyond one workgroup's LDS panel: panel by panel, two launches each
            if (!j->probe_ws) return SOBER_E_ARG;
            NX_TRY(sober_cholesky_probe_batched(j->C, M, M, j->shifts, n_r, j->chol_work, flags + 2, pivots, j->probe_ws,
                                                j->probe_ws_bytes, stream));
        } else if (j->probe_mc) {
            if (!j->probe_ws) return SOBER_E_ARG;
            NX_TRY(sober::cholesky_probe_mc(j->C, M, M, j->shifts, n_r, j->chol_work, flags + 2, pivots, j->probe_ws,
                                            j->probe_ws_bytes, !mc, stream));
        } else {
            NX_TRY(sober_cholesky_probe_piv(j->C, M, M, j->shifts, n_r, j->chol_work, flags + 2, pivots, stream));
        }
        NX_TRY(sober_jitter_ladder_auto(j->C, M, M, flags + 2, n_r, flags + 1, stream));
    }
    if (phase == 1) return 0;       // (the caller's host work -- stepping the generator for R -- overlaps with the probes)
    // ---- the range finder of torch.svd_lowrank (Halko et al. Alg. 4.4, torch/_lowrank.py:64-79): only range(Q) of the
    //      LAST block enters the result, so the intermediate blocks take one CholeskyQR pass, the last one two
    const int last = 2 * niter;
    double *Q = j->Y[0], *free_buf = j->Y[1];
    NX_TRY(sober_dgemm(0, 0, M, s, M, 1.0, j->C, M, j->R, s, 0.0, Q, s, stream));                         // A R
    NX_TRY(nx_orth(j, &Q, &free_buf, infos_rf, pivs_rf, 0, last > 0 ? 1 : 2, stream));
    int slot = 2, k = 0;
    for (int it = 0; it < niter; ++it) {
        for (int half = 0; half < 2; ++half) {
            ++k;
            NX_TRY(sober_dgemm(half == 0 ? 1 : 0, 0, M, s, M, 1.0, j->C, M, Q, s, 0.0, free_buf, s, stream));   // A^H Q, then A Q
            double* t = Q; Q = free_buf; free_buf = t;                                                 // (Q is consumed)
            NX_TRY(nx_orth(j, &Q, &free_buf, infos_rf, pivs_rf, slot, (half == 1 && k == last) ? 2 : 1, stream));
            slot += 2;
        }
    }
    // ---- U = Q^T (s x M) and P = [U diag(mean), -(U diag(mean)) T]: the Nystrom test functions with the posterior
    //      correction folded in (the transposition and P's left block in one launch)
    if (j->P && (j->T ? j->n_obs <= 0 : false)) return SOBER_E_ARG;
    const int ldp = M + (j->T ? j->n_obs : 0);
    NX_TRY(sober::transpose_projection(Q, s, M, j->mean_nys, j->Ut, j->P, ldp, stream));
    if (j->P && j->T) NX_TRY(sober_dgemm(0, 0, s, j->n_obs, M, -1.0, j->P, ldp, j->T, j->n_obs, 0.0, j->P + M, ldp, stream));
    NX_HIP(hipMemcpyAsync(j->h_flags_block, j->flags_block, (size_t)sober_nystrom_flags_bytes(n_r, niter),
                          hipMemcpyDeviceToHost, st));
    return 0;
}

// ---- the ROW TABLE's side of a step's plan behind one call (continuous kernels) -----------------------------------------
// rows = [X_nys; X_obs] / lengthscale (one launch into one table: no concatenated copy), Kall = k(rows, X_nys)
// ((M + n_obs) x M), W = S S^T (SOBER/_gp.py:277), T = K(X_nys, X_obs) W (M x n_obs; SOBER/_gp.py:293,295) and the Gram
// matrix of SOBER/_rchq.py:35, G = K(X_nys, X_nys) - T K(X_obs, X_nys), in the reference's association order.  With
// n_obs = 0 (mode "kernel"): rows and G = k(X_nys, X_nys) only.  The same launches the host language used to issue one
// by one (~130 us of a host-paced stream in front of the Nystrom chain), same order, same bits.
extern "C" int sober_plan_rows(int kind, const double* X_nys, int M, int64_t ld_nys, const double* X_obs, int n_obs,
                               int64_t ld_obs, int d, const double* lengthscale, int ls_len, double outputscale,
                               const double* S_cache, int ld_s, double* rows, int dt, double* Kall, double* W, double* T,
                               double* G, void* stream) {
    if (!X_nys || !lengthscale || !rows || !G || M <= 0 || d <= 0 || n_obs < 0 || dt < d) return SOBER_E_ARG;
    if (kind != SOBER_KIND_RBF && kind != SOBER_KIND_MATERN52) return SOBER_E_ARG;
    if (n_obs == 0) {
        NX_TRY(sober_scale_points(X_nys, M, d, ld_nys, lengthscale, ls_len, rows, dt, stream));
        return sober_pairwise(kind, rows, nullptr, M, rows, nullptr, nullptr, M, dt, outputscale, G, M, stream);
    }
    if (!X_obs || !S_cache || !Kall || !W || !T || ld_s < n_obs) return SOBER_E_ARG;
    NX_TRY(sober::scale_points2(X_nys, M, ld_nys, X_obs, n_obs, ld_obs, d, lengthscale, ls_len, rows, dt, stream));
    NX_TRY(sober_pairwise(kind, rows, nullptr, (int64_t)M + n_obs, rows, nullptr, nullptr, M, dt, outputscale, Kall, M, stream));
    NX_TRY(sober_dgemm(0, 1, n_obs, n_obs, n_obs, 1.0, S_cache, ld_s, S_cache, ld_s, 0.0, W, n_obs, stream));
    const double* KXn = Kall + (size_t)M * M;                              // k(X_obs, X_nys), n_obs x M
    NX_TRY(sober_dgemm(1, 0, M, n_obs, n_obs, 1.0, KXn, M, W, n_obs, 0.0, T, n_obs, stream));
    NX_HIP(hipMemcpyAsync(G, Kall, sizeof(double) * (size_t)M * M, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return sober_dgemm(0, 0, M, M, n_obs, -1.0, T, n_obs, KXn, M, 1.0, G, M, stream);
}
