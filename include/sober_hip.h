/*
 * sober_hip.h -- C ABI of libsober_hip.so, the MI355X (gfx950) implementation of
 * SOBER's kernel-recombination hot path.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into caller-owned memory unless its name
 *     starts with h_; the library never allocates or frees caller data
 *     (workspaces are passed in, their sizes come from the *_ws_bytes queries);
 *   - work is enqueued on the caller's stream (`stream` is a hipStream_t passed as
 *     void*; NULL = the default stream); no entry point synchronises;
 *   - return value: 0 = ok, < 0 = argument error (SOBER_E_*), > 0 = hipError_t;
 *   - FP64 everywhere (the reference runs torch.double, SOBER/_settings.py:8);
 *     candidate / position indices are int32 on the device (N < 2^31);
 *   - "scaled points" are rows of DT doubles: x[j] / lengthscale[j] for j < d, zero
 *     padding for d <= j < DT (DT = sober_padded_dim(d)); for the Tanimoto kernel a
 *     point is DT/... see sober_pack_bits.
 *
 * Each entry cites the reference code it replaces (paths relative to the
 * reference checkout, ma921/SOBER v2.0.2).
 */
#ifndef SOBER_HIP_H
#define SOBER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SOBER_ABI_VERSION 7

/* kernel families: model.covar_module.forward behind SOBER/_gp.py:292-294 */
#define SOBER_KIND_RBF       0   /* outputscale * exp(-0.5 * |x/l - y/l|^2)                      */
#define SOBER_KIND_MATERN52  1   /* outputscale * (1 + sqrt5 r + 5/3 r^2) exp(-sqrt5 r)          */
#define SOBER_KIND_TANIMOTO  2   /* outputscale * (x.y + 1e-6) / (1e-6 + |x|^2 + |y|^2 - x.y),
                                    SOBER/_drug_modelling.py:15-25,37 (bit-packed 0/1 inputs)    */

#define SOBER_E_ARG      -1      /* null pointer / non-positive size / bad enum                  */
#define SOBER_E_DIM      -2      /* dimension not supported by the compiled tile set             */
#define SOBER_E_WS       -3      /* workspace too small                                           */
#define SOBER_E_EXCHANGE -5      /* a multi-workgroup kernel gave up waiting for a partner (bounded spins); no result */

int sober_abi_version(void);
/* 1 = a diagnostic build (in-kernel time stamps, -DSOBER_DIAG_BUILD: `make stamps`); sober_amd refuses to load one
 * unless SOBER_ALLOW_DIAG_LIB=1.  `make all` gives 0. */
int sober_diag_build(void);
/* The A/B and test switches of the environment that the LIBRARY reads (SOBER_LEVEL_TWO_LAUNCHES, SOBER_TANI_NO_QUEUE,
 * SOBER_CAR_FORCE_GIVEUP, SOBER_CAR_UNFUSED, SOBER_CAR_EXACT_RATIO, SOBER_LEVEL_NO_CLASSES) are read ONCE, when it is loaded; a
 * process that changes one afterwards (the tests do) calls this to have them read again.  Returns 0.  (The Python host side has
 * switches of its own, read where they act: SOBER_NO_QUEUE at HipOps construction, SOBER_PREDICT_MATERIALISED, SOBER_PREDICT_FROM_W and
 * SOBER_NYSTROM_DEBUG per call, SOBER_HIP_LIB / SOBER_ALLOW_DIAG_LIB at load -- INTEGRATION.md lists them.)           */
int sober_reload_switches(void);
/* sizeof(sober_level_job) / sizeof(sober_nystrom_job) as the library was built: a binding that lays the structs out itself
 * (ctypes, cgo, JNA ...) checks its own size against these before the first call.                                   */
int sober_level_job_size(void);
int sober_nystrom_job_size(void);

/* DT for a d-dimensional continuous kernel (multiple of 4, >= d), or -2 if d > SOBER max.       */
int sober_padded_dim(int d);
/* number of 64-bit words per point for a d-bit Tanimoto fingerprint.                             */
int sober_bit_words(int d);

/* The random test matrix of torch.svd_lowrank (SOBER/_rchq.py:37; torch.randn(M, q, dtype=float64) on the CPU
 * generator) in two halves.  sober_mt19937_uniform53 (HOST function, no GPU work) steps the generator state `state`
 * (the bytes of torch.get_rng_state(), updated in place: afterwards it is what torch.randn would have left) and
 * writes the uniforms the normal fill consumes: numel of them, plus 16 when numel % 16 != 0 (out holds numel + 16
 * doubles).  sober_box_muller turns them into numel normals on the device, paired like ATen's fill.            */
int sober_mt19937_uniform53(uint8_t* state, int64_t state_bytes, int64_t numel, double* out);
int sober_box_muller(const double* u, int64_t numel, double* out, void* stream);

/* x[i, 0:d] / lengthscale -> out[i, 0:dt] (zero padded).  ls_len is 1 (isotropic) or d (ARD).
 * Replaces the `x1.div(lengthscale)` prologue of gpytorch's RBF/Matern forward that
 * SOBER/_gp.py:292-294 calls on every kernel evaluation.                                         */
int sober_scale_points(const double* X, int64_t n, int d, int64_t ldx,
                       const double* lengthscale, int ls_len,
                       double* out, int dt, void* stream);

/* 0/1 FP64 fingerprints -> bit-packed words (nwords per row, little-endian bit j of word j/64) and
 * per-row popcount (|x|^2) as double.  *bad_flag (device int32, zero it first) is set to 1 when an
 * input is neither 0.0 nor 1.0.  Replaces the FP64 storage consumed by
 * SOBER/_drug_modelling.py:20-22.                                                                */
/* the same for the rows idx[0:n] of X only: out[i, 0:dt] = X[idx[i], 0:d] / lengthscale; gather_src != NULL: also
 * gather_out[i] = gather_src[idx[i]] (the live positions' weights, SOBER/_rchq.py:84, without a launch of their own)   */
int sober_scale_points_idx(const double* X, const int32_t* idx, int64_t n, int d, int64_t ldx, const double* lengthscale,
                           int ls_len, double* out, int dt, const double* gather_src, double* gather_out, void* stream);
int sober_pack_bits(const double* X, int64_t n, int d, int64_t ldx,
                    uint64_t* words, int nwords, double* norms, int32_t* bad_flag, void* stream);

/* out[i*ldo + j] = k(a_i, b_{idx ? idx[j] : j}),  i < m, j < n.
 * a / b are scaled points (continuous kinds; dt doubles per row) or packed words (Tanimoto; dt =
 * words per row, a_norm / b_norm = popcounts).  The materialised kernel matrix of
 * `model.covar_module.forward(x, y)`, SOBER/_gp.py:292-294 and SOBER/_kernel.py:28.              */
int sober_pairwise(int kind, const void* a, const double* a_norm, int64_t m,
                   const void* b, const double* b_norm, const int32_t* idx, int64_t n,
                   int dt, double outputscale, double* out, int64_t ldo, void* stream);

/* out[j] = c0 + sum_i k(a_i, b_j) * v[i]   (posterior mean over a pool, j < n): the
 * `predict_mean` of SOBER/_gp.py:240-253 needed by weighted_covariance, SOBER/_kernel.py:40-41.  */
int sober_kernel_matvec(int kind, const void* a, const double* a_norm, const double* v, int64_t m,
                        const void* b, const double* b_norm, int64_t n,
                        int dt, double outputscale, double c0, double* out, void* stream);

/* The hot kernel (K1-K3 of SURVEY.md 2.1): fused "kernel evaluation x weight -> set sums".
 * For GLOBAL list positions p in [pos0, pos0 + count) (candidate c = idx[p - pos0], set s = p mod S,
 * element p div S; pos0 = 0 on one GPU, the start of this rank's contiguous position range when the
 * list is sharded):
 *     partG  [chunk][row][col0 + s] += k(rows[row], cand[c]) * mu[c] * (wmul ? wmul[c] : 1)
 *     partTot[chunk][col0 + s]      += mu[c]                    for p < tot_limit only
 * with the element range split into n_chunks chunks (deterministic partial sums, no atomics).
 * Never materialises the (E, M, S) tensor of SOBER/_rchq.py:122-126; running it over ALL live
 * positions (count = R) reproduces the leftover double count of :128-136 (quirk Q1), and
 * tot_limit = E*S keeps `tot_weights` (:152) free of it.  A second call with S = 1 over the
 * leftover positions yields the column added to the last set at :153-164.
 *   rows     n_rows scaled points (X_nys stacked on X_obs: the posterior correction of
 *            SOBER/_gp.py:295 is applied after the sum, it is linear)
 *   partG    n_chunks * n_rows * ldg doubles, partTot n_chunks * ldg doubles (may be NULL)     */
int sober_level_reduce(int kind, const void* rows, const double* rows_norm, int n_rows,
                       const void* cand, const double* cand_norm, int dt,
                       const int32_t* idx, int64_t pos0, int64_t count, int S,
                       const double* mu, const double* wmul, double outputscale,
                       int n_chunks, double* partG, int ldg, int col0,
                       double* partTot, int64_t tot_limit, void* stream);
/* Matrix-core variant of sober_level_reduce for the continuous kernels: rows / cand are AUGMENTED
 * points (sober_augment_points: centred, scaled, with -|x|^2/2 and 1 in the two slots after the d
 * coordinates, DA = sober_aug_dim(d) doubles per row), so that one v_mfma_f64_16x16x4 chain yields
 * -|x~ - y~|^2 / 2 for a 16 x 16 tile of (row, candidate) pairs; the vector pipe only evaluates the
 * exponential (table-driven FP64 exp) and accumulates.  Same outputs and chunking as
 * sober_level_reduce.                                                                            */
int sober_aug_dim(int d);
/* The plan's two augmented tables in two launches (round 6): center[0:d] = the column means of X_nys (fixed summation order),
 * then rows_aug ((M + n_obs) x da, side 0: [X_nys; X_obs], read where they lie -- no concatenated copy) and cand_aug (N x da,
 * side 1) like sober_augment_points.  n_obs = 0 / X_obs = NULL: mode "kernel".  d <= 30.                              */
int sober_augment_plan(const double* X_nys, int64_t M, int64_t ld_nys, const double* X_obs, int64_t n_obs, int64_t ld_obs,
                       const double* X_cand, int64_t N, int64_t ld_cand, int d, const double* lengthscale, int ls_len,
                       double* center, double* rows_aug, double* cand_aug, int da, void* stream);
int sober_augment_points(const double* X, int64_t n, int d, int64_t ldx, const double* lengthscale,
                         int ls_len, const double* center, int side, double* out, int da, void* stream);
int sober_level_reduce_mfma(int kind, const double* rows, int n_rows, const double* cand, int da,
                            const int32_t* idx, int64_t pos0, int64_t count, int S,
                            const double* mu, const double* wmul, double outputscale,
                            int n_chunks, double* partG, int ldg, int col0,
                            double* partTot, int64_t tot_limit, void* stream);

/* chunk count the library would pick for a level of `count` positions in S sets.                 */
/* The live list idx_story = arange(N)[mu != 0] (SOBER/_rchq.py:63-65) as an ordered stream compaction whose count
 * stays on the device (*count_out, int64): no synchronisation to size the output, unlike torch.nonzero.  idx_out holds
 * N entries; ws from sober_nonzero_ws_bytes(N), ZERO before its first use -- the call (one launch: ticket-ordered tiles with a
 * decoupled look-back) leaves it zero again; one call at a time per workspace (ABI 7: the workspace used to be arbitrary).   */
int64_t sober_nonzero_ws_bytes(int64_t N);
int sober_nonzero_i32(const double* mu, int64_t N, int32_t* idx_out, int64_t* count_out, void* ws, int64_t ws_bytes,
                      void* stream);

int sober_level_chunks(int n_rows, int64_t pos0, int64_t count, int S);
/* The same for sober_level_reduce_mfma (its workgroups are waves of one 64-row x 16-set tile that sum their element
 * ranges on chip: 1-2 partial sums per tile), and the largest value it can take up to e_total_ub elements per set (what
 * a launch sized ahead of time, sober_level_reduce_mfma_queued, is given). */
int sober_level_parts_mfma(int n_rows, int64_t pos0, int64_t count, int S);
int sober_level_parts_mfma_cap(int n_rows, int64_t e_total_ub, int S);

/* G[row*ldo + s] = sum_chunk partG[chunk][row][s]  (s < S).  When extraG != NULL the leftover
 * partials extraG[xchunk][row][0:n_xcols] (and extraTot[xchunk][0:n_xcols]) -- a level_reduce run
 * over the leftover positions alone -- are summed and folded into set S-1
 * (SOBER/_rchq.py:160-164).                                                                       */
int sober_sum_partials(const double* partG, const double* partTot, int n_chunks, int n_rows,
                       int ldg, int S, const double* extraG, const double* extraTot,
                       int n_xchunks, int n_xcols, double* G, int ldo, double* tot, void* stream);

/* Row-major FP64 GEMM on the matrix cores (v_mfma_f64_16x16x4_f64):
 * C[m,n] = alpha * op(A)[m,k] * op(B)[k,n] + beta * C.  Used for W = S S^T (SOBER/_gp.py:277),
 * T = KxX W and the posterior correction (SOBER/_gp.py:295), U @ X_for_nys (SOBER/_rchq.py:148). */
int sober_dgemm(int transa, int transb, int m, int n, int k, double alpha,
                const double* A, int lda, const double* B, int ldb,
                double beta, double* C, int ldc, void* stream);

/* X_tmp[s*n + i] = Xtr[i*ldx + s] / tot[s]   (SOBER/_rchq.py:151,166); tot may be NULL (no
 * division: the final direct level, :78).                                                        */
int sober_barycentres(const double* Xtr, int ldx, int n, int S, const double* tot,
                      double* X_tmp, void* stream);
/* Projection and barycentres in one launch: Ct[j][i] = (A B)[i][j] / coldiv[j] (A: m x k, B: k x n, row-major; coldiv may
 * be null) -- bit-identical to sober_dgemm followed by sober_barycentres. */
int sober_dgemm_coldiv_t(int m, int n, int k, const double* A, int lda, const double* B, int ldb, const double* coldiv,
                         double* Ct, int ldct, void* stream);

/* K7 (SOBER/_rchq.py:198-221): rescale the kept sets' weights, zero the cancelled ones and write
 * the compacted survivor list (element-major, kept sets in ascending order, leftovers appended iff
 * the last set survived).  keep_rank[s] = rank of set s in idx_star or -1.  New global length =
 * E * n_keep + (keep_rank[S-1] >= 0 ? R - E*S : 0).  idx_cur holds the `count` positions starting at
 * global position pos0; survivors are written to idx_new[new_global_position - new_pos0].         */
int sober_level_update(const int32_t* idx_cur, int64_t pos0, int64_t count, int S, int64_t E,
                       const int32_t* keep_rank, const double* w_star, const double* tot,
                       int n_keep, double* mu, int32_t* idx_new, int64_t new_pos0, void* stream);

/* Final direct level write-back (SOBER/_rchq.py:108-110): mu[:] = 0 is the caller's memset; this
 * scatters mu[idx_cur[sel[k]]] = w[k] and writes out_idx[k] = idx_cur[sel[k]] (int64).           */
int sober_scatter_weights(const int32_t* idx_cur, const int32_t* sel, const double* w, int n_sel,
                          double* mu, int64_t* out_idx, void* stream);

/* out[t] = src[idx[t]], t < n.                                                                       */
int sober_gather_f64(const double* src, const int32_t* idx, int64_t n, double* out, void* stream);
/* The same write-back driven by the Caratheodory step's keep_rank[0:n] on the device (no host list): position t
 * with rank k = keep_rank[t] >= 0 gives mu[idx[t]] = w_star[k], out_idx[k] = idx[t] + row_offset, out_w[k] = w_star[k]. */
int sober_final_scatter(const int32_t* idx, int n, const int32_t* keep_rank, const double* w_star,
                        int64_t row_offset, double* mu, int64_t* out_idx, double* out_w, void* stream);
/* mu[0:N] = 0 (SOBER/_rchq.py:109) followed by that write-back, both skipped when the device word *n_keep is negative
 * (the Caratheodory step in front reported no result: the weights stay as they were for the caller's other route). */
int sober_final_commit(const int32_t* idx, int n, const int32_t* keep_rank, const double* w_star, const int32_t* n_keep,
                       int64_t row_offset, double* mu, int64_t N, int64_t* out_idx, double* out_w, void* stream);

/* idx[p] = p-th index with mu != 0 is the caller's job (torch.nonzero); helper: int64 -> int32.   */
int sober_i64_to_i32(const int64_t* in, int64_t n, int32_t* out, void* stream);

/* HOST function (h_ pointers are host memory): the pivot loop of Tchernychova_Lyons_CAR,
 * SOBER/_rchq.py:237-266.  h_Phi is the null-space basis, N x m row-major (ld = m), destroyed;
 * h_mu (N) is updated in place.  Same IEEE operation order as the reference's tensor expressions
 * (no FMA contraction).  Returns the number of pivots performed (< m when quirk Q6 fires).        */
int sober_car_pivot_host(double* h_Phi, int N, int m, double* h_mu);
/* Same loop with one division per column (q_c = Phi[idx,c]/Phi[idx,0]) instead of one per element: last-bit
 * differences, ~6x faster; used for N > 200 (batch > 100), where the host route runs.               */
int sober_car_pivot_host_fast(double* h_Phi, int N, int m, double* h_mu);

/* DEVICE Caratheodory step (K5 + K6): the whole Tchernychova_Lyons_CAR of SOBER/_rchq.py:224-270 in
 * one persistent workgroup.  X (N, m-1) row-major barycentres (ld = ldx), mu_in (N) set masses.
 * The null-space basis is rebuilt from the right Householder reflectors of the Golub-Kahan
 * bidiagonalisation of [1 | X]^T -- the basis LAPACK/MKL gesdd returns in Vh[m:, :] (:231-234) --
 * followed by the N-m pivots of :237-266.  Outputs: keep_rank[N] (rank of each surviving row in
 * idx_star, -1 if cancelled), w_star[0:n_keep], *n_keep, mu_out[N]; phi_out (N x (N-m), may be NULL)
 * receives the null-space basis before the pivots (test hook).
 * Three implementations behind the same entry point (the third, csrc/car_big.hip, for everything beyond the second: below):
 * one compute unit (csrc/car.hip: N <= 208, m <= 112,
 * batch <= 100) and, beyond that, a multi-CU version (csrc/car_mc.hip: N <= 448, m <= 256, batch <= 224:
 * the matrix spread over ceil(N/52) workgroups with ONE cross-workgroup exchange per bidiagonalisation
 * step, the pivots as a streaming pipeline of 32 waves).  n_keep = -1 reports a give-up of the multi-CU
 * exchange (bounded spins).
 * sober_car_supported(N, m) = 1 iff one of the three covers the size (N <= 2048).                 */
int sober_car_supported(int N, int m);
int64_t sober_car_ws_bytes(int N, int m);
int sober_car_device(const double* X, int ldx, int N, int m, const double* mu_in,
                     int32_t* keep_rank, double* w_star, int32_t* n_keep, double* mu_out,
                     double* phi_out, void* ws, int64_t ws_bytes, void* stream);
/* out[i] = v_rcp_f64(x[i]): the hardware reciprocal seed exactly as the pivot kernels' screened ratio test
 * (SOBER/_rchq.py:240-247) uses it -- test hook that pins its relative error (< 2^-22 per band step, csrc/car.hip). */
int sober_probe_rcp(const double* x, double* out, int64_t n, void* stream);

/* The same step with the choice of launches in the caller's hand (SOBER/_rchq.py:224-270 never fails, so a give-up
 * must be recoverable):
 *   SOBER_CAR_DEFAULT  what sober_car_device runs: launches whose workgroups wait for partner workgroups (the fused
 *                      bidiagonalisation + Phi of csrc/car.hip, every kernel of csrc/car_mc.hip).  All waits are
 *                      bounded; a give-up (partners not co-resident: a shared or partitioned device) reports
 *                      *n_keep = -1 and leaves keep_rank / w_star unwritten.
 *   SOBER_CAR_SAFE     only launches in which no workgroup depends on another one (stand-alone bidiagonalisation,
 *                      one wave per column of Phi, the one-workgroup pivot stream): cannot give up; exists for the
 *                      one-CU sizes and, through csrc/car_big.hip (a launch per dependency), for every size up to
 *                      N = 2048 (sober_car_safe_supported); SOBER_E_DIM beyond -- the host route is next.
 * The level loops below redo a level that came back with n_keep = -1 in SOBER_CAR_SAFE mode, set job->car_mode so
 * that the rest of the run (and, through the caller, the following steps) stay there, and only return
 * SOBER_E_EXCHANGE when that mode does not cover the size.
 * sober_car_giveup_forced() = 1 while the switch SOBER_CAR_FORCE_GIVEUP is on (environment, read at load time and by
 * sober_reload_switches): the test switch that makes every SOBER_CAR_DEFAULT launch give up at once.              */
#define SOBER_CAR_DEFAULT 0
#define SOBER_CAR_SAFE 1
int sober_car_safe_supported(int N, int m);
int sober_car_giveup_forced(void);
int sober_car_device_ex(const double* X, int ldx, int N, int m, const double* mu_in,
                        int32_t* keep_rank, double* w_star, int32_t* n_keep, double* mu_out,
                        double* phi_out, void* ws, int64_t ws_bytes, int mode, void* stream);
/* The extra elimination of the acquisition-guided branch (SOBER/_rchq.py:87-106, :177-196) after a Caratheodory step
 * with the objective as one more test function: the n1 surviving weights w1 (in rank order; rank1[0:Nsets] maps a
 * set to its rank or -1) move along the null vector phi of [X_p; 1] -- the one-column phi_out of sober_car_device on
 * the n1 survivors -- oriented so that sum w * objp does not decrease, until one more reaches zero.
 * Out: keep_rank[0:Nsets] (new ranks, -1 = cancelled), w_star[0:n_keep], *n_keep.                               */
/* The objective's row of the level's set sums (SOBER/_rchq.py:138-146, :157-163): out[s] = sum of obj[c] mu[c] over the list
 * positions p in [pos0, pos0 + count) with p mod S = s (c = idx[p - pos0]), the leftovers p >= E S also into set S - 1 (Q1).   */
int sober_obj_set_sums(const double* obj, const double* mu, const int32_t* idx, int64_t pos0, int64_t count, int S, int64_t E,
                       double* out, void* stream);
int sober_second_elimination(const double* phi, const double* objp, const double* w1, const int32_t* rank1, int n1,
                             int Nsets, int32_t* keep_rank, double* w_star, int32_t* n_keep, void* stream);
/* Round 6: that null vector in ONE launch (csrc/null_vector.hip): Gauss-Jordan elimination with partial pivoting on
 * A2 = [features of the n1 survivors; 1] (nfun + 1 rows, n1 = nfun + 2 points) read straight from the level's X (Nsets x ldx,
 * the first nfun columns: the objective column is not part of A2) through rank1 -- instead of a second run of the Caratheodory
 * kernels on the survivors.  null_row[0:Nsets]: the vector by SET, zero on cancelled sets; *status: 0, -1 / -2 when the first
 * step gave up / did not leave exactly n1 sets (*n_keep1 read on the device), -3 when A2 has a zero pivot row (null space not
 * a line: the reference's answer is then whatever its SVD returns -- the caller's host route).  nfun <= 111 (one workgroup).
 * sober_second_elimination_rows: the elimination with null vector and objective BY SET and the first step's verdict read on
 * the device: *n_keep = -2 (nothing written) unless *n_keep1 == n1, -1 when the first step gave up.                  */
int sober_null_vector_supported(int nfun);
int sober_null_vector(const double* X, int ldx, int Nsets, int nfun, const int32_t* rank1, const int32_t* n_keep1, int n1,
                      double* null_row, int32_t* status, void* stream);
int sober_second_elimination_rows(const double* null_row, const double* obj_row, const double* w1, const int32_t* rank1,
                                  const int32_t* n_keep1, const int32_t* status, int n1, int Nsets, int32_t* keep_rank,
                                  double* w_star, int32_t* n_keep, void* stream);
/* (status: sober_null_vector's verdict word read on the device -- != 0 gives *n_keep = -2 --, or NULL.)
 * The objective's set sums of a QUEUED level (sober_level_loop_obj below): the level's size from *dR (R_known >= 0: by value),
 * the barycentre column out[s] / tot[s] written to xcol[s * ldx] and ocol[s]; nothing is written once the chain has stopped. */
int sober_obj_set_sums_queued(const double* obj, const double* mu, const int32_t* idx, int S, const int64_t* dR, int64_t R_known,
                              const double* tot, double* xcol, int ldx, double* ocol, void* stream);

/* The multi-CU implementation by itself (any size it covers, also the small ones: test and timing hook). */
int sober_car_mc_supported(int N, int m);
int64_t sober_car_mc_ws_bytes(int N, int m);
int sober_car_mc_device(const double* X, int ldx, int N, int m, const double* mu_in,
                        int32_t* keep_rank, double* w_star, int32_t* n_keep, double* mu_out,
                        double* phi_out, void* ws, int64_t ws_bytes, void* stream);
/* The memory-resident implementation (csrc/car_big.hip, round 6): any N <= 2048 points with m < N test functions (batch <= 1023),
 * the matrix in L2 / HBM, one launch per dependency -- k_big_right / k_big_left per bidiagonalisation step, one wave per null
 * vector for Phi, the pivots in panels of eight (a launch per panel) -- so no workgroup ever waits for another one: it is what
 * sober_car_device runs beyond batch 224 (host LAPACK + C++ pivots before) and the SOBER_CAR_SAFE rung of the multi-CU sizes.
 * Same outputs as sober_car_device; never reports n_keep = -1.                                                        */
int sober_car_big_supported(int N, int m);
int64_t sober_car_big_ws_bytes(int N, int m);
int sober_car_big_device(const double* X, int ldx, int N, int m, const double* mu_in,
                         int32_t* keep_rank, double* w_star, int32_t* n_keep, double* mu_out,
                         double* phi_out, void* ws, int64_t ws_bytes, void* stream);
/* Self test of the gfx950 lane-swap reductions the multi-CU kernels rely on: in (64) ->
 * out[0:64] = sum over the wave, out[64:128] = sum over lanes l, l^16, l^32, l^48.                  */
int sober_mc_selftest(const double* in, double* out, void* stream);

/* Dense FP64 Cholesky in one persistent workgroup (n <= sober_chol_max_n()): the lower triangle of A
 * (n x n row-major, ld) is overwritten by L with (A + shift I) = L L^T; *info = 0 on success, j+1 when
 * the leading minor of order j+1 is not positive definite (LAPACK dpotrf convention); *min_pivot (may
 * be NULL) = smallest pivot.  Replaces the `torch.linalg.cholesky` probes of is_psd / make_cov_psd,
 * SOBER/_utils.py:117-157 (the "Nystrom Cholesky"), and is the core of the CholeskyQR used for the
 * range finder of torch.svd_lowrank (SOBER/_rchq.py:37).                                           */
int sober_chol_max_n(void);
/* The ladder's probes of make_cov_psd for sober_chol_max_n() < n <= sober_nystrom_max_n() (2048): every rung panel by
 * panel, two launches per 32 columns, the 32 x 32 diagonal blocks through the same code as sober_cholesky (same verdict
 * and minimum pivot).  work: n_shifts slabs of n x n doubles; xinv_ws: n_shifts x 1024 doubles.  info / min_pivot as
 * sober_cholesky_probe_piv.  sober_nystrom_basis takes this route by itself (job.probe_ws must be set).            */
int sober_nystrom_max_n(void);
int sober_cholesky_probe_batched(const double* src, int n, int ld_src, const double* shifts, int n_shifts,
                                 double* work, int32_t* info, double* min_pivot, void* xinv_ws, int64_t ws_bytes,
                                 void* stream);
int sober_cholesky(double* A, int n, int ld, double shift, int32_t* info, double* min_pivot, void* stream);
/* Same, and xinv (ceil(n/32) blocks of 32 x 32 doubles, row-major) receives the inverses of the 32 x 32 diagonal
 * blocks of L (identity-padded in the last block) -- the operands of sober_trsm_blocks.                       */
int sober_cholesky_inv(double* A, int n, int ld, double shift, int32_t* info, double* min_pivot, double* xinv,
                       void* stream);
/* The same, also leaving *ratio_out = smallest pivot / largest diagonal entry of the input (~cond^-2 of a Gram matrix:
 * what decides whether one CholeskyQR pass is enough) -- on the device, no reduction kernels around the call. */
int sober_cholesky_inv_ratio(double* A, int n, int ld, double shift, int32_t* info, double* min_pivot, double* xinv,
                             double* ratio_out, void* stream);
/* Q[r, 0:q] = Y[r, 0:q] L^-T (L lower triangular q x q, q <= sober_chol_max_n(): the Q factor of Y when L L^T = Y^T Y), blocked
 * on the matrix cores with the inverted diagonal blocks of sober_cholesky_inv: block-to-block dependency only.                                                          */
int sober_trsm_blocks(const double* Y, int64_t m, int q, int ldy, const double* L, int ldl, const double* Xinv,
                      double* Q, int ldq, void* stream);
/* All rungs of the jitter ladder in one launch: workgroup b factorises (src + shifts[b] I) in slab b of
 * `work` (n_shifts * n * n doubles) and sets info[b] (0 = positive definite).  src is not modified.     */
int sober_cholesky_probe(const double* src, int n, int ld_src, const double* shifts, int n_shifts,
                         double* work, int32_t* info, void* stream);
/* The same, and min_pivot[b] (may be NULL) = smallest pivot rung b met: the failing one (<= 0) when info[b] != 0.
 * is_psd (SOBER/_utils.py:117-129) asks LAPACK's Cholesky AND `linalg.eig >= 0`: a rung whose smallest pivot is
 * within rounding of zero is one where those and this kernel may disagree -- the caller then lets the host's
 * LAPACK decide (sober_amd/_ops_hip.py:nystrom_basis_device).                                                    */
int sober_cholesky_probe_piv(const double* src, int n, int ld_src, const double* shifts, int n_shifts,
                             double* work, int32_t* info, double* min_pivot, void* stream);

/* The same probes with 8 workgroups per rung (block rows dealt round-robin; the workgroups of a rung on one XCD):
 * 0.49 -> ~0.2 ms at n = 500.  n_shifts <= 16; ws of sober_cholesky_probe_mc_ws_bytes bytes (scratch).  info[r] = -7:
 * the rung's workgroups lost each other (bounded spins) and the rung has NO verdict -- probe again with
 * sober_cholesky_probe_piv.  Same info / min_pivot otherwise.                                                  */
int64_t sober_cholesky_probe_mc_ws_bytes(int n, int n_shifts);
int sober_cholesky_probe_mc(const double* src, int n, int ld_src, const double* shifts, int n_shifts,
                            double* work, int32_t* info, double* min_pivot, void* ws, int64_t ws_bytes,
                            void* stream);
/* out = sqrt(nan_to_num(C) * nan_to_num(C)^T) elementwise (quirk Q2, SOBER/_utils.py:143-144);
 * flag[0] |= 1 when C is not exactly symmetric (:127).  Zero flag first.                            */
int sober_abs_sym(const double* C, int n, int ld, double* out, int ldo, int32_t* flag, void* stream);
/* ... and max_i out[i][i] into *dmax (device double, ZERO it first; cov.diag().max() of the ladder's borderline test) */
int sober_abs_sym_dmax(const double* C, int n, int ld, double* out, int ldo, int32_t* flag, double* dmax,
                       void* stream);

/* k rungs of make_cov_psd's jitter ladder (SOBER/_utils.py:151-152) with the reference's sequence of
 * roundings: jitter = 1e-5; repeat k times { diag(A) += jitter; jitter *= 2 }.                        */
int sober_jitter_ladder(double* A, int n, int ld, int k, void* stream);
/* The same with k taken on the device from sober_cholesky_probe's info[0:n_rungs]: k = first rung that is
 * positive definite; if none is, all n_rungs additions are made and A is reduced to its diagonal
 * (`cov.diag().diag()`, :153-156); *k_out = k.                                                             */
int sober_jitter_ladder_auto(double* A, int n, int ld, const int32_t* info, int n_rungs, int32_t* k_out,
                             void* stream);

/* K1-K3 for bit-packed fingerprints on the INT8 matrix cores (csrc/level_reduce_tani.hip): same contract as
 * sober_level_reduce with kind = Tanimoto; dt = 64-bit words per fingerprint, 8 / 16 / 32 (512 .. 2048 bits)
 * (sober_level_reduce_tani_supported); popcount(x & y) as v_mfma_i32_16x16x64_i8 on bits expanded to bytes.   */
int sober_level_reduce_tani_supported(int dt);
int sober_level_reduce_tani(const void* rows, const double* rows_norm, int n_rows, const void* cand,
                            const double* cand_norm, int dt, const int32_t* idx, int64_t pos0, int64_t count, int S,
                            const double* mu, const double* wmul, double outputscale, int n_chunks, double* partG,
                            int ldg, int col0, double* partTot, int64_t tot_limit, void* stream);

/* Xp[rank[s]][0:n] = X[s][0:n], objp[rank[s]] = ocol[s] for the sets s with 0 <= rank[s] < n1: the survivors of a
 * Caratheodory step in rank order (SOBER/_rchq.py:87-91, :177-181), without a host decision.  Xp: n1 x n, objp: n1. */
int sober_rank_scatter(const double* X, int ldx, int N, int n, const double* ocol, const int32_t* rank, int n1,
                       double* Xp, double* objp, void* stream);
/* *dst = v, in stream order (device memory; a one-thread kernel). */
int sober_set_i64(int64_t* dst, int64_t v, void* stream);

/* Queued-level forms (see sober_level_loop): the same kernels with the level size read from device memory.
 * sober_level_reduce_mfma_queued: leftover = 0: positions [0, *dR), set masses over [0, E S) (S = S_main);
 * leftover = 1: the leftover positions [E S_main, *dR) over S pseudo-sets.  Launch sized for count_ub positions
 * and n_chunks_ub chunks (the LARGEST count sober_level_chunks can return up to count_ub, not its value there). */
int sober_level_reduce_mfma_queued(int kind, const double* rows, int n_rows, const double* cand, int da,
                                   const int32_t* idx, int64_t count_ub, int S, int S_main, int leftover,
                                   const double* mu, const double* wmul, double outputscale, int n_chunks_ub,
                                   double* partG, int ldg, double* partTot, const int64_t* dR, void* stream);
/* Both placements in ONE launch: the workgroups of the leftover launch (n_xcols pseudo-sets, at most S - 1 positions,
 * partial sums extraG / extraTot with row stride n_xcols, n_xchunks_ub = sober_level_parts_mfma_cap(n_rows,
 * ceil((S - 1) / n_xcols), n_xcols)) ride behind the main launch's: the same arithmetic as the two launches
 * (SOBER/_rchq.py:128-136 and :153-164), one dispatch less per level. */
int sober_level_reduce_mfma_queued_pair(int kind, const double* rows, int n_rows, const double* cand, int da,
                                        const int32_t* idx, int64_t count_ub, int S, int n_xcols, const double* mu,
                                        const double* wmul, double outputscale, int n_chunks_ub, double* partG,
                                        int ldg, double* partTot, int n_xchunks_ub, double* extraG, double* extraTot,
                                        const int64_t* dR, void* stream);
/* chunked = 1: the partial sums come from the fingerprint kernel (sober_level_reduce_tani_queued: sober_level_chunks_tani
 * of them at the level's exact size), 0: from the matrix-core FP64 kernel (sober_level_parts_mfma slots per tile). */
int sober_sum_partials_queued(const double* partG, const double* partTot, int n_rows, int ldg, int S,
                              const double* extraG, const double* extraTot, int n_xcols, double* G, int ldo,
                              double* tot, const int64_t* dR, int chunked, void* stream);
/* The Tanimoto level kernel with the level size read from device memory (leftover = 0 / 1 like
 * sober_level_reduce_mfma_queued); n_chunks_ub >= sober_level_chunks_tani_cap(n_rows, ceil(count_ub / S), S) (the kernel
 * and sober_sum_partials_queued(chunked = 1) derive the exact count from the level's size with the same formula;
 * a smaller n_chunks_ub is SOBER_E_ARG).                                                                        */
int sober_level_reduce_tani_queued(const void* rows, const double* rows_norm, int n_rows, const void* cand,
                                   const double* cand_norm, int dt, const int32_t* idx, int64_t count_ub, int S,
                                   int S_main, int leftover, const double* mu, const double* wmul, double outputscale,
                                   int n_chunks_ub, double* partG, int ldg, double* partTot, const int64_t* dR,
                                   void* stream);
int sober_level_reduce_tani_queued_pair(const void* rows, const double* rows_norm, int n_rows, const void* cand,
                                        const double* cand_norm, int dt, const int32_t* idx, int64_t count_ub, int S,
                                        int n_xcols, const double* mu, const double* wmul, double outputscale,
                                        int n_chunks_ub, double* partG, int ldg, double* partTot, int n_xchunks_ub,
                                        double* extraG, double* extraTot, const int64_t* dR, void* stream);
int sober_level_chunks_cap(int n_rows, int64_t e_total_ub, int S);
/* the same two for the fingerprint matrix-core kernel (sober_level_reduce_tani*: one workgroup per compute unit, the chunk
 * count chosen for one or two rounds of the 256 compute units -- csrc/common.hpp: level_chunks_tani_for) */
int sober_level_chunks_tani(int n_rows, int64_t pos0, int64_t count, int S);
int sober_level_chunks_tani_cap(int n_rows, int64_t e_total_ub, int S);
/* K7 (SOBER/_rchq.py:198-221) with R = *dR_cur and n_keep = keep_rank[S] read on the device; *dR_next = the next
 * level's R, or -1 with mu and the list untouched when the host loop has to take over (R <= S, no progress,
 * n_keep outside 1..S, more than R_ub_next survivors).                                                       */
int sober_level_update_queued(const int32_t* idx_cur, int64_t R_ub, int S, const int32_t* keep_rank,
                              const double* w_star, const double* tot, double* mu, int32_t* idx_new,
                              const int64_t* dR_cur, int64_t* dR_next, int64_t R_ub_next, void* stream);

/* ---- levels whose set sums are already held (round 6, csrc/level_class.hip) -----------------------------------------
 * Survivors are compacted element-major (SOBER/_rchq.py:198-221): with n_keep = b kept sets, S = 2b and no leftovers the
 * survivor of element e in the kept set of rank k moves to element e div 2, set (e mod 2) b + k, weight mu w*_k / tot_k
 * (:204-205).  The next level's set sums (:116-126) are therefore level l's sums over the elements of one parity, times
 * w*_k / tot_k: no kernel evaluation.  Class sums over c = e mod 2^D ARE set sums with 2^D S sets, so level 0 runs the level
 * kernel over S' = 2^D S sets (sober_level_reduce_mfma_wpt, waves per tile from sober_level_class_wpt, partial slots
 * sober_level_class_slots(wpt)); sober_class_sum turns its partial slots into class sums Gc (n_rows x CL S), class masses
 * totc (CL S) and the level's own G (n_rows x S) / tot (S); sober_class_derive_queued forms level l + 1 (CL / 2 classes:
 * Gn, totn -- untouched when CL = 2 -- and G / tot) from level l's (CL classes) with scale[k] = w*_k / tot_k and sof[k] =
 * the set of rank k, both written by sober_level_update_queued_cls -- which also stops the queued chain (*dR_next = -1)
 * unless exactly need_keep sets survived and the level had no leftovers.  *dR <= S: the derive launch leaves at once.   */
int sober_level_class_wpt(int n_rows, int64_t e_total, int S);
int sober_level_class_slots(int wpt);
int sober_level_reduce_mfma_wpt(int kind, const double* rows, int n_rows, const double* cand, int da,
                                const int32_t* idx, int64_t count, int S, const double* mu, const double* wmul,
                                double outputscale, int wpt, int n_chunks, double* partG, int ldg, double* partTot,
                                void* stream);
int sober_class_sum(const double* partG, const double* partTot, int n_chunks, int n_rows, int S, int CL,
                    double* Gc, double* totc, double* G, double* tot, void* stream);
int sober_class_derive_queued(const double* Gc, const double* totc, int n_rows, int S, int CL, const double* scale,
                              const int32_t* sof, double* Gn, double* totn, double* G, double* tot, const int64_t* dR,
                              void* stream);
/* the general form of the two: need_keep = 0 -- no class tables --, R_known >= 0: the level's size by value (the chain's first
 * level, known to the host) instead of *dR_cur                                                                        */
int sober_level_update_queued_ex(const int32_t* idx_cur, int64_t R_ub, int S, const int32_t* keep_rank,
                                 const double* w_star, const double* tot, double* mu, int32_t* idx_new,
                                 const int64_t* dR_cur, int64_t* dR_next, int64_t R_ub_next, int need_keep,
                                 double* cls_scale, int32_t* cls_sof, int64_t R_known, void* stream);
int sober_level_update_queued_cls(const int32_t* idx_cur, int64_t R_ub, int S, const int32_t* keep_rank,
                                  const double* w_star, const double* tot, double* mu, int32_t* idx_new,
                                  const int64_t* dR_cur, int64_t* dR_next, int64_t R_ub_next, int need_keep,
                                  double* cls_scale, int32_t* cls_sof, void* stream);
/* The depth D (0 .. SOBER_CLASS_MAX_DEPTH) sober_level_moments / sober_level_loop use for a pool of R live positions in
 * sets of S (variant: SOBER_LEVEL_MFMA or SOBER_LEVEL_TANI, 0 for the others): the largest D with R % (2^D S) == 0, at least
 * two elements per set left at level D, 2^D x the class launch's partial slots within SOBER_LEVEL_MAX_CHUNKS -- and, for the
 * fingerprint kernel (one workgroup per compute unit), a class launch that still fills the chip.                       */
#define SOBER_CLASS_MAX_DEPTH 4
int sober_level_class_depth(int variant, int n_rows, int64_t R, int S);

/* KMeans of SOBER/_weights.py:100-126: Lloyd, centroids initialised to the first K rows, exactly
 * `iters` iterations, first-index argmin (a NaN distance wins like torch.argmin), empty cluster ->
 * NaN centroid.  X is (N, d) row-major raw points.  labels: N int32.  ws: sober_kmeans_ws_bytes (a smaller one
 * selects the slower forms that need less: the FP64-only E step, or without any the (x - c)^2 kernel).       */
int64_t sober_kmeans_ws_bytes(int64_t N, int d, int K);
/* Shapes whose E step is screened on the BF16 matrix cores (csrc/kmeans.hip): byte offset, inside a workspace of
 * sober_kmeans_ws_bytes, of a uint32 = the number of points the exact pass had to decide, summed over the
 * iterations of the last call; -1 for a shape that is not screened.                                            */
int64_t sober_kmeans_stat_offset(int64_t N, int d, int K);
/* The workspace that selects the screened E step at every shape that allows it (stat_offset >= 0) -- sober_kmeans_ws_bytes
 * asks for it only from the pool size on where it is the faster form (N x ceil((d + 1) / 4) >= 300000); 0 otherwise. */
int64_t sober_kmeans_ws_bytes_screened(int64_t N, int d, int K);
int sober_kmeans_lloyd(const double* X, int64_t N, int d, int K, int iters,
                       double* centroids, int32_t* labels, void* ws, int64_t ws_bytes, void* stream);

/* GP prediction over a pool in ONE launch (SOBER/_gp.py:212-238, SOBER/_pi.py:20-38; csrc/predict.hip): per candidate
 * mean = c0 + k(x, X_obs) alpha, var = kxx - k W k^T + noise and, with lfi_out != NULL, pi = Phi((mean - eta) / sqrt(var))
 * (log_flag: log(pi + FP32 eps)) -- the K(X_obs, x) column evaluated once into LDS, W k on the FP64 matrix cores, nothing
 * of size n_obs x N ever in memory.  obs / cand: prepared point sets (sober_scale_points rows of dt doubles, or
 * sober_pack_bits words + popcounts for Tanimoto); W: n_obs x n_obs, SYMMETRIC (S S^T of SOBER/_gp.py:277), ld ldw;
 * kxx_const: k(x, x) of the continuous kernels (Tanimoto takes it from the popcounts); mean_out may be NULL; eta_ptr != NULL:
 * the threshold is read from device memory (a max another kernel left there) instead of the argument.
 * sober_predict_fused_supported: n_obs <= 511 (255 until round 6; eight row tiles per wave beyond) and a register-tiled point dimension; otherwise SOBER_E_DIM and the
 * caller keeps the materialised route (sober_pairwise + sober_dgemm + sober_predict_finish).                          */
int sober_predict_fused_supported(int kind, int n_obs, int dt);
int sober_predict_fused(int kind, const void* obs, const double* obs_norm, int n_obs, const void* cand,
                        const double* cand_norm, int64_t N, int dt, double outputscale, const double* W, int ldw,
                        const double* alpha, double c0, double kxx_const, double noise, double* mean_out, double* var_out,
                        double eta, const double* eta_ptr, double* lfi_out, int log_flag, void* stream);
/* The same launch from the ROOT of W (round 6): St = S^T (n_obs x n_obs, row stride ldst) with W = S S^T -- W is never formed,
 * var = kxx - |St k|^2 + noise.  *tri_flag != 0 (a device word; NULL = 0) promises that St is LOWER triangular -- gpytorch's
 * covar_cache of an exact GP is the transposed inverse Cholesky factor, SOBER/_gp.py:272-277 --: the tile products above the
 * diagonal are skipped (91 of 169 at n_obs = 200; the busiest wave carries 28 instead of 52).                          */
int sober_predict_fused_root(int kind, const void* obs, const double* obs_norm, int n_obs, const void* cand,
                             const double* cand_norm, int64_t N, int dt, double outputscale, const double* St, int ldst,
                             const int32_t* tri_flag, const double* alpha, double c0, double kxx_const, double noise,
                             double* mean_out, double* var_out, double eta, const double* eta_ptr, double* lfi_out, int log_flag,
                             void* stream);
/* Posterior variance (and optionally the LFI weight pi) over a pool, SOBER/_gp.py:212-238 and
 * SOBER/_pi.py:31-38: KX = k(X_obs, pool) (n_obs x N, sober_pairwise), V = W KX (sober_dgemm);
 * var[j] = k(x_j, x_j) - sum_i KX[i][j] V[i][j] + noise; pi[j] = Phi((mean[j] - eta)/sqrt(var[j])),
 * log_flag: log(pi + FP32 eps).  norms != NULL selects the Tanimoto k(x, x).                        */
int sober_predict_finish(const double* KX, const double* V, int n_obs, int64_t N, int64_t ld,
                         const double* mean, double kxx_const, const double* norms, double outputscale,
                         double noise, double* var_out, double eta, double* lfi_out, int log_flag,
                         void* stream);

/* cleansing_weights of SOBER/_weights.py:21-38, in place: w < eps -> 0, inf/nan -> eps, then
 * normalise by the sum (or 1/n everywhere when the sum is 0).  ws: sober_reduce_ws_bytes(n).     */
int64_t sober_reduce_ws_bytes(int64_t n);
int sober_cleansing_weights(double* w, int64_t n, double eps, void* ws, int64_t ws_bytes, void* stream);

/* Draws of the weighted KDE prior, SOBER/_wkde.py:162-219 (MultivariateNormal(X_c, Sigma).sample and the
 * bounds test of rejection_sampling): x[r] = Xobs[comp[r]] + L eps[r] with L L^T = Sigma (d x d row-major,
 * lower triangle read; d <= 64), inside[r] = 1 iff lo <= x[r] <= hi component-wise (lo = hi = NULL: no
 * bounds, inside may be NULL).  eps: n x d standard normals supplied by the caller's generator.          */
int sober_wkde_draw(const double* eps, int64_t n, int d, const int32_t* comp, const double* Xobs, int ldx,
                    const double* L, const double* lo, const double* hi, double* x, int32_t* inside,
                    void* stream);

/* K1-K3 for a kernel matrix that is resident in HBM: Kmat[c*ldk + r] = k(x_r, cand_c), r < n_rows <= 1024
 * (candidate-major).  Same contract as sober_level_reduce -- partG[(chunk*n_rows + r)*ldg + col0 + s] =
 * sum over the chunk's elements of Kmat[idx[p - pos0]][r] * mu * (wmul), p = e*S + s in [pos0, pos0+count);
 * partTot likewise for the set masses (positions < tot_limit) -- but the kernel values are gathered, not
 * evaluated.  Used for callables that follow the kernel protocol of SOBER/_rchq.py:9,20 and for BASQ's
 * g-space kernel.  n_chunks from sober_level_chunks.                                                  */
int sober_level_gather(const double* Kmat, int n_rows, int64_t ldk, const int32_t* idx, int64_t pos0,
                       int64_t count, int S, const double* mu, const double* wmul, int n_chunks,
                       double* partG, int ldg, int col0, double* partTot, int64_t tot_limit, void* stream);

/* Epilogue of ScaleMmltGP.gspace_kernel, SOBER/BASQ/_scale_mmlt.py:256-275, in place on K (n x m):
 * K[c][r] <- mug_rows[r] * mug_cand[c] * (exp(K[c][r] - corr[c][r]) - 1), where on entry K = os*k(cand, x)
 * and corr = k(cand, X) W k(X, x) (the posterior correction of SOBER/_gp.py:295).                      */
int sober_gspace_finish(double* K, const double* corr, int64_t n, int m, int64_t ldk, int64_t ldc,
                        const double* mug_cand, const double* mug_rows, void* stream);

/* ---- level executor: one level of Mod_Tchernychova_Lyons (SOBER/_rchq.py:116-175) per call ----------------
 * The halving loop is a chain of ~10 small launches per level with one host decision at its end (how many
 * sets survived); issuing them one by one from the host language costs more than the kernels run.  A job
 * describes the per-step constants once and the per-level position range; sober_level_moments enqueues
 *   level_reduce (variant) -> level_reduce over the leftovers -> sum_partials -> P G          => Xtr, tot
 * and sober_level_car enqueues
 *   barycentres -> Caratheodory step (sober_car_device) -> async copy of keep_rank[0:S] and n_keep to h_flags.
 * Between the two a sharded run all-reduces Xtr and tot.  Nothing synchronises; the caller
 * waits on the stream before reading h_flags (pinned host memory, S + 1 int32).
 * Workspaces are caller-owned with fixed capacities: partG SOBER_LEVEL_MAX_CHUNKS * n_rows * S doubles,
 * partTot MAX_CHUNKS * S, extraG MAX_CHUNKS * n_rows * SOBER_LEVEL_XS, extraTot MAX_CHUNKS * XS.            */
#define SOBER_LEVEL_VALU        0    /* sober_level_reduce      (rows/cand = scaled points or packed words) */
#define SOBER_LEVEL_MFMA        1    /* sober_level_reduce_mfma (rows/cand = augmented points)              */
#define SOBER_LEVEL_GATHER      2    /* sober_level_gather      (cand = Kmat, candidate-major, ld = kmat_ld) */
#define SOBER_LEVEL_TANI        3    /* sober_level_reduce_tani (rows/cand = packed words, 512..2048 bits)   */
#define SOBER_LEVEL_MAX_CHUNKS  64
#define SOBER_LEVEL_XS          16   /* pseudo-sets of the leftover launch, folded into set S-1             */
typedef struct sober_level_job {
    /* per step */
    int32_t variant, kind;
    const void* rows;   const double* rows_norm;
    const void* cand;   const double* cand_norm;
    int32_t n_rows, dim;                        /* dim: DT (VALU), DA (MFMA), unused (GATHER)               */
    int64_t kmat_ld;
    const double* wmul; double outputscale;
    int32_t S, n;
    const double* P;                            /* n x n_rows projection, SOBER/_rchq.py:148                */
    double *partG, *partTot, *extraG, *extraTot;
    double *G, *Xtr, *tot;                      /* n_rows x S, n x S, S                                     */
    double *X_tmp;                              /* S x n barycentres                                        */
    int32_t* keep_rank;                         /* S + 1 int32: keep_rank[S] = n_keep                       */
    double *w_star, *mu_out;                    /* S each                                                   */
    void* car_ws; int64_t car_ws_bytes;         /* sober_car_ws_bytes(S, n + 1)                             */
    int32_t* h_flags;                           /* pinned host, S + 1 int32                                 */
    void* ev[4];                                /* optional hipEvent_t pairs for the level kernel's main launch (ev[0],
                                                   ev[1]) and its leftover launch (ev[2], ev[3]).  Matrix-core variants:
                                                   carried IN the dispatch (sober_set_launch_events: the kernel's own
                                                   begin / end timestamps, what rocprofv3 reports); the others: recorded
                                                   on the stream before / after the launch                          */
    /* per level */
    const int32_t* idx; int64_t pos0, count, E;
    double* mu;                                 /* read by the set sums; rescaled by sober_level_loop's updates */
    int32_t phase;                              /* sober_level_moments: 0 = everything, 1 = set sums only (up to G, tot),
                                                   2 = projection only (Xtr = P G).  The set sums of the first level do
                                                   not depend on the Nystrom basis: they run while the host still works
                                                   on it.                                                                */
    /* queued levels (optional, MFMA variant): SOBER_LEVEL_QUEUE + 1 int64 each -- dR device, h_dR pinned host.
       NULL: every level is sized by the host after a stream synchronisation.                                  */
    int64_t* dR; int64_t* h_dR;
    int32_t car_mode;                           /* SOBER_CAR_DEFAULT / SOBER_CAR_SAFE for the Caratheodory steps; the loops
                                                   raise it to SOBER_CAR_SAFE after a give-up (in/out)                   */
    uint64_t ev_used[2];                        /* out (sober_level_loop with events): bit l of [0] / [1] = the event pair of
                                                   level l's main / leftover launch was attached to a launch          */
    /* levels derived from class sums (optional, MFMA / TANI variants with queued levels; see sober_class_derive_queued):
       class_depth = D > 0 asks sober_level_moments(phase 1) for level 0's sums by 2^D element classes -- and tells
       sober_level_loop (first_sums_ready) that they are there: levels 1 .. D are then gathered and scaled, not evaluated.
       Gc[2] / totc[2]: ping-pong class sums, n_rows x 2^D S and 2^D S doubles each; cls_scale (S doubles), cls_sof (S int32) */
    int32_t class_depth;
    double* Gc[2]; double* totc[2];
    double* cls_scale; int32_t* cls_sof;
} sober_level_job;
#define SOBER_LEVEL_QUEUE 24
int sober_level_moments(const sober_level_job* job, void* stream);
int sober_level_car(const sober_level_job* job, void* stream);
/* After a sober_level_car whose verdict was n_keep = -1 (job->h_flags[S] < 0 once the stream is synchronised): the step
 * alone is redone in SOBER_CAR_SAFE mode on the barycentres that are still in place, synchronised, and job->car_mode
 * stays SOBER_CAR_SAFE.  0 = a fresh verdict is in h_flags; SOBER_E_EXCHANGE = that mode does not cover the size.    */
int sober_level_car_retry(sober_level_job* job, void* stream);
/* The whole halving loop of an UNSHARDED pool (SOBER/_rchq.py:116-221, repeated while R > S) in one call: per level
 * sober_level_moments, sober_level_car, a stream synchronisation (the host has to know how many sets survived
 * before it can size the next level), sober_level_update, buffer swap -- without a round trip through the host
 * language between them.  idx_a holds the R live positions on entry, idx_b is the other half of the ping-pong.
 * first_sums_ready != 0: the set sums of the first level are already in job->G / job->tot (phase 1 was run by the
 * caller).  events: NULL or 4 hipEvent_t per level (see job->ev).  Outputs: level_R[l] = live positions at level l,
 * *n_levels, *R_final (<= S), *in_b = 1 when the final live list is in idx_b.  Returns SOBER_E_NOPROGRESS when a
 * level cancels nothing (the reference would loop forever, :241-242), SOBER_E_WS when max_levels is too small.  */
#define SOBER_E_NOPROGRESS -4
/* With job->dR / job->h_dR set (and the MFMA variant) the loop first ENQUEUES every level that certainly exists
 * (at most b of a level's 2b sets survive, so the number of live positions after a level is known up to its
 * leftovers: R_new in [E b, E b + r]) without any host decision in between: the launches are sized from the upper
 * bound, every kernel reads the exact R from dR[level] (written by the previous level's sober_level_update_queued),
 * surplus workgroups exit.  One synchronisation for all of them; the last one or two levels, whose existence
 * depends on the leftovers, and anything irregular (no progress, a step that kept more than b sets) fall back to
 * the synchronised loop from the first level the chain did not complete.                                      */
int sober_level_loop(sober_level_job* job, int64_t R, int32_t* idx_a, int32_t* idx_b, int first_sums_ready,
                     void** events, int max_levels, int64_t* level_R, int32_t* n_levels, int64_t* R_final,
                     int32_t* in_b, void* stream);
/* The same queued chain for the ACQUISITION-GUIDED branch (calc_obj: SOBER/_rchq.py:67-69, :138-150, :173-196; round 6): per level
 * the set sums, the objective's set sums and barycentre column (sober_obj_set_sums_queued), the Caratheodory step with n + 2
 * functions, the second elimination's direction (sober_null_vector) and the elimination (sober_second_elimination_rows), the
 * update -- every verdict read on the device, one synchronisation for the chain.  A level whose outcome is not the regular one
 * (the first step left other than n + 2 sets, a rank-deficient survivors' matrix, a give-up, no progress) stops the chain with
 * weights and list untouched; *n_levels complete levels and *R_final live positions come back, the rest (that level, the
 * leftover-dependent last levels, the final direct level) is the caller's synchronised route.  job: as for sober_level_loop
 * with car_ws sized by sober_car_ws_bytes(S, n + 2); class sums are not used.  SOBER_E_DIM: n beyond sober_null_vector.     */
typedef struct sober_obj_job {
    const double* obj;          /* calc_obj by candidate (already negated: :69)                                        */
    double* X_tmp;              /* S x (n + 1): the barycentres with the objective's column                            */
    double* ocol;               /* S: that column, contiguous                                                          */
    int32_t* kr1; double* w1; int32_t* nk1;     /* the first step's ranks (S), weights (S), survivor count             */
    double* null_row;           /* S                                                                                   */
    int32_t* status;            /* sober_null_vector's verdict                                                         */
} sober_obj_job;
int sober_obj_job_size(void);
int sober_level_loop_obj(sober_level_job* job, const sober_obj_job* obj_job, int64_t R, int32_t* idx_a, int32_t* idx_b,
                         int first_sums_ready, int max_levels, int64_t* level_R, int32_t* n_levels, int64_t* R_final,
                         int32_t* in_b, void* stream);
/* The level loop of a ROW-SHARDED pool, native: rank `rank` of `world` owns the list positions
 * [bounds[rank], bounds[rank+1]) (bounds: world + 1 int64, updated in place after every level from the replicated
 * verdict: closed-form compaction, no candidate row moves).  Per level: local set sums -> ONE all-reduce of the flat
 * (n S + S) double buffer job->Xtr (job->tot must follow it directly) through `allreduce(comm, buf, n, stream)` ->
 * replicated Caratheodory step -> one synchronisation -> local sober_level_update.  Runs while the global number
 * of live positions exceeds max(S, R_stop); the caller gathers the live rows once and finishes replicated
 * (sober_level_loop + sober_level_final).  `allreduce` = sober_rccl_allreduce_f64 with an RCCL communicator from
 * sober_rccl_comm_init in production; any function with that contract otherwise (the one-GPU tests run several ranks
 * through a host transport).                                                                                    */
typedef int (*sober_allreduce_fn)(void* comm, double* buf, int64_t n, void* stream);
int sober_level_loop_sharded(sober_level_job* job, int rank, int world, int64_t* bounds, int32_t* idx_a,
                             int32_t* idx_b, int first_sums_ready, sober_allreduce_fn allreduce, void* comm,
                             int64_t R_stop, int max_levels, int64_t* level_R, int32_t* n_levels, int32_t* in_b,
                             void* stream);
/* RCCL bound at run time (csrc/rccl_link.cpp): dlopen of the library the host names (torch's own librccl.so), a
 * communicator from a 128-byte unique id (rank 0 draws it, the host side distributes it), the in-place FP64 sum. */
int sober_rccl_load(const char* path);
int sober_rccl_unique_id(char* out128);
int sober_rccl_comm_init(const char* id128, int rank, int world, void** comm);
int sober_rccl_comm_destroy(void* comm);
int sober_rccl_allreduce_f64(void* comm, double* buf, int64_t n, void* stream);
int64_t sober_rccl_allreduce_ptr(void);     /* the address of sober_rccl_allreduce_f64, as an integer */

/* One-shot direct-peer all-reduce (csrc/peer_reduce.hip): every rank owns an exchange region in its own
 * memory, maps its peers' regions (IPC handles between processes: sober_peer_create hands out the 64-byte handle of
 * mine, sober_peer_connect takes all `world` of them in rank order; sober_peer_connect_ptrs: plain device pointers,
 * ranks inside one process) and a call is ONE kernel per rank that publishes its message, waits for the peers' flags
 * (bounded) and sums the `world` contributions in RANK ORDER -- the same bits on every rank, no ring steps.
 * sober_peer_allreduce_f64 is a sober_allreduce_fn (n <= n_max doubles).  sober_peer_status, after the stream has been
 * synchronised: 0, or SOBER_E_EXCHANGE when a wait ran out (a rank that never arrived within ~30 s) -- the rank's own
 * message of the FIRST failing call is then copied back to `restore` (n doubles; may be NULL).  The time-out is decided
 * per rank and is fatal for the sharded run (sober_level_loop_sharded returns it): the group-wide fall-back to RCCL
 * happens at set-up, when the self-check does not pass on every rank, not afterwards. */
int64_t sober_peer_region_bytes(int64_t n_max);
int sober_peer_create(int rank, int world, int64_t n_max, void** comm, char* handle64);
int sober_peer_connect(void* comm, const char* handles);
int sober_peer_connect_ptrs(void* comm, void* const* regions);
int64_t sober_peer_region(void* comm);      /* the address of my region, as an integer */
int sober_peer_set_spin_limit(void* comm, unsigned spin_limit);
int sober_peer_allreduce_f64(void* comm, double* buf, int64_t n, void* stream);
int64_t sober_peer_allreduce_ptr(void);      /* the address of sober_peer_allreduce_f64, as an integer */
int sober_peer_status(void* comm, double* restore, int64_t n, void* stream);
int sober_peer_destroy(void* comm);

/* The final direct level of an unsharded pool (n + 1 < R <= S, SOBER/_rchq.py:77-114) in one call: kernel columns of
 * the R live candidates (sober_pairwise on the SCALED points rows_sc / cand_sc, dt doubles or words per row), P K,
 * transpose, their weights, the Caratheodory step, mu[0:N] = 0 and the device-side write-back.  K: n_rows x R scratch,
 * mu_live: R scratch.  out_idx / out_w: R entries, the first n_keep valid; n_keep arrives in job->h_flags[S] (pinned)
 * once the stream has been synchronised.  n_keep = -1 (a give-up of the step, job->car_mode = SOBER_CAR_DEFAULT): mu
 * has NOT been touched -- call again with job->car_mode = SOBER_CAR_SAFE or take the host route.               */
int sober_level_final(const sober_level_job* job, const void* rows_sc, const double* rows_norm, const void* cand_sc,
                      const double* cand_norm, int dt, const int32_t* idx, int R, int64_t N, int64_t row_offset,
                      double* K, double* mu_live, int64_t* out_idx, double* out_w, void* stream);
/* The row table's side of a step's plan in one call (RBF / Matern-5/2): rows = [X_nys; X_obs] / lengthscale
 * ((M + n_obs) x dt, dt = padded dimension of sober_scale_points), Kall = k(rows, X_nys) ((M + n_obs) x M),
 * W = S_cache S_cache^T (SOBER/_gp.py:277), T = K(X_nys, X_obs) W (M x n_obs, SOBER/_gp.py:293-295) and the Gram matrix
 * of SOBER/_rchq.py:35, G = K(X_nys, X_nys) - T K(X_obs, X_nys) (M x M).  n_obs = 0 (mode "kernel"): rows and
 * G = k(X_nys, X_nys); X_obs, S_cache, Kall, W, T may then be NULL. */
int sober_plan_rows(int kind, const double* X_nys, int M, int64_t ld_nys, const double* X_obs, int n_obs,
                    int64_t ld_obs, int d, const double* lengthscale, int ls_len, double outputscale,
                    const double* S_cache, int ld_s, double* rows, int dt, double* Kall, double* W, double* T,
                    double* G, void* stream);

/* sober_level_loop followed -- without going back to the host language -- by sober_level_final when the loop ends on a
 * list of n + 1 < R <= S positions that the Caratheodory kernels of job->car_mode cover (final->done = 1: the stream
 * has been synchronised, n_keep is in job->h_flags[S], out_idx / out_w hold the result; done = 0: nothing beyond
 * sober_level_loop happened).  The arguments of sober_level_final travel in `final` (NULL: sober_level_loop). */
typedef struct sober_final_job {
    const void* rows_sc; const double* rows_norm;
    const void* cand_sc; const double* cand_norm;
    int32_t dt, done;
    int64_t N, row_offset;
    double *K, *mu_live;
    int64_t* out_idx; double* out_w;
    /* cand_sc == NULL (continuous kernels, round 6): the pool is not kept in scaled form -- the final level scales the rows it
       touches itself (sober_scale_points_idx) from the RAW pool: cand_raw (N x ld_raw doubles, d_raw coordinates), the
       lengthscale (ls_len 1 or d_raw, device) and sc_buf, 2 b x dt doubles of scratch                                      */
    const double* cand_raw; int64_t ld_raw; int32_t d_raw, ls_len; const double* ls; double* sc_buf;
} sober_final_job;
/* sober_level_final with its arguments in the struct (what sober_level_loop_final runs; the raw-pool form exists only here) */
int sober_level_final_job(const sober_level_job* job, const sober_final_job* final, const int32_t* idx, int R, void* stream);
int sober_final_job_size(void);
int sober_level_loop_final(sober_level_job* job, sober_final_job* final, int64_t R, int32_t* idx_a, int32_t* idx_b,
                           int first_sums_ready, void** events, int max_levels, int64_t* level_R, int32_t* n_levels,
                           int64_t* R_final, int32_t* in_b, void* stream);
/* P = [U diag(mean), -(U diag(mean)) T]  (s x (M + n_obs), row-major): phi(x) = P k([X_nys; X_obs], x) is the vector of
 * Nystrom test functions U C(X_nys, x) of SOBER/_rchq.py:78,148,156 with the posterior correction of SOBER/_gp.py:295
 * folded in (it is linear).  Ut: s x M; mean (M, weighted mode: SOBER/_kernel.py:41) may be NULL; T = K(X_nys, X_obs) W
 * (M x n_obs) may be NULL (mode "kernel": P = U diag(mean), s x M).                                                  */
int sober_projection(const double* Ut, int s, int M, const double* mean, const double* T, int n_obs, double* P,
                     void* stream);

/* ker_svd_sparsify (SOBER/_rchq.py:34-39) from the Gram matrix on, in ONE call (csrc/nystrom_exec.cpp): make_cov_psd
 * (SOBER/_utils.py:131-157: |cov| + symmetry test, every rung of the jitter ladder probed at once, the first positive
 * definite rung or the diagonal fallback applied on the device), the range finder of torch.svd_lowrank on the matrix
 * cores (A R, then `niter` times A^H Q and A Q, CholeskyQR after each product: one pass for the intermediate blocks,
 * two for the last), U = Q^T, the projection P, and the copy of every flag to pinned memory.  No host decision inside.
 * The flag block (sober_nystrom_flags_bytes; zeroed by the call):
 *   pivots[n_rungs + 1] f64 (smallest pivot per rung, then the largest diagonal entry of |cov|) |
 *   pivs_rf[2 (1 + 2 niter)] f64 | flags[2 + n_rungs] i32 ([0] != 0: not exactly symmetric, [1] rung taken, [2:] the
 *   rungs' info) | infos_rf[2 (1 + 2 niter)] i32 -- read as sober_amd/_ops_hip.py:nystrom_basis_device does.          */
typedef struct sober_nystrom_job {
    int32_t M, s, n_rungs, niter, probe_mc;     /* Gram size, basis size, ladder rungs (max_iter + 1), power iterations,
                                                   != 0: eight workgroups per rung (sober_cholesky_probe_mc)          */
    int32_t reserved0;                          /* (0)                                                                  */
    const double* G;                            /* M x M Gram, row-major ld M: kernel(pt, pt) of SOBER/_rchq.py:35      */
    const double* shifts;                       /* n_rungs jitter totals 1e-5 (2^k - 1), device                        */
    const double* R;                            /* M x s standard normals, device: the CPU generator's draw            */
    double* C;                                  /* M x M scratch: |cov|, then the repaired matrix                      */
    double* chol_work;                          /* n_rungs x M x M scratch                                             */
    void* probe_ws; int64_t probe_ws_bytes;     /* sober_cholesky_probe_mc_ws_bytes (probe_mc)                         */
    double* Y[2];                               /* M x s each                                                          */
    double* Gm;                                 /* s x s                                                               */
    double* xinv;                               /* ceil(s / 32) x 1024                                                 */
    void* flags_block; int64_t flags_bytes;     /* device, >= sober_nystrom_flags_bytes                                */
    void* h_flags_block;                        /* pinned host copy of the block                                       */
    double* Ut;                                 /* out: s x M, the basis as rows                                       */
    const double* T; int32_t n_obs;             /* projection (P == NULL skips it): see sober_projection               */
    const double* mean_nys; double* P;
} sober_nystrom_job;
int64_t sober_nystrom_flags_bytes(int n_rungs, int niter);
/* phase 0: everything; 1: make_cov_psd only (flags zeroed, probes, ladder; R not needed yet -- the host steps its
 * generator for R while the probes run); 2: the rest (range finder, U, projection, flag copy).                      */
int sober_nystrom_basis(const sober_nystrom_job* job, int phase, void* stream);

/* Arms a pair of hipEvent_t for the NEXT level-kernel launch (matrix-core variants) made by the calling thread: the
 * launch carries them in its dispatch (hipExtLaunchKernelGGL), so hipEventElapsedTime(start, stop) is the kernel's own
 * duration -- the number rocprofv3 --kernel-trace reports for it -- and no marker packet enters the stream.
 * (NULL, NULL) disarms.                                                                                              */
int sober_set_launch_events(void* start, void* stop);
/* ev0, ev1 (hipEvent_t) recorded back to back: the empty bracket (what two marker packets cost the stream).    */
int sober_record_event_pair(void* ev0, void* ev1, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SOBER_HIP_H */
