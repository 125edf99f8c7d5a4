// Dense FP64 Cholesky for the Nystrom side of the path, as ONE persistent workgroup (blocked,
// right-looking; panel in LDS, trailing update register-tiled), plus the row-wise triangular solve
// that turns it into a QR factorisation (CholeskyQR).
//
//  * SOBER/_utils.py:117-157 (is_psd / make_cov_psd) probes `torch.linalg.cholesky(cov + jitter I)` on
//    the N_nys x N_nys Gram up to 12 times; only success/failure is used.  k_chol answers that on
//    the device (info = 0: positive definite; info = j+1: the leading minor of order j+1 is not).
//  * torch.svd_lowrank (SOBER/_rchq.py:37) orthonormalises five M x q blocks with Householder QR; an
//    orthonormal basis of the same flag of subspaces comes from Q = Y R^-1 with R^T R = Y^T Y
//    (applied twice for orthogonality to rounding); Q differs from LAPACK's only by column signs,
//    which cancel in U = Q U_B.
#include "common.hpp"

namespace sober {

constexpr int CH_NB = 32;            // panel width
constexpr int CH_T = 512;             // 8 waves: 256 VGPRs per lane for the register-resident diagonal block
constexpr int CH_LDPP = CH_NB + 4;   // row stride of the LDS panel: 16 rows x 4 k of an MFMA fragment fall on distinct bank pairs
constexpr int CH_MAXN = 536;         // panel (n - NB) x LDPP doubles + two NB x (NB+1) blocks must fit in 160 KiB LDS

typedef double ch_double4 __attribute__((ext_vector_type(4)));

constexpr int CH_G = 4;              // tiles of the trailing update per wave and trip
__device__ double ch_sink[CH_T];     // where the stores of lanes outside the lower triangle go (never read)

__device__ __forceinline__ double ch_rdlane(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l),
                            __builtin_amdgcn_readlane(__double2loint(v), l));
}

// A (n x n, row-major, ld) is overwritten by L in its lower triangle (upper triangle untouched).
// shift is added to the diagonal on the fly (the jitter rung), so the caller's matrix can be probed
// at several rungs from a scratch copy.
// Batched form (gridDim.x > 1): workgroup b factorises (src + shifts[b] I) in its own n x n slab of
// `A` (slab stride n*ld doubles) and reports info[b]; used to probe every rung of the jitter ladder of
// SOBER/_utils.py:145-156 in ONE launch (the rungs are independent).
//
// Right-looking, panel width 32, per panel:
//   (b) wave 0 factorises the 32 x 32 diagonal block IN REGISTERS (lane = row; pivots and multipliers are
//       v_readlane broadcasts) and inverts it (lane = column of the inverse);
//   (c) the panel below is L21 = A21 L11^-T as a small GEMM on the matrix cores (no per-row substitution chain);
//   (d) the trailing update A22 -= L21 L21^T runs on the matrix cores from the LDS-resident panel, one 16 x 16
//       tile per wave and trip, the next tile of A22 already in flight.
// (b) of k_chol, by ONE wave: Cholesky of the 32 x 32 diagonal block D (LDS, lower part, zero padded) in registers
// (lane = row; pivots and multipliers are v_readlane broadcasts), L11 back to D, its inverse to Xs (and to xo).
__device__ __forceinline__ void ch_diag_block(double* __restrict__ D, double* __restrict__ Xs, double* __restrict__ s_dinv,
                                              int* __restrict__ s_fail_p, double* __restrict__ s_minp_p, int nb, int kb,
                                              int lane, double* __restrict__ xout) {
    constexpr int LDP = CH_NB + 1;
    int& s_fail = *s_fail_p;
    double& s_minp = *s_minp_p;

            const int i = lane;
            double d[CH_NB];
#pragma unroll
            for (int c = 0; c < CH_NB; ++c) d[c] = (i < CH_NB) ? D[min(i, CH_NB - 1) * LDP + c] : 0.0;
            int fail = 0;
            double minp = s_minp;
            double my_rinv = 1.0;                         // lane j: 1 / L[j][j] (identity padding: 1)
#pragma unroll
            for (int j = 0; j < CH_NB; ++j) {
                if (j < nb && fail == 0) {                // uniform
                    const double djj = ch_rdlane(d[j], j);
                    if (!(djj > 0.0)) {                   // also catches NaN
                        fail = kb + j + 1;
                        minp = fmin(minp, djj);           // the failing pivot (<= 0): how far from positive definite
                    } else {
                        // multipliers through the reciprocal square root: one short dependent sequence per
                        // column instead of sqrt followed by a division (off-diagonal entries within 2 ulp)
                        // sqrt and 1/sqrt from ONE v_rsq_f64 seed + Newton steps (<= 1 ulp): the IEEE sqrt() and
                        // rsqrt() sequences are ~60 dependent instructions on the critical path of every column
                        double rinv = __builtin_amdgcn_rsq(djj);
                        {
                            double h = 0.5 * rinv, e = fma(-(djj * rinv), h, 0.5);
                            rinv = fma(rinv, e, rinv);
                            h = 0.5 * rinv; e = fma(-(djj * rinv), h, 0.5);
                            rinv = fma(rinv, e, rinv);
                        }
                        double l = djj * rinv;
                        l = fma(fma(-l, l, djj), 0.5 * rinv, l);
                        minp = fmin(minp, djj);
                        my_rinv = (i == j) ? rinv : my_rinv;
                        d[j] = (i == j) ? l : ((i > j) ? d[j] * rinv : 0.0);
#pragma unroll
                        for (int c = j + 1; c < CH_NB; ++c)
                            d[c] = fma(-d[j], ch_rdlane(d[j], c), d[c]);   // rows i < c: unused upper-triangle values
                    }
                }
            }
            if (i == 0) { s_fail = fail; s_minp = minp; }
            if (fail == 0) {
                // L11 -> LDS (padding rows >= nb become identity rows so that the inverse exists)
                if (i < CH_NB) {
#pragma unroll
                    for (int c = 0; c < CH_NB; ++c)
                        D[i * LDP + c] = (i < nb) ? ((c <= i) ? d[c] : 0.0) : ((c == i) ? 1.0 : 0.0);
                    s_dinv[i] = my_rinv;
                }
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // X = L11^-1 by columns, lane = column jx of X, column-oriented substitution: once x[k] is final,
                // the updates of the rows below are independent of one another; L[r][k] and 1/L[k][k] are
                // wave-uniform LDS reads (broadcasts) that do not depend on x
                const int jx = lane;
                double x[CH_NB];
#pragma unroll
                for (int r = 0; r < CH_NB; ++r) x[r] = (r == jx) ? 1.0 : 0.0;
#pragma unroll
                for (int k = 0; k < CH_NB; ++k) {
                    x[k] *= s_dinv[k];
#pragma unroll
                    for (int r = k + 1; r < CH_NB; ++r) x[r] = fma(-D[r * LDP + k], x[k], x[r]);
                }
                if (i < CH_NB) {
#pragma unroll
                    for (int c = 0; c < CH_NB; ++c) Xs[c * LDP + jx] = x[c];                 // X[c][jx]
                    if (xout != nullptr) {                    // the inverted diagonal blocks feed k_trsm_blocks
                        double* xo = xout + (size_t)(kb / CH_NB) * CH_NB * CH_NB;
#pragma unroll
                        for (int c = 0; c < CH_NB; ++c) xo[c * CH_NB + jx] = x[c];
                    }
                }
            }
        }

// SMALL (n <= CH_SMALLN, one matrix): the matrix itself is staged into LDS and every phase works there -- the q x q Gram
// matrices of CholeskyQR (q = batch - 1 ~ 100) spent most of their 86 us per call on global-memory round trips between
// the phases of four panels, not on arithmetic.
constexpr int CH_SMALLN = 120;      // 2 x 32 x 33 + (n - 32) x 36 + n (n + 1) doubles must fit in 160 KiB - 512
template <bool SMALL>
__global__ __launch_bounds__(CH_T) void k_chol(double* __restrict__ A, int n, int ld, double shift,
                                              int32_t* __restrict__ info, double* __restrict__ min_pivot,
                                              const double* __restrict__ src, int lds_src,
                                              const double* __restrict__ shifts, double* __restrict__ xout,
                                              double* __restrict__ ratio_out) {
    extern __shared__ double lds[];
    if (src != nullptr) {                     // batched: copy the lower triangle into my slab first
        A += (size_t)blockIdx.x * n * ld;
        info += blockIdx.x;
        if (min_pivot) min_pivot += blockIdx.x;
        shift = shifts[blockIdx.x];
        for (int i = threadIdx.x >> 6; i < n; i += CH_T / 64)
            for (int j = threadIdx.x & 63; j <= i; j += 64) A[(size_t)i * ld + j] = src[(size_t)i * lds_src + j];
        __threadfence_block();
        __syncthreads();
    }
    constexpr int LDP = CH_NB + 1;
    double* D = lds;                         // NB x (NB+1): diagonal block / L11
    double* Xs = lds + CH_NB * LDP;          // NB x (NB+1): L11^-1
    constexpr int LDPP = CH_LDPP;
    double* P = lds + 2 * CH_NB * LDP;       // (n - kb - nb) x LDPP panel
    double* const A_glob = A;
    const int ld_glob = ld;
    if constexpr (SMALL) {                   // lower triangle -> LDS (odd row stride), and work there from here on
        double* As = P + (size_t)(n > CH_NB ? n - CH_NB : 0) * LDPP;
        for (int i = threadIdx.x >> 6; i < n; i += CH_T / 64)
            for (int j = threadIdx.x & 63; j <= i; j += 64) As[i * (n + 1) + j] = A_glob[(size_t)i * ld_glob + j];
        A = As;
        ld = n + 1;
    }
    __shared__ int s_fail;
    __shared__ double s_minp;
    __shared__ double s_dinv[CH_NB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    if (tid == 0) { s_fail = 0; s_minp = __builtin_inf(); }
    // largest diagonal entry of the input (ratio_out = smallest pivot / this: ~cond^-2 for a Gram matrix)
    __shared__ double s_dm[CH_T / 64];
    if (ratio_out != nullptr) {
        double dl = -__builtin_inf();
        for (int i = tid; i < n; i += CH_T) dl = fmax(dl, A_glob[(size_t)i * ld_glob + i] + shift);   // (the caller's matrix: the LDS copy is not synchronised yet)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) dl = fmax(dl, __shfl_xor(dl, o, 64));
        if (lane == 0) s_dm[wave] = dl;
    }
    __syncthreads();
#ifdef CH_STAMPS
    long long st_t[5] = {0, 0, 0, 0, 0}, st_last = wall_clock64();
#define CH_STAMP(K) { const long long now_ = wall_clock64(); st_t[K] += now_ - st_last; st_last = now_; }
#else
#define CH_STAMP(K)
#endif

    bool ahead = false;                                   // this panel's diagonal block was factorised by the previous (d)
    for (int kb = 0; kb < n; kb += CH_NB) {
        const int nb = min(CH_NB, n - kb);
        const int nr = n - kb - nb;                       // rows below the diagonal block
        if (!ahead) {
            // ---- (a) diagonal block -> LDS (lower part), + shift; zero padded to 32 x 32
#pragma unroll
            for (int t = tid; t < CH_NB * CH_NB; t += CH_T) {
                const int i = t >> 5, j = t & 31;
                double v = (i < nb && j <= i) ? A[(size_t)(kb + i) * ld + kb + j] : 0.0;
                if (i == j && i < nb) v += shift;
                D[i * LDP + j] = v;
            }
            __syncthreads();
            CH_STAMP(0)
            // ---- (b) wave 0: Cholesky of D in registers, then its inverse
            if (wave == 0) ch_diag_block(D, Xs, s_dinv, &s_fail, &s_minp, nb, kb, lane, xout);
            __syncthreads();
            CH_STAMP(1)
        }
        if (s_fail != 0) break;                            // uniform
        // ---- write L_kk back
#pragma unroll
        for (int t = tid; t < CH_NB * CH_NB; t += CH_T) {
            const int i = t >> 5, j = t & 31;
            if (i < nb && j <= i) A[(size_t)(kb + i) * ld + kb + j] = D[i * LDP + j];
        }
        if (nr == 0) break;
        // ---- (c) panel: L21 = A21 X^T on the matrix cores, one 16 x 32 row block per wave and trip (A21 rows are
        // read once, the next block's rows are in flight during the MFMAs); result kept in LDS
        {
            const int ntile = (nr + 15) / 16;
            double af[8], an[8];
            if (wave < ntile) {
                const double* arow = A + (size_t)(kb + nb + min(wave * 16 + li, nr - 1)) * ld + kb;
#pragma unroll
                for (int u = 0; u < 8; ++u) af[u] = arow[min(4 * u + lk, nb - 1)];
            }
            for (int t = wave; t < ntile; t += CH_T / 64) {
                const int r0 = t * 16;
                {
                    const int tn = min(t + CH_T / 64, ntile - 1);
                    const double* arow = A + (size_t)(kb + nb + min(tn * 16 + li, nr - 1)) * ld + kb;
#pragma unroll
                    for (int u = 0; u < 8; ++u) an[u] = arow[min(4 * u + lk, nb - 1)];
                }
                ch_double4 acc0 = (ch_double4){0.0, 0.0, 0.0, 0.0}, acc1 = acc0;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int k = 4 * u + lk;
                    const double a = (k < nb) ? af[u] : 0.0;
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Xs[li * LDP + k], acc0, 0, 0, 0);   // B[k][j] = X[j][k]
                    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Xs[(16 + li) * LDP + k], acc1, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = r0 + lk + 4 * r;
                    if (row < nr) {
                        P[row * LDPP + li] = acc0[r];
                        P[row * LDPP + 16 + li] = acc1[r];
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) af[u] = an[u];
            }
        }
        __syncthreads();
        CH_STAMP(2)
        // L21 goes back to A from LDS
        for (int r = tid >> 5; r < nr; r += CH_T / 32) {
            const int j = tid & 31;
            if (j < nb) A[(size_t)(kb + nb + r) * ld + kb + j] = P[r * LDPP + j];
        }
        // ---- (d) trailing update (lower triangle): A22[r][c] -= P[r] . P[c], 16 x 16 tiles on the matrix cores.
        // A wave walks its tiles (linear index wave, wave + 8, ...) in batches of CH_G; the whole loop body is
        // straight-line code -- loads of the next batch, MFMAs, then stores that are redirected to a sink instead
        // of being predicated -- so that the s_waitcnt in front of a batch's first use counts exactly and the
        // stores just issued are not waited for (loads and stores share vmcnt on gfx950).
        {
            const int nt = (nr + 15) / 16;                 // tiles per side
            // LOOK-AHEAD: wave 0 takes the (at most three) tiles of the NEXT diagonal block -- linear indices 0, 1, 2 --
            // and then factorises that block (the ~13 us of (b): one wave's dependent chain) while waves 1..7 walk the
            // rest of the update (linear index 3 + (wave - 1), + 7, ...); with fewer than 8 tiles there is nothing to hide
            // behind and everybody walks as before.
            const int wv = __builtin_amdgcn_readfirstlane(wave);
            const bool look = nt * (nt + 1) / 2 >= 8;
            const int t_step = look ? (wv == 0 ? 1 : CH_T / 64 - 1) : CH_T / 64;
            const int nt_mine = (look && wv == 0) ? min(nt, 2) : nt;       // wave 0 stops after tile row 1
            int ntr = 0, ntc = look ? (wv == 0 ? 0 : 2 + wv) : wv;         // cursor over this wave's tiles
            while (ntc > ntr) { ntc -= ntr + 1; ++ntr; }
            int tr[CH_G], tc[CH_G], trn[CH_G], tcn[CH_G];
            double cv[CH_G][4], cn[CH_G][4];
            double* const a22 = A + (size_t)(kb + nb) * ld + kb + nb;
#define CH_TAKE(R, C)                                                       \
            { R = ntr; C = ntc; ntc += t_step; while (ntc > ntr) { ntc -= ntr + 1; ++ntr; } }
#define CH_LOAD(DST, R, C)                                                  \
            _Pragma("unroll") for (int e = 0; e < 4; ++e)                   \
                DST[e] = a22[(size_t)min((R) * 16 + lk + 4 * e, nr - 1) * ld + min((C) * 16 + li, nr - 1)];
#pragma unroll
            for (int g = 0; g < CH_G; ++g) { CH_TAKE(tr[g], tc[g]) CH_LOAD(cv[g], tr[g], tc[g]) }
            while (tr[0] < nt_mine) {
#pragma unroll
                for (int g = 0; g < CH_G; ++g) { CH_TAKE(trn[g], tcn[g]) CH_LOAD(cn[g], trn[g], tcn[g]) }
                ch_double4 acc[CH_G][2];
#pragma unroll
                for (int g = 0; g < CH_G; ++g) {
                    acc[g][0] = (ch_double4){0.0, 0.0, 0.0, 0.0};
                    acc[g][1] = acc[g][0];
                    const int ra = min(tr[g] * 16 + li, nr - 1), rb = min(tc[g] * 16 + li, nr - 1);
#pragma unroll
                    for (int u = 0; u < 8; u += 2) {
                        acc[g][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(P[ra * LDPP + 4 * u + lk],
                                                                         P[rb * LDPP + 4 * u + lk], acc[g][0], 0, 0, 0);
                        acc[g][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(P[ra * LDPP + 4 * u + 4 + lk],
                                                                         P[rb * LDPP + 4 * u + 4 + lk], acc[g][1], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int g = 0; g < CH_G; ++g) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int row = tr[g] * 16 + lk + 4 * e, col = tc[g] * 16 + li;
                        double* dst = (row < nr && col <= row && tr[g] < nt_mine) ? a22 + (size_t)row * ld + col : ch_sink + tid;
                        *dst = cv[g][e] - (acc[g][0][e] + acc[g][1][e]);
                    }
                }
#pragma unroll
                for (int g = 0; g < CH_G; ++g) {
                    tr[g] = trn[g]; tc[g] = tcn[g];
#pragma unroll
                    for (int e = 0; e < 4; ++e) cv[g][e] = cn[g][e];
                }
            }
#undef CH_TAKE
#undef CH_LOAD
            if (look && wave == 0) {
                // the next diagonal block is final (my own stores: wait for them), so (a) and (b) of the next panel
                // happen here, beside the other waves' share of the update
                __builtin_amdgcn_s_waitcnt(0);
                const int kb2 = kb + nb, nb2 = min(CH_NB, n - kb2);
#pragma unroll
                for (int t = lane; t < CH_NB * CH_NB; t += 64) {
                    const int i = t >> 5, j = t & 31;
                    double v = (i < nb2 && j <= i) ? A[(size_t)(kb2 + i) * ld + kb2 + j] : 0.0;
                    if (i == j && i < nb2) v += shift;
                    D[i * LDP + j] = v;
                }
                __builtin_amdgcn_wave_barrier();
                ch_diag_block(D, Xs, s_dinv, &s_fail, &s_minp, nb2, kb2, lane, xout);
            }
            ahead = look;
        }
        __threadfence_block();
        __syncthreads();
        CH_STAMP(3)
    }
#ifdef CH_STAMPS
    if (tid == 0 && n >= 16)                               // debugging build: ticks (100 MHz) in row 0's upper triangle
        for (int k = 0; k < 4; ++k) A[8 + k] = (double)st_t[k];
#endif
    if constexpr (SMALL) {                   // L back to the caller's matrix (rows finished before a failure included)
        __syncthreads();
        for (int i = tid >> 6; i < n; i += CH_T / 64)
            for (int j = tid & 63; j <= i; j += 64) A_glob[(size_t)i * ld_glob + j] = A[i * (n + 1) + j];
    }
    if (tid == 0) {
        *info = s_fail;
        if (min_pivot) *min_pivot = s_minp;
        if (ratio_out) {
            double dm = s_dm[0];
#pragma unroll
            for (int w = 1; w < CH_T / 64; ++w) dm = fmax(dm, s_dm[w]);
            *ratio_out = s_minp / dm;
        }
    }
}

// Q = Y L^-T (the Q factor of CholeskyQR) on the matrix cores, from the INVERTED 32 x 32 diagonal blocks that
// k_chol leaves behind (xinv): for column block jb
//     Q_jb = (Y_jb - sum_{kb < jb} Q_kb L[jb][kb]^T) X_jb^T ,      X_jb = L[jb][jb]^-1 ,
// so the only sequential dependency is block to block (q / 32 steps), not column to column.  One wave per 16 rows
// of Y; its finished Q blocks stay in LDS in the A-fragment-friendly layout for the later blocks.
constexpr int TB_WAVES = 2;
__global__ __launch_bounds__(TB_WAVES * 64) void k_trsm_blocks(const double* __restrict__ Y, int64_t m, int q, int ldy,
                                                              const double* __restrict__ L, int ldl,
                                                              const double* __restrict__ Xinv,
                                                              double* __restrict__ Q, int ldq) {
    extern __shared__ double tb_lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int qpad = ((q + 31) / 32) * 32, LS = qpad + 4;             // stash row stride: fragments on distinct bank pairs
    double* stash = tb_lds + (size_t)wave * (16 * LS + 16 * 36);      // 16 x LS: the Q blocks finished so far
    double* tbuf = stash + 16 * LS;                                   // 16 x 36: the block being multiplied by X^T
    const int64_t R0 = ((int64_t)blockIdx.x * TB_WAVES + wave) * 16;
    if (R0 >= m) return;                                              // whole wave; no workgroup barrier below
    const int nblk = qpad / 32;
    for (int jb = 0; jb < nblk; ++jb) {
        const int c0 = 32 * jb;
        ch_double4 acc[2];
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = min(R0 + lk + 4 * r, m - 1);
                const int col = c0 + 16 * jt + li;
                const double v = Y[row * ldy + min(col, q - 1)];
                acc[jt][r] = (col < q) ? v : 0.0;
            }
        for (int kb = 0; kb < jb; ++kb) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int kc = 32 * kb + 4 * u + lk;                  // kc < c0 <= q - 1... (kb < jb: a full block)
                const double a = -stash[li * LS + kc];
#pragma unroll
                for (int jt = 0; jt < 2; ++jt) {
                    const int lr = c0 + 16 * jt + li;                 // row of L = column of this block
                    const double b = L[(size_t)min(lr, q - 1) * ldl + kc];
                    acc[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, (lr < q) ? b : 0.0, acc[jt], 0, 0, 0);
                }
            }
        }
        // T = acc -> LDS (row-major 16 x 32), back as A fragments: out = T X_jb^T
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) tbuf[(lk + 4 * r) * 36 + 16 * jt + li] = acc[jt][r];
        __builtin_amdgcn_wave_barrier();
        const double* X = Xinv + (size_t)jb * 32 * 32;
        ch_double4 out[2] = {(ch_double4){0.0, 0.0, 0.0, 0.0}, (ch_double4){0.0, 0.0, 0.0, 0.0}};
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const double a = tbuf[li * 36 + 4 * u + lk];
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
                out[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, X[(16 * jt + li) * 32 + 4 * u + lk], out[jt], 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = R0 + lk + 4 * r;
                const int col = c0 + 16 * jt + li;
                stash[(lk + 4 * r) * LS + col] = out[jt][r];
                if (row < m && col < q) Q[row * ldq + col] = out[jt][r];
            }
        __builtin_amdgcn_wave_barrier();
    }
}

// cov -> sqrt(nan_to_num(cov) * nan_to_num(cov).T) (SOBER/_utils.py:143-144) and the exact-symmetry
// test of :127 on the input: flag[0] |= 1 if some cov[i][j] != cov[j][i].
__global__ __launch_bounds__(256) void k_abs_sym(const double* __restrict__ C, int n, int ld,
                                                 double* __restrict__ out, int ldo, int32_t* __restrict__ flag) {
    // 32 x 32 tile and its mirror image through LDS: both reads are row-contiguous
    __shared__ double tm[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    for (int r = ty; r < 32; r += 8) {
        const int i = j0 + r, j = i0 + tx;                            // mirror tile element C[j0 + r][i0 + tx]
        tm[r][tx] = (i < n && j < n) ? C[(size_t)i * ld + j] : 0.0;
    }
    __syncthreads();
    bool asym = false;
    const double big = 1.7976931348623157e308;
    for (int r = ty; r < 32; r += 8) {
        const int i = i0 + r, j = j0 + tx;
        if (i < n && j < n) {
            double a = C[(size_t)i * ld + j], b = tm[tx][r];          // C[j][i]
            asym |= !(a == b);
            a = (a != a) ? 0.0 : fmin(fmax(a, -big), big);            // torch.nan_to_num
            b = (b != b) ? 0.0 : fmin(fmax(b, -big), big);
            out[(size_t)i * ldo + j] = sqrt(a * b);
        }
    }
    if (__syncthreads_or(asym) && threadIdx.x == 0) atomicOr(flag, 1);
}

// the jitter ladder of make_cov_psd (SOBER/_utils.py:151-152), k rungs at once with the reference's own
// sequence of roundings: jitter = 1e-5; repeat k times { diag += jitter; jitter *= 2 }
__global__ void k_jitter_ladder(double* __restrict__ A, int n, int ld, int k) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double d = A[(size_t)i * ld + i], jit = 1e-5;
    for (int t = 0; t < k; ++t) { d = __dadd_rn(d, jit); jit = __dmul_rn(jit, 2.0); }
    A[(size_t)i * ld + i] = d;
}

// the same ladder with the rung chosen ON THE DEVICE from the probe's verdicts: k = index of the first rung with
// info == 0; when none of the n_rungs is positive definite, all n_rungs additions are made and the matrix is
// reduced to its diagonal -- `cov = cov.diag().diag()` of SOBER/_utils.py:153-156 (n_iter > max_iter).
// *k_out = k.  No host decision between the probe and the range finder.
__global__ void k_jitter_ladder_auto(double* __restrict__ A, int n, int ld, const int32_t* __restrict__ info,
                                     int n_rungs, int32_t* __restrict__ k_out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    int k = 0;
    while (k < n_rungs && info[k] != 0) ++k;               // (uniform)
    if (i == 0 && j == 0) *k_out = k;
    if (j >= n) return;
    if (i == j) {
        double d = A[(size_t)i * ld + i], jit = 1e-5;
        for (int t = 0; t < k; ++t) { d = __dadd_rn(d, jit); jit = __dmul_rn(jit, 2.0); }
        A[(size_t)i * ld + i] = d;
    } else if (k == n_rungs) {
        A[(size_t)i * ld + j] = 0.0;
    }
}

}  // namespace sober

extern "C" int sober_chol_max_n(void) { return sober::CH_MAXN; }

extern "C" int sober_cholesky_inv_ratio(double* A, int n, int ld, double shift, int32_t* info, double* min_pivot,
                                        double* xinv, double* ratio_out, void* stream);
extern "C" int sober_cholesky_inv(double* A, int n, int ld, double shift, int32_t* info, double* min_pivot,
                                  double* xinv, void* stream) {
    return sober_cholesky_inv_ratio(A, n, ld, shift, info, min_pivot, xinv, nullptr, stream);
}

extern "C" int sober_cholesky_inv_ratio(double* A, int n, int ld, double shift, int32_t* info, double* min_pivot,
                                        double* xinv, double* ratio_out, void* stream) {
    if (!A || !info || n <= 0 || ld < n) return SOBER_E_ARG;
    if (n > sober::CH_MAXN) return SOBER_E_DIM;
    const int nr = n > sober::CH_NB ? n - sober::CH_NB : 0;
    size_t bytes = ((size_t)2 * sober::CH_NB * (sober::CH_NB + 1) + (size_t)nr * sober::CH_LDPP) * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void*)sober::k_chol<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    160 * 1024 - 512));
        HIP_TRY(hipFuncSetAttribute((const void*)sober::k_chol<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    160 * 1024 - 512));
        attr_set = true;
    }
    if (n <= sober::CH_SMALLN) {
        bytes += (size_t)n * (n + 1) * sizeof(double);
        hipLaunchKernelGGL(sober::k_chol<true>, dim3(1), dim3(sober::CH_T), bytes, (hipStream_t)stream, A, n, ld, shift,
                           info, min_pivot, (const double*)nullptr, 0, (const double*)nullptr, xinv, ratio_out);
    } else {
        hipLaunchKernelGGL(sober::k_chol<false>, dim3(1), dim3(sober::CH_T), bytes, (hipStream_t)stream, A, n, ld, shift,
                           info, min_pivot, (const double*)nullptr, 0, (const double*)nullptr, xinv, ratio_out);
    }
    LAUNCH_CHECK();
    return 0;
}

extern "C" int sober_cholesky(double* A, int n, int ld, double shift, int32_t* info, double* min_pivot,
                              void* stream) {
    return sober_cholesky_inv(A, n, ld, shift, info, min_pivot, nullptr, stream);
}

extern "C" int sober_trsm_blocks(const double* Y, int64_t m, int q, int ldy, const double* L, int ldl,
                                 const double* Xinv, double* Q, int ldq, void* stream) {
    if (!Y || !L || !Xinv || !Q || m <= 0 || q <= 0 || q > 256 || ldy < q || ldl < q || ldq < q) return SOBER_E_ARG;
    const int qpad = ((q + 31) / 32) * 32;
    const size_t bytes = (size_t)sober::TB_WAVES * (16 * (qpad + 4) + 16 * 36) * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void*)sober::k_trsm_blocks, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    120 * 1024));
        attr_set = true;
    }
    const int64_t row_blocks = (m + 15) / 16;
    hipLaunchKernelGGL(sober::k_trsm_blocks, dim3((unsigned)((row_blocks + sober::TB_WAVES - 1) / sober::TB_WAVES)),
                       dim3(sober::TB_WAVES * 64), bytes, (hipStream_t)stream, Y, m, q, ldy, L, ldl, Xinv, Q, ldq);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int sober_cholesky_probe_piv(const double* src, int n, int ld_src, const double* shifts, int n_shifts,
                                        double* work, int32_t* info, double* min_pivot, void* stream);
extern "C" int sober_cholesky_probe(const double* src, int n, int ld_src, const double* shifts, int n_shifts,
                                    double* work, int32_t* info, void* stream) {
    return sober_cholesky_probe_piv(src, n, ld_src, shifts, n_shifts, work, info, nullptr, stream);
}

extern "C" int sober_cholesky_probe_piv(const double* src, int n, int ld_src, const double* shifts, int n_shifts,
                                        double* work, int32_t* info, double* min_pivot, void* stream) {
    if (!src || !shifts || !work || !info || n <= 0 || ld_src < n || n_shifts <= 0) return SOBER_E_ARG;
    if (n > sober::CH_MAXN) return SOBER_E_DIM;
    const int nr = n > sober::CH_NB ? n - sober::CH_NB : 0;
    const size_t bytes = ((size_t)2 * sober::CH_NB * (sober::CH_NB + 1) + (size_t)nr * sober::CH_LDPP) * sizeof(double);
    HIP_TRY(hipFuncSetAttribute((const void*)sober::k_chol<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024 - 512));
    hipLaunchKernelGGL(sober::k_chol<false>, dim3(n_shifts), dim3(sober::CH_T), bytes, (hipStream_t)stream, work, n, n, 0.0,
                       info, min_pivot, src, ld_src, shifts, (double*)nullptr, (double*)nullptr);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int sober_abs_sym(const double* C, int n, int ld, double* out, int ldo, int32_t* flag, void* stream) {
    if (!C || !out || !flag || n <= 0 || ld < n || ldo < n || n > 65535) return SOBER_E_ARG;
    hipLaunchKernelGGL(sober::k_abs_sym, dim3((n + 31) / 32, (n + 31) / 32), dim3(256), 0, (hipStream_t)stream, C, n,
                       ld, out, ldo, flag);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int sober_jitter_ladder(double* A, int n, int ld, int k, void* stream) {
    if (!A || n <= 0 || ld < n || k < 0) return SOBER_E_ARG;
    if (k == 0) return 0;
    hipLaunchKernelGGL(sober::k_jitter_ladder, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, A, n, ld, k);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int sober_jitter_ladder_auto(double* A, int n, int ld, const int32_t* info, int n_rungs, int32_t* k_out,
                                        void* stream) {
    if (!A || !info || !k_out || n <= 0 || ld < n || n_rungs <= 0) return SOBER_E_ARG;
    hipLaunchKernelGGL(sober::k_jitter_ladder_auto, dim3((n + 255) / 256, n), dim3(256), 0, (hipStream_t)stream, A, n,
                       ld, info, n_rungs, k_out);
    LAUNCH_CHECK();
    return 0;
}
