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
#include "internal.hpp"

namespace sober {

constexpr int CH_NB = 32;            // panel width
constexpr int CH_T = 512;             // 8 waves: 256 VGPRs per lane for the register-resident diagonal block
constexpr int CH_LDPP = CH_NB + 4;   // row stride of the LDS panel: 16 rows x 4 k of an MFMA fragment fall on distinct bank pairs
constexpr int CH_MAXN = 536;         // panel (n - NB) x LDPP doubles + two NB x (NB+1) blocks must fit in 160 KiB LDS

typedef double ch_double4 __attribute__((ext_vector_type(4)));

constexpr int CH_G = 4;              // tiles of the trailing update per wave and trip
__device__ double ch_sink[CH_T];     // where the stores of lanes outside the lower triangle go (never read)
#ifdef CM_STAMPS
__device__ double ch_dbg[2];        // diagnostic build: ticks of the last ch_diag_block (factorisation, inversion)
#endif     // where the stores of lanes outside the lower triangle go (never read)

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
//   (b) wave 0 factorises the 32 x 32 diagonal block IN REGISTERS and builds its inverse in the same pass (lanes 0..31:
//       rows of L, lanes 32..63: columns of the inverse; pivots and multipliers are v_readlane broadcasts);
//   (c) the panel below is L21 = A21 L11^-T as a small GEMM on the matrix cores (no per-row substitution chain);
//   (d) the trailing update A22 -= L21 L21^T runs on the matrix cores from the LDS-resident panel, one 16 x 16
//       tile per wave and trip, the next tile of A22 already in flight.
// (b) of k_chol, by ONE wave: Cholesky of the 32 x 32 diagonal block D (LDS, lower part, zero padded) in registers,
// L11 back to D, its inverse to Xs (and to xout).  s_dinv is unused (kept for the call sites' signature).
template <int NC>      // NC = 8 / 16 / 32 columns are factorised (nb <= NC); the rest of the block is identity
__device__ __attribute__((noinline)) void ch_diag_block_n(double* __restrict__ D, double* __restrict__ Xs, double* __restrict__ s_dinv,
                                              int* __restrict__ s_fail_p, double* __restrict__ s_minp_p, int nb, int kb,
                                              int lane, double* __restrict__ xout) {
    constexpr int LDP = CH_NB + 1;
    int& s_fail = *s_fail_p;
    double& s_minp = *s_minp_p;

            // One straight-line block (no branch per column): the factorisation's column j and step j of the
            // column-oriented substitution for X = L^-1 (lane = column of X) use the SAME broadcast multipliers
            // L[c][j], and the scheduler can put the independent updates of column j beside the dependent
            // rsq / Newton sequence of column j + 1.  (Measured before: factorisation with a branch per column 7.8 us,
            // then the substitution with its multipliers read back from LDS 7.0 us -- the whole launch's critical path.)
            // A failed pivot only records itself; what is computed after it is never used.  Rows and columns >= nb are
            // an identity block (pivots 1, not counted).
            // The two halves of the wave split the work: lane i < 32 holds row i of the block being factorised
            // (v = L's row), lane 32 + i holds column i of the inverse being built (v = X's column).  Both obey the SAME
            // update v[c] -= v[j] L[c][j] with the same broadcast multipliers, so one fma per (j, c) serves both;
            // only the scaling of column j differs (a select).  Column j of L and row j of X leave the registers as soon
            // as they are final (every lane writes, none is masked off).
            const int i = lane & (CH_NB - 1);
            const bool xh = lane >= CH_NB;                   // the inverse's half
            double v[CH_NB];
#ifdef CM_STAMPS
            const long long dbg0 = wall_clock64();
#endif
#pragma unroll
            for (int c = 0; c < CH_NB; ++c) {
                double dv = D[i * LDP + c];                   // (rows >= nb are zero padded: read, then replaced)
                asm volatile("" : "+v"(dv));                  // (an unconditional load: no branch around it per column)
                v[c] = (xh | (i >= nb)) ? ((c == i) ? 1.0 : 0.0) : dv;      // identity: X's start, and the padding rows
            }
            const bool xw = xout != nullptr && xh;           // this lane's X entries also go to the caller's xout
            double* xo = xw ? xout + (size_t)(kb / CH_NB) * CH_NB * CH_NB : ch_sink;
            const int xo_ld = xw ? CH_NB : 0, xo_i = xw ? i : lane;
            // where a final entry goes: L[i][j] -> D[i][j]; X[j][i] -> Xs[j][i]  (Xs = D + NB * LDP by construction)
            double* const o_base = xh ? Xs + i : D + i * LDP;
            const int o_step = xh ? LDP : 1;
            int fail = 0;
            double minp = s_minp;
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                const double djj = ch_rdlane(v[j], j);
                const bool live = (j < nb) & (fail == 0);     // uniform
                minp = live ? fmin(minp, djj) : minp;         // (the failing pivot, <= 0, included: how far from positive definite)
                fail = (live & !(djj > 0.0)) ? kb + j + 1 : fail;       // also catches NaN
                // sqrt and 1/sqrt from ONE v_rsq_f64 seed + Newton steps (<= 1 ulp): the IEEE sqrt() and rsqrt()
                // sequences are ~60 dependent instructions on the critical path of every column; multipliers through
                // the reciprocal square root (off-diagonal entries within 2 ulp)
                double rinv = __builtin_amdgcn_rsq(djj);
                {
                    double h = 0.5 * rinv, e = fma(-(djj * rinv), h, 0.5);
                    rinv = fma(rinv, e, rinv);
                    h = 0.5 * rinv; e = fma(-(djj * rinv), h, 0.5);
                    rinv = fma(rinv, e, rinv);
                }
                double l = djj * rinv;
                l = fma(fma(-l, l, djj), 0.5 * rinv, l);
                const double sc = v[j] * rinv;                // L[i][j] = a / L[j][j];  x[j] / L[j][j]
                v[j] = xh ? sc : ((i == j) ? l : ((i > j) ? sc : 0.0));
                o_base[j * o_step] = v[j];
                xo[j * xo_ld + xo_i] = v[j];                  // (the inverted diagonal blocks feed k_trsm_blocks)
                // (the multipliers L[c][j] read back from the column just written to LDS -- one wave-uniform ds_read per
                //  two of them instead of two v_readlane each -- measured 9.3 us against 8.2 us for this form)
#pragma unroll
                for (int c = j + 1; c < NC; ++c) {            // (columns >= NC >= nb: identity, their multipliers are zero)
                    const double lcj = ch_rdlane(v[j], c);    // L[c][j]  (lane c < 32: the factor's half)
                    v[c] = fma(-v[j], lcj, v[c]);             // rows i < c: unused upper-triangle values
                }
            }
#pragma unroll
            for (int j = NC; j < CH_NB; ++j) { o_base[j * o_step] = v[j]; xo[j * xo_ld + xo_i] = v[j]; }   // the identity part
            if (lane == 0) { s_fail = fail; s_minp = minp; }
#ifdef CM_STAMPS
            if (lane == 0) { ch_dbg[0] = (double)(wall_clock64() - dbg0); ch_dbg[1] = 0.0; }
#endif
        }

// a ragged last block (nb = 3 at q = 99, 7 at q = 199) pays for the columns it has, rounded up to 8 or 16
__device__ __forceinline__ void ch_diag_block(double* __restrict__ D, double* __restrict__ Xs, double* __restrict__ s_dinv,
                                              int* __restrict__ s_fail_p, double* __restrict__ s_minp_p, int nb, int kb,
                                              int lane, double* __restrict__ xout) {
    if (nb <= 8) ch_diag_block_n<8>(D, Xs, s_dinv, s_fail_p, s_minp_p, nb, kb, lane, xout);
    else if (nb <= 16) ch_diag_block_n<16>(D, Xs, s_dinv, s_fail_p, s_minp_p, nb, kb, lane, xout);
    else ch_diag_block_n<CH_NB>(D, Xs, s_dinv, s_fail_p, s_minp_p, nb, kb, lane, xout);
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
    // SMALL: the lower triangle goes to LDS (odd row stride) and the work happens there.  Round 6: the first diagonal block is
    // read straight from the caller's matrix and wave 0 factorises it WHILE waves 1-7 stage the triangle (the staging -- a
    // dozen dependent global loads per wave, ~6 us -- used to stand in front of the first block's chain)
    double* As = nullptr;
    if constexpr (SMALL) {
        As = P + (size_t)(n > CH_NB ? n - CH_NB : 0) * LDPP;
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
    bool ahead0 = false;
    if constexpr (SMALL) {
        const int nb0 = min(CH_NB, n);
#pragma unroll
        for (int t = tid; t < CH_NB * CH_NB; t += CH_T) {
            const int i = t >> 5, j = t & 31;
            double v = (i < nb0 && j <= i) ? A_glob[(size_t)i * ld_glob + j] : 0.0;
            if (i == j && i < nb0) v += shift;
            D[i * LDP + j] = v;
        }
        __syncthreads();
        if (wave == 0) {
            ch_diag_block(D, Xs, s_dinv, &s_fail, &s_minp, nb0, 0, lane, xout);
        } else {
            for (int i = wave - 1; i < n; i += CH_T / 64 - 1)
                for (int j = lane; j <= i; j += 64) As[i * (n + 1) + j] = A_glob[(size_t)i * ld_glob + j];
        }
        A = As;
        ld = n + 1;
        __syncthreads();
        ahead0 = true;
    }
#ifdef CH_STAMPS
    long long st_t[5] = {0, 0, 0, 0, 0}, st_last = wall_clock64();
#define CH_STAMP(K) { const long long now_ = wall_clock64(); st_t[K] += now_ - st_last; st_last = now_; }
#else
#define CH_STAMP(K)
#endif

    bool ahead = ahead0;                                  // this panel's diagonal block was factorised by the previous (d)
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
    if constexpr (SMALL) {                   // L back to the caller's matrix (rows finished before a failure included)
        __syncthreads();
        for (int i = tid >> 6; i < n; i += CH_T / 64)
            for (int j = tid & 63; j <= i; j += 64) A_glob[(size_t)i * ld_glob + j] = A[i * (n + 1) + j];
    }
#ifdef CH_STAMPS
    {
        CH_STAMP(4)                                        // (the copy back)
        if (tid == 0 && n >= 16)                           // debugging build: ticks (100 MHz) in row 0's upper triangle
            for (int k = 0; k < 5; ++k) A_glob[8 + k] = (double)st_t[k];
    }
#endif
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

// ---------------- the ladder's probes, several compute units per rung ----------------
// k_chol<false> probes every rung of the jitter ladder in one launch, one workgroup each: 0.49 ms at n = 500, bound by
// the trailing update's read-modify-write traffic through ONE compute unit's memory path (22 MB per rung), with
// 245 CUs idle.  Here CM_G workgroups share a rung: block row j (32 rows) belongs to ONE workgroup (cm_owner), which alone
// reads and writes it.  Per panel k:
//   owner(k)  factorises and inverts the diagonal block (ch_diag_block, one wave) and publishes the inverse X_k;
//   everybody forms its rows of the panel L21 = A21 X_k^T, stores them (they stay where they are in the slab) and
//             raises its flag; then fetches the panel rows of the others (up to its own last block row) into LDS;
//   everybody updates its own block rows -- the owner of block k + 1 its diagonal tile first, then one wave factorises
//             that block while the rest of the workgroup (and all other workgroups) finish the update.
// The workgroups of a rung sit on ONE XCD (hardware XCC id + a ticket per XCD, as in car_mc.hip): a plain store is in
// that XCD's L2 once vmcnt says so, and an L1-bypassing (sc1) load of a neighbour sees it -- no fences, no
// cross-XCD round trips.  XCD x serves rungs x and x + 8.  Tile arithmetic is k_chol's (same MFMA order, same split
// accumulators): info and pivots equal the one-workgroup kernel's.
constexpr int CM_G = 8;
constexpr int CM_MAXB = (CH_MAXN + CH_NB - 1) / CH_NB;
constexpr unsigned CM_SPIN = 1u << 19;
constexpr unsigned CM_OFF_TICKET = 0, CM_OFF_ERR = 64, CM_HDR = 128;
constexpr unsigned CM_RUNG_BYTES = 1024;                                  // FX[MAXB] u32 | FP[MAXB][G] u32 | MP[MAXB] f64
constexpr unsigned CM_OFF_FX = 0, CM_OFF_FP = CM_MAXB * 4, CM_OFF_MP = 768;
static_assert(CM_OFF_FP + CM_MAXB * CM_G * 4 <= CM_OFF_MP && CM_OFF_MP + CM_MAXB * 8 <= CM_RUNG_BYTES, "flag layout");
// block row -> workgroup, dealt back and forth (0 1 .. 7 7 6 .. 0 0 1 ..): a workgroup's rows j and 15 - j carry
// (j - k) + (15 - j - k) tiles' worth of update at step k -- the same for every workgroup while both are active, where
// plain round robin left the owner of rows 7 and 15 with 1.7x the average (and every step waits for the slowest)
__device__ __forceinline__ int cm_owner(int j) { const int r = j % (2 * CM_G); return r < CM_G ? r : 2 * CM_G - 1 - r; }
__device__ __forceinline__ int cm_row(int g, int idx) { return (idx >> 1) * 2 * CM_G + ((idx & 1) ? 2 * CM_G - 1 - g : g); }
constexpr int CM_INFO_EXCHANGE = -7;                                      // info of a rung whose workgroups lost each other
typedef __amdgpu_buffer_rsrc_t ch_rsrc_t;
typedef unsigned int ch_u32x4 __attribute__((ext_vector_type(4)));

constexpr int CM_POLL = 16 | (int)0x80000000;                             // sc1 + volatile: a fresh L2 read every time
__device__ __forceinline__ unsigned cm_poll(ch_rsrc_t rs, unsigned off) {          // thread-level: wait for a non-zero word
    unsigned spins = 0, v;
    while ((v = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, CM_POLL)) == 0u) {
        if (++spins > CM_SPIN) return 0xffu;
        if ((spins & 255u) == 0u && (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs, CM_OFF_ERR, 0, CM_POLL) != 0u) return 0xffu;
        __builtin_amdgcn_s_sleep(1);
    }
    return v;
}

__global__ __launch_bounds__(CH_T) void k_chol_mc(double* __restrict__ work, int n, int ld, int32_t* __restrict__ info,
                                                  double* __restrict__ min_pivot, const double* __restrict__ src,
                                                  int lds_src, const double* __restrict__ shifts, int n_rungs,
                                                  unsigned char* ws, unsigned ws_bytes, unsigned x_off) {
    extern __shared__ double lds[];
    constexpr int LDP = CH_NB + 1, LDPP = CH_LDPP;
    double* D = lds;
    double* Xs = lds + CH_NB * LDP;
    double* P = lds + 2 * CH_NB * LDP;                                    // rows below block k, relative, x LDPP
    __shared__ int s_role, s_fail;
    __shared__ double s_minp;
    __shared__ double s_dinv[CH_NB];
    __shared__ unsigned s_flag;
    __shared__ int s_sub;                                                  // arrivals at the barrier of waves 1 .. 7
    int sub_epoch = 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const ch_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(ws, 0, (int)ws_bytes, 0x00020000);
    if (tid == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u;          // HW_REG_XCC_ID
        const unsigned t = __hip_atomic_fetch_add((unsigned*)(ws + CM_OFF_TICKET) + xcc, 1u, __ATOMIC_RELAXED,
                                                  __HIP_MEMORY_SCOPE_AGENT);
        const unsigned rung = xcc + 8u * (t / CM_G);
        s_role = rung < (unsigned)n_rungs ? (int)(rung * CM_G + t % CM_G) : -1;
        s_fail = 0;
        s_flag = 0u;
        s_sub = 0;
        s_minp = __builtin_inf();
    }
    __syncthreads();
    if (s_role < 0) return;
    const int rung = s_role / CM_G, g = s_role % CM_G;
    const int nblk = (n + CH_NB - 1) / CH_NB;
    if (g >= nblk) return;
    int jmax = g;                                                         // my last block row
    for (int q = 1; cm_row(g, q) < nblk; ++q) jmax = cm_row(g, q);
    double* const A = work + (size_t)rung * n * ld;
    const ch_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(A, 0, (int)((size_t)n * ld * sizeof(double)), 0x00020000);
    const double shift = shifts[rung];
    const unsigned fbase = CM_HDR + (unsigned)rung * CM_RUNG_BYTES;
    const unsigned xbase = x_off + (unsigned)rung * CM_MAXB * (CH_NB * CH_NB * 8);
    // my block rows of the lower triangle -> my slab
    for (int q = 0; cm_row(g, q) < nblk; ++q) {
        const int j = cm_row(g, q);
        const int r1 = min(n, CH_NB * (j + 1));
        for (int i = CH_NB * j + (tid >> 6); i < r1; i += CH_T / 64)
            for (int c = lane; c <= i; c += 64) A[(size_t)i * ld + c] = src[(size_t)i * lds_src + c];
    }
    __threadfence_block();
    __syncthreads();

#ifdef CM_STAMPS
    long long cs_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, cs_last = wall_clock64();
#define CM_STAMP(K) { const long long now_ = wall_clock64(); cs_t[K] += now_ - cs_last; cs_last = now_; if (tid == 0 && n >= 480 && k < 16) A[(size_t)g * ld + 300 + 6 * k + K] = (double)now_; }
#else
#define CM_STAMP(K)
#endif
    bool ahead = false;
    int verdict = -1;                                                     // >= 0: I write the rung's result
    for (int k = 0; k <= jmax; ++k) {
        const int kb = CH_NB * k, nb = min(CH_NB, n - kb);
        const bool own_k = cm_owner(k) == g;
        if (own_k) {
            if (!ahead) {
#pragma unroll
                for (int t = tid; t < CH_NB * CH_NB; t += CH_T) {
                    const int i = t >> 5, j = t & 31;
                    double v = (i < nb && j <= i) ? A[(size_t)(kb + i) * ld + kb + j] : 0.0;
                    if (i == j && i < nb) v += shift;
                    D[i * LDP + j] = v;
                }
                __syncthreads();
                if (wave == 0) ch_diag_block(D, Xs, s_dinv, &s_fail, &s_minp, nb, kb, lane, nullptr);
                __syncthreads();
            }
            if (tid == 0) {
                const double mp = s_minp;
                __builtin_amdgcn_raw_buffer_store_b32(__double2loint(mp), rs, fbase + CM_OFF_MP + 8 * k, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__double2hiint(mp), rs, fbase + CM_OFF_MP + 8 * k + 4, 0, 0);
            }
            if (s_fail != 0 || k == nblk - 1) {                           // uniform: the rung is decided here
                if (tid == 0) {
                    __builtin_amdgcn_s_waitcnt(0);
                    __builtin_amdgcn_raw_buffer_store_b32(2, rs, fbase + CM_OFF_FX + 4 * k, 0, 0);
                }
                verdict = k;
                break;
            }
            // publish X_k
            for (int t = tid; t < CH_NB * CH_NB / 2; t += CH_T) {
                const int i = t >> 4, j = (t & 15) * 2;
                ch_u32x4 v;
                v.x = (unsigned)__double2loint(Xs[i * LDP + j]);     v.y = (unsigned)__double2hiint(Xs[i * LDP + j]);
                v.z = (unsigned)__double2loint(Xs[i * LDP + j + 1]); v.w = (unsigned)__double2hiint(Xs[i * LDP + j + 1]);
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, xbase + (unsigned)k * (CH_NB * CH_NB * 8) + (unsigned)t * 16, 0, 0);
            }
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
            if (tid == 0) __builtin_amdgcn_raw_buffer_store_b32(1, rs, fbase + CM_OFF_FX + 4 * k, 0, 0);
        } else {
            if (tid == 0) s_flag = cm_poll(rs, fbase + CM_OFF_FX + 4 * k);
            __syncthreads();
            if (s_flag != 1u) {                                            // 2: the rung is decided; 0xff: lost contact
                if (s_flag == 0xffu && tid == 0) {
                    __builtin_amdgcn_raw_buffer_store_b32(1, rs, CM_OFF_ERR, 0, 16);
                    info[rung] = CM_INFO_EXCHANGE;
                }
                break;
            }
            for (int t = tid; t < CH_NB * CH_NB / 2; t += CH_T) {
                const int i = t >> 4, j = (t & 15) * 2;
                const ch_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, xbase + (unsigned)k * (CH_NB * CH_NB * 8) + (unsigned)t * 16, 0, 16);
                Xs[i * LDP + j] = __hiloint2double((int)v.y, (int)v.x);
                Xs[i * LDP + j + 1] = __hiloint2double((int)v.w, (int)v.z);
            }
            __syncthreads();
        }
        CM_STAMP(0)
        ahead = false;
        if (k == jmax) break;                                              // no rows of mine below block k
        // ---- my rows of the panel: L21 = A21 X^T (16 x 32 per wave and trip) -> LDS and the slab
        int i_first = 0;                                                   // index of my first block row below k
        while (cm_row(g, i_first) <= k) ++i_first;
        int n_own = 0;
        while (cm_row(g, i_first + n_own) < nblk) ++n_own;
        for (int t = wave; t < 2 * n_own; t += CH_T / 64) {
            const int j = cm_row(g, i_first + (t >> 1)), r0 = CH_NB * j + 16 * (t & 1);  // global rows r0 .. r0 + 15
            const double* arow = A + (size_t)min(r0 + li, n - 1) * ld + kb;
            double af[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) af[u] = arow[min(4 * u + lk, nb - 1)];
            ch_double4 acc0 = (ch_double4){0.0, 0.0, 0.0, 0.0}, acc1 = acc0;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int kk = 4 * u + lk;
                const double a = (kk < nb) ? af[u] : 0.0;
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Xs[li * LDP + kk], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Xs[(16 + li) * LDP + kk], acc1, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = r0 + lk + 4 * r;
                if (row < n) {
                    double* pr = P + (size_t)(row - kb - CH_NB) * LDPP;
                    pr[li] = acc0[r];
                    pr[16 + li] = acc1[r];
                    if (li < nb) A[(size_t)row * ld + kb + li] = acc0[r];
                    if (16 + li < nb) A[(size_t)row * ld + kb + 16 + li] = acc1[r];
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        CM_STAMP(1)
        if (tid == 0) __builtin_amdgcn_raw_buffer_store_b32(1, rs, fbase + CM_OFF_FP + 4 * (k * CM_G + g), 0, 0);
        // I own the next diagonal block: its tile needs my own panel rows only, so wave 0 goes straight on to that tile and
        // the factorisation -- the chain of the whole launch -- while waves 1 .. 7 wait for the other workgroups' panel rows,
        // fetch them and update the rest of my rows behind a barrier of their own (an LDS counter: s_barrier counts all
        // eight waves).  Everybody else does the three phases with workgroup barriers.
        const bool look = cm_owner(k + 1) == g;
        const int nr = n - kb - CH_NB;                                     // rows below block k (P's extent)
        double* const a22 = A + (size_t)(kb + CH_NB) * ld + kb + CH_NB;
        const int n_strips = 2 * n_own;
        // my 16-row strips below block k: strip s = half s & 1 of my (s >> 1)-th block row below k; its tiles are the
        // 16-wide columns 0 .. R0 / 16 of P (up to the diagonal)
        auto strip_r0 = [&](int sidx) { return CH_NB * (cm_row(g, i_first + (sidx >> 1)) - k - 1) + 16 * (sidx & 1); };
        auto sub_barrier = [&]() {                                         // waves 1 .. 7 of this workgroup
            ++sub_epoch;
            if (lane == 0) __hip_atomic_fetch_add(&s_sub, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            while (__hip_atomic_load(&s_sub, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < 7 * sub_epoch)
                __builtin_amdgcn_s_sleep(1);
        };
        // ---- the panel rows of the others, block rows k + 1 .. jmax
        auto poll_others = [&](int o) {                                    // o: the other workgroup this lane asks
            if (o < CM_G && o != g) {
                int q = 0;
                while (cm_row(o, q) <= k) ++q;
                const int jf = cm_row(o, q);                               // that workgroup's first block row below k
                unsigned v = 1u;
                if (jf <= jmax) v = cm_poll(rs, fbase + CM_OFF_FP + 4 * (k * CM_G + o));
                if (v != 1u) s_flag = 0xffu;
            }
        };
        // (512 x 16 bytes = one 32 x 32 block; eight blocks' loads in flight before the first LDS write)
        auto gather = [&](int t0, int nthr) {
            for (int cb = k + 1; cb <= jmax; cb += 8)
                for (int gi = t0; gi < CH_NB * CH_NB / 2; gi += nthr) {
                    const int i = gi >> 4, j = (gi & 15) * 2;
                    ch_u32x4 v[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int c = cb + q, row = min(CH_NB * c + i, n - 1);
                        if (c <= jmax && cm_owner(c) != g)
                            v[q] = __builtin_amdgcn_raw_buffer_load_b128(ra, (unsigned)(((size_t)row * ld + kb + j) * 8), 0, 16);
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int c = cb + q, row = CH_NB * c + i;
                        if (c <= jmax && cm_owner(c) != g && row < n) {
                            double* pr = P + (size_t)(row - kb - CH_NB) * LDPP + j;
                            pr[0] = __hiloint2double((int)v[q].y, (int)v[q].x);
                            pr[1] = __hiloint2double((int)v[q].w, (int)v[q].z);
                        }
                    }
                }
        };
        // ---- update of my block rows: a wave walks the 16 x 16 tiles (linear over strips from strip cs0) with a cursor,
        // CH_G at a time, the next batch's loads in flight during this batch's MFMAs; stores of unused slots go to a sink
        // (k_chol's (d), same arithmetic per tile)
        auto walk = [&](int cs0, int wslot, int nwalk) {
            int cs = cs0, ct = wslot;                                      // cursor: strip, tile within the strip
                int tr[CH_G], tc[CH_G], trn[CH_G], tcn[CH_G];
                double cv[CH_G][4], cn[CH_G][4];
#define CM_TAKE(R, C)                                                                           \
                { while (cs < n_strips && ct > strip_r0(cs) / 16) { ct -= strip_r0(cs) / 16 + 1; ++cs; }   \
                  R = cs < n_strips ? strip_r0(cs) : -1; C = cs < n_strips ? 16 * ct : 0; ct += nwalk; }
#define CM_LOAD(DST, R, C)                                                                      \
                _Pragma("unroll") for (int e = 0; e < 4; ++e)                                   \
                    DST[e] = a22[(size_t)min(max(R, 0) + lk + 4 * e, nr - 1) * ld + min((C) + li, nr - 1)];
#pragma unroll
                for (int q = 0; q < CH_G; ++q) { CM_TAKE(tr[q], tc[q]) CM_LOAD(cv[q], tr[q], tc[q]) }
                while (tr[0] >= 0) {
#pragma unroll
                    for (int q = 0; q < CH_G; ++q) { CM_TAKE(trn[q], tcn[q]) CM_LOAD(cn[q], trn[q], tcn[q]) }
                    ch_double4 acc[CH_G][2];
#pragma unroll
                    for (int q = 0; q < CH_G; ++q) {
                        acc[q][0] = (ch_double4){0.0, 0.0, 0.0, 0.0};
                        acc[q][1] = acc[q][0];
                        const int pa = min(max(tr[q], 0) + li, nr - 1), pb = min(tc[q] + li, nr - 1);
#pragma unroll
                        for (int u = 0; u < 8; u += 2) {
                            acc[q][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(P[pa * LDPP + 4 * u + lk], P[pb * LDPP + 4 * u + lk], acc[q][0], 0, 0, 0);
                            acc[q][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(P[pa * LDPP + 4 * u + 4 + lk], P[pb * LDPP + 4 * u + 4 + lk], acc[q][1], 0, 0, 0);
                        }
                    }
#pragma unroll
                    for (int q = 0; q < CH_G; ++q)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int row = tr[q] + lk + 4 * e, col = tc[q] + li;
                            double* dst = (tr[q] >= 0 && row < nr && col <= row) ? a22 + (size_t)row * ld + col : ch_sink + tid;
                            *dst = cv[q][e] - (acc[q][0][e] + acc[q][1][e]);
                        }
#pragma unroll
                    for (int q = 0; q < CH_G; ++q) {
                        tr[q] = trn[q]; tc[q] = tcn[q];
#pragma unroll
                        for (int e = 0; e < 4; ++e) cv[q][e] = cn[q][e];
                    }
                }
#undef CM_TAKE
#undef CM_LOAD
        };
        if (!look) {
            poll_others(tid);
            __syncthreads();
            CM_STAMP(2)
            if (s_flag != 0xffu) gather(tid, CH_T);
            __syncthreads();
            CM_STAMP(3)
            if (s_flag != 0xffu) walk(0, wave, CH_T / 64);
        } else if (wave == 0) {
            __builtin_amdgcn_s_setprio(3);
            // block row k + 1's diagonal tile (strips 0 and 1: three tiles), then its factorisation
            const int R0[3] = {0, 16, 16}, C0[3] = {0, 0, 16};
            double cv[3][4];
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    cv[q][e] = a22[(size_t)min(R0[q] + lk + 4 * e, nr - 1) * ld + min(C0[q] + li, nr - 1)];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                ch_double4 acc0 = (ch_double4){0.0, 0.0, 0.0, 0.0}, acc1 = acc0;
                const int pa = min(R0[q] + li, nr - 1), pb = min(C0[q] + li, nr - 1);
#pragma unroll
                for (int u = 0; u < 8; u += 2) {
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(P[pa * LDPP + 4 * u + lk], P[pb * LDPP + 4 * u + lk], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(P[pa * LDPP + 4 * u + 4 + lk], P[pb * LDPP + 4 * u + 4 + lk], acc1, 0, 0, 0);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = R0[q] + lk + 4 * e, col = C0[q] + li;
                    double* dst = (row < nr && col <= row) ? a22 + (size_t)row * ld + col : ch_sink + tid;
                    *dst = cv[q][e] - (acc0[e] + acc1[e]);
                }
            }
            __builtin_amdgcn_s_waitcnt(0);
            const int kb2 = kb + CH_NB, nb2 = min(CH_NB, n - kb2);
#pragma unroll
            for (int t = lane; t < CH_NB * CH_NB; t += 64) {
                const int i = t >> 5, j = t & 31;
                double v = (i < nb2 && j <= i) ? A[(size_t)(kb2 + i) * ld + kb2 + j] : 0.0;
                if (i == j && i < nb2) v += shift;
                D[i * LDP + j] = v;
            }
            __builtin_amdgcn_wave_barrier();
            ch_diag_block(D, Xs, s_dinv, &s_fail, &s_minp, nb2, kb2, lane, nullptr);
            __builtin_amdgcn_s_setprio(0);
        } else {
            if (wave == 1) poll_others(lane);
            sub_barrier();
            if (s_flag != 0xffu) gather(tid - 64, CH_T - 64);
            sub_barrier();
            if (s_flag != 0xffu) walk(2, wave - 1, CH_T / 64 - 1);      // (strips 0 and 1, the next diagonal tile, are wave 0's)
        }
        CM_STAMP(4)
        ahead = look;
        __threadfence_block();
        __syncthreads();
        CM_STAMP(5)
        if (s_flag == 0xffu) {
            if (tid == 0) {
                __builtin_amdgcn_raw_buffer_store_b32(1, rs, CM_OFF_ERR, 0, 16);
                info[rung] = CM_INFO_EXCHANGE;
            }
            break;
        }
    }
#ifdef CM_STAMPS
    if (tid == 0 && n >= 480) { for (int q = 0; q < 6; ++q) A[(size_t)g * ld + 400 + q] = (double)cs_t[q]; A[(size_t)g * ld + 410] = ch_dbg[0]; A[(size_t)g * ld + 411] = ch_dbg[1]; }
#endif
    if (verdict >= 0 && tid == 0) {
        double mp = s_minp;                                                // every block's owner left its running minimum
        for (int q = 0; q < verdict; ++q) {
            const unsigned lo = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs, fbase + CM_OFF_MP + 8 * q, 0, 16);
            const unsigned hi = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs, fbase + CM_OFF_MP + 8 * q + 4, 0, 16);
            mp = fmin(mp, __hiloint2double((int)hi, (int)lo));
        }
        info[rung] = s_fail;
        if (min_pivot) min_pivot[rung] = mp;
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
                                                 double* __restrict__ out, int ldo, int32_t* __restrict__ flag,
                                                 unsigned long long* __restrict__ dmax_bits) {
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
            const double v = sqrt(a * b);
            out[(size_t)i * ldo + j] = v;
            // the largest diagonal entry of the result (>= 0, never NaN: the order of the bit patterns is the order of
            // the values), for the borderline test of the jitter ladder -- was a torch reduction and a copy behind this
            if (dmax_bits != nullptr && i == j) atomicMax(dmax_bits, (unsigned long long)__double_as_longlong(v));
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

namespace sober {
// (pre != NULL: a second area zeroed in the same launch, BEFORE info is written -- info may lie inside it: the Nystrom job's
//  flag block)
__global__ void k_probe_mc_init(uint32_t* __restrict__ ws, int64_t n_words, int32_t* __restrict__ info, int n_info, int32_t value,
                                uint32_t* __restrict__ pre, int64_t n_pre) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_words) ws[t] = 0u;
    if (pre != nullptr) {
        // the words info[] occupies are written by the thread that sets them (no ordering between threads is needed)
        const int64_t i0 = (uint32_t*)info - pre;
        if (t < n_pre && !(t >= i0 && t < i0 + n_info)) pre[t] = 0u;
    }
    if (t < n_info) info[t] = value;
}
}  // namespace sober

namespace sober {
// ---------------- the ladder's probes beyond one workgroup's LDS panel: CH_MAXN < n <= CB_MAXN (round 4) ----------------
// k_chol and k_chol_mc keep the panel below the diagonal block in LDS, (n - 32) x 36 doubles: n <= 536.  A Nystrom set of
// up to 2048 points is probed panel by panel with TWO launches each, every rung in the same launches:
//   k_cb_diag    one wave per rung: the 32 x 32 diagonal block (+ the rung's shift on its diagonal, as it is met) is
//                factorised and inverted in registers (ch_diag_block: k_chol's own pivots, verdict and minimum);
//   k_cb_update  one workgroup per 64 x 64 tile of the trailing lower triangle and rung: its rows of the panel
//                L21 = A21 X^T for both sides of the tile (formed on the matrix cores from the block column itself: nobody
//                has to finish a panel first, and nobody needs L afterwards -- only the verdicts are used), then
//                A22 -= L21 L21^T.  A block column is read by its own panel's launches only, after every update of it.
// 2 x n / 32 launches back to back on the stream; a failed rung keeps computing on whatever it holds (its verdict stands).
constexpr int CB_MAXN = 2048;             // (round 6: 1024 before; nothing in the panel kernels depends on it -- the work slabs are n x n per rung)
constexpr int CB_TILE = 64;

__global__ __launch_bounds__(256) void k_cb_init(double* __restrict__ work, int n, const double* __restrict__ src, int ld_src,
                                                 int32_t* __restrict__ info, double* __restrict__ min_pivot, int n_rungs) {
    const int i = blockIdx.x, rung = blockIdx.y;
    double* A = work + (size_t)rung * n * n;
    for (int c = threadIdx.x; c <= i; c += 256) A[(size_t)i * n + c] = src[(size_t)i * ld_src + c];
    if (i == 0 && threadIdx.x == 0) { info[rung] = 0; if (min_pivot) min_pivot[rung] = __builtin_inf(); }
}

__global__ __launch_bounds__(64) void k_cb_diag(double* __restrict__ work, int n, int kb, const double* __restrict__ shifts,
                                                double* __restrict__ xinv, int32_t* __restrict__ info,
                                                double* __restrict__ min_pivot) {
    constexpr int LDP = CH_NB + 1;
    __shared__ double D[CH_NB * LDP], Xs[CH_NB * LDP], s_dinv[CH_NB];
    __shared__ int s_fail;
    __shared__ double s_minp;
    const int rung = blockIdx.x, lane = threadIdx.x;
    double* A = work + (size_t)rung * n * n;
    const int nb = min(CH_NB, n - kb);
    const double shift = shifts[rung];
#pragma unroll
    for (int t = lane; t < CH_NB * CH_NB; t += 64) {
        const int i = t >> 5, j = t & 31;
        double v = (i < nb && j <= i) ? A[(size_t)(kb + i) * n + kb + j] : 0.0;
        if (i == j && i < nb) v += shift;
        D[i * LDP + j] = v;
    }
    if (lane == 0) { s_fail = 0; s_minp = __builtin_inf(); }
    __syncthreads();
    ch_diag_block(D, Xs, s_dinv, &s_fail, &s_minp, nb, kb, lane, nullptr);
    __syncthreads();
    double* xo = xinv + (size_t)rung * CH_NB * CH_NB;
#pragma unroll
    for (int t = lane; t < CH_NB * CH_NB; t += 64) xo[t] = Xs[(t >> 5) * LDP + (t & 31)];
    if (lane == 0 && info[rung] == 0) {                       // (a decided rung keeps its verdict and its minimum)
        if (min_pivot) min_pivot[rung] = fmin(min_pivot[rung], s_minp);
        if (s_fail != 0) info[rung] = s_fail;
    }
}

__global__ __launch_bounds__(256) void k_cb_update(double* __restrict__ work, int n, int kb, const double* __restrict__ xinv) {
    constexpr int LDP = CH_NB + 1, LDPP = CH_LDPP;
    __shared__ double Xs[CH_NB * LDP], Pr[CB_TILE * LDPP], Pc[CB_TILE * LDPP];
    const int rung = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    double* A = work + (size_t)rung * n * n;
    const int base = kb + CH_NB;                              // first row / column of the trailing matrix
    // linear tile index -> (bi >= bj) in the lower triangle of 64-blocks
    int bi = 0, rest = blockIdx.x;
    while (rest > bi) { rest -= bi + 1; ++bi; }
    const int bj = rest;
    const int R0 = base + CB_TILE * bi, C0 = base + CB_TILE * bj;
    const double* xi = xinv + (size_t)rung * CH_NB * CH_NB;
    for (int t = tid; t < CH_NB * CH_NB; t += 256) Xs[(t >> 5) * LDP + (t & 31)] = xi[t];
    __syncthreads();
    // my 16 rows of the panel on either side of the tile: L21 = A21 X^T (k_chol's (c))
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        if (side == 1 && bi == bj) break;                     // (uniform: the diagonal tile has one side)
        const int r0 = (side == 0 ? R0 : C0) + 16 * wave;
        const double* arow = A + (size_t)min(r0 + li, n - 1) * n + kb;
        double af[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) af[u] = arow[4 * u + lk];
        ch_double4 acc0 = (ch_double4){0.0, 0.0, 0.0, 0.0}, acc1 = acc0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = 4 * u + lk;
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(af[u], Xs[li * LDP + k], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(af[u], Xs[(16 + li) * LDP + k], acc1, 0, 0, 0);
        }
        double* P = side == 0 ? Pr : Pc;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * wave + lk + 4 * r;
            P[row * LDPP + li] = acc0[r];
            P[row * LDPP + 16 + li] = acc1[r];
        }
    }
    __syncthreads();
    const double* Pcol = (bi == bj) ? Pr : Pc;
    // A22 -= L21 L21^T for my 16 rows x 64 columns (k_chol's (d): same split accumulators)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        if (bi == bj && ct > wave) break;                     // (uniform: tiles above the diagonal)
        double cv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            cv[e] = A[(size_t)min(R0 + 16 * wave + lk + 4 * e, n - 1) * n + min(C0 + 16 * ct + li, n - 1)];
        ch_double4 acc0 = (ch_double4){0.0, 0.0, 0.0, 0.0}, acc1 = acc0;
        const int pa = 16 * wave + li, pb = 16 * ct + li;
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Pr[pa * LDPP + 4 * u + lk], Pcol[pb * LDPP + 4 * u + lk], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Pr[pa * LDPP + 4 * u + 4 + lk], Pcol[pb * LDPP + 4 * u + 4 + lk], acc1, 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = R0 + 16 * wave + lk + 4 * e, col = C0 + 16 * ct + li;
            if (row < n && col <= row) A[(size_t)row * n + col] = cv[e] - (acc0[e] + acc1[e]);
        }
    }
}
}  // namespace sober

namespace sober {
__global__ void k_orth_merge(int32_t* __restrict__ info, double* __restrict__ piv, double* __restrict__ ratio,
                             const int32_t* __restrict__ info2, const double* __restrict__ piv2, const double* __restrict__ ratio2) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (*info == 0) *info = *info2;
    *piv = fmin(*piv, *piv2);                                 // (a NaN pivot of either half stays visible: fmin would hide it)
    if (*piv2 != *piv2) *piv = *piv2;
    if (ratio != nullptr) {
        *ratio = fmin(*ratio, *ratio2);
        if (*ratio2 != *ratio2) *ratio = *ratio2;
    }
}
}  // namespace sober
int sober::orth_merge(int32_t* info, double* piv, double* ratio, const int32_t* info2, const double* piv2, const double* ratio2,
                      void* stream) {
    if (!info || !piv || !info2 || !piv2 || (ratio && !ratio2)) return SOBER_E_ARG;
    hipLaunchKernelGGL(sober::k_orth_merge, dim3(1), dim3(64), 0, (hipStream_t)stream, info, piv, ratio, info2, piv2, ratio2);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int sober_nystrom_max_n(void) { return sober::CB_MAXN; }

// every rung of the ladder for CH_MAXN < n <= CB_MAXN (see above); work: n_shifts slabs of n x n doubles, xinv_ws:
// n_shifts x 32 x 32 doubles.  info / min_pivot as sober_cholesky_probe_piv.
extern "C" int sober_cholesky_probe_batched(const double* src, int n, int ld_src, const double* shifts, int n_shifts,
                                            double* work, int32_t* info, double* min_pivot, void* xinv_ws,
                                            int64_t ws_bytes, void* stream) {
    if (!src || !shifts || !work || !info || !xinv_ws || n <= 0 || ld_src < n || n_shifts <= 0) return SOBER_E_ARG;
    if (n > sober::CB_MAXN || n_shifts > 64) return SOBER_E_DIM;
    if (ws_bytes < (int64_t)n_shifts * sober::CH_NB * sober::CH_NB * 8) return SOBER_E_WS;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(sober::k_cb_init, dim3(n, n_shifts), dim3(256), 0, st, work, n, src, ld_src, info, min_pivot, n_shifts);
    LAUNCH_CHECK();
    for (int kb = 0; kb < n; kb += sober::CH_NB) {
        hipLaunchKernelGGL(sober::k_cb_diag, dim3(n_shifts), dim3(64), 0, st, work, n, kb, shifts, (double*)xinv_ws, info,
                           min_pivot);
        LAUNCH_CHECK();
        const int rest = n - kb - sober::CH_NB;
        if (rest <= 0) break;
        const int T = (rest + sober::CB_TILE - 1) / sober::CB_TILE;
        hipLaunchKernelGGL(sober::k_cb_update, dim3(T * (T + 1) / 2, n_shifts), dim3(256), 0, st, work, n, kb,
                           (const double*)xinv_ws);
        LAUNCH_CHECK();
    }
    return 0;
}

namespace sober {
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
    static std::atomic<unsigned long long> attr_set{0};             // (one bit per device)
    if (sober_attr_needed(attr_set)) {
        HIP_TRY(hipFuncSetAttribute((const void*)sober::k_chol<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    160 * 1024 - 512));
        HIP_TRY(hipFuncSetAttribute((const void*)sober::k_chol<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    160 * 1024 - 512));
        sober_attr_done(attr_set);
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
    if (!Y || !L || !Xinv || !Q || m <= 0 || q <= 0 || q > sober::CH_MAXN || ldy < q || ldl < q || ldq < q) return SOBER_E_ARG;
    const int qpad = ((q + 31) / 32) * 32;
    const size_t bytes = (size_t)sober::TB_WAVES * (16 * (qpad + 4) + 16 * 36) * sizeof(double);
    static std::atomic<unsigned long long> attr_set{0};             // (one bit per device)
    if (sober_attr_needed(attr_set)) {
        HIP_TRY(hipFuncSetAttribute((const void*)sober::k_trsm_blocks, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    160 * 1024 - 512));          // (q = 536: 2 x 74.7 KB)
        sober_attr_done(attr_set);
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

// workspace of the multi-CU probe: header + flags (zeroed per call) | the published inverses
extern "C" int64_t sober_cholesky_probe_mc_ws_bytes(int n, int n_shifts) {
    if (n <= 0 || n_shifts <= 0 || n_shifts > 16 || n > sober::CH_MAXN) return 0;
    const int64_t flags = ((int64_t)sober::CM_HDR + (int64_t)n_shifts * sober::CM_RUNG_BYTES + 255) / 256 * 256;
    return flags + (int64_t)n_shifts * sober::CM_MAXB * sober::CH_NB * sober::CH_NB * 8;
}

// sober_cholesky_probe_piv with CM_G workgroups per rung (n_shifts <= 16).  info[r] = -7: the rung's workgroups lost each
// other (bounded spins; e.g. a dispatcher that does not spread 256 workgroups evenly over the XCDs) -- no verdict
// for that rung, the caller falls back to sober_cholesky_probe_piv or the host.
int64_t sober::probe_mc_flag_bytes(int n_shifts) {
    return ((int64_t)sober::CM_HDR + (int64_t)n_shifts * sober::CM_RUNG_BYTES + 255) / 256 * 256;
}

int sober::nystrom_flags_init(void* flags_block, int64_t flags_bytes, void* probe_ws, int64_t ws_flag_bytes, int32_t* info,
                              int n_info, void* stream) {
    if (!flags_block || !probe_ws || !info || flags_bytes <= 0 || (flags_bytes & 3) || ws_flag_bytes <= 0 || n_info <= 0)
        return SOBER_E_ARG;
    const int64_t n_pre = flags_bytes / 4, n_ws = ws_flag_bytes / 4;
    const int64_t n = n_pre > n_ws ? n_pre : n_ws;
    hipLaunchKernelGGL(sober::k_probe_mc_init, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (uint32_t*)probe_ws, n_ws, info, n_info, (int32_t)sober::CM_INFO_EXCHANGE, (uint32_t*)flags_block, n_pre);
    LAUNCH_CHECK();
    return 0;
}

int sober::cholesky_probe_mc(const double* src, int n, int ld_src, const double* shifts, int n_shifts, double* work,
                             int32_t* info, double* min_pivot, void* ws, int64_t ws_bytes, bool init, void* stream) {
    if (!src || !shifts || !work || !info || !ws || n <= 0 || ld_src < n || n_shifts <= 0) return SOBER_E_ARG;
    if (n > sober::CH_MAXN || n_shifts > 16) return SOBER_E_DIM;
    const int64_t need = sober_cholesky_probe_mc_ws_bytes(n, n_shifts);
    if (ws_bytes < need || need > 0x7fffffff) return SOBER_E_WS;
    const int64_t flags = sober::probe_mc_flag_bytes(n_shifts);
    const int nr = n > sober::CH_NB ? n - sober::CH_NB : 0;
    const size_t bytes = ((size_t)2 * sober::CH_NB * (sober::CH_NB + 1) + (size_t)nr * sober::CH_LDPP) * sizeof(double);
    static std::atomic<unsigned long long> attr_set{0};             // (one bit per device)
    if (sober_attr_needed(attr_set)) {
        HIP_TRY(hipFuncSetAttribute((const void*)sober::k_chol_mc, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    160 * 1024 - 512));
        sober_attr_done(attr_set);
    }
    // the flag area zeroed, and every rung starts as "no verdict": a rung that finds no workgroups at all (a device
    // whose XCC ids do not run over 0..7, e.g. a partitioned one) must not read as info = 0 -- ONE launch (the two
    // memsets this replaces were four fill kernels and 35 us of a host-paced stream in front of the probe)
    if (init) {
        hipLaunchKernelGGL(sober::k_probe_mc_init, dim3((unsigned)((flags / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           (uint32_t*)ws, (int64_t)(flags / 4), info, n_shifts, (int32_t)sober::CM_INFO_EXCHANGE,
                           (uint32_t*)nullptr, (int64_t)0);
        LAUNCH_CHECK();
    }
    // one workgroup per CU (the LDS request sees to that), 32 per XCD: 16 of them find a seat (two rungs x CM_G)
    const size_t lds_bytes = bytes > (size_t)84 * 1024 ? bytes : (size_t)84 * 1024;
    hipLaunchKernelGGL(sober::k_chol_mc, dim3(256), dim3(sober::CH_T), lds_bytes, (hipStream_t)stream, work, n, n, info,
                       min_pivot, src, ld_src, shifts, n_shifts, (unsigned char*)ws, (unsigned)need, (unsigned)flags);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int sober_cholesky_probe_mc(const double* src, int n, int ld_src, const double* shifts, int n_shifts,
                                       double* work, int32_t* info, double* min_pivot, void* ws, int64_t ws_bytes,
                                       void* stream) {
    return sober::cholesky_probe_mc(src, n, ld_src, shifts, n_shifts, work, info, min_pivot, ws, ws_bytes, true, stream);
}

extern "C" int sober_abs_sym_dmax(const double* C, int n, int ld, double* out, int ldo, int32_t* flag, double* dmax,
                                  void* stream) {
    if (!C || !out || !flag || n <= 0 || ld < n || ldo < n || n > 65535) return SOBER_E_ARG;
    hipLaunchKernelGGL(sober::k_abs_sym, dim3((n + 31) / 32, (n + 31) / 32), dim3(256), 0, (hipStream_t)stream, C, n,
                       ld, out, ldo, flag, (unsigned long long*)dmax);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int sober_abs_sym(const double* C, int n, int ld, double* out, int ldo, int32_t* flag, void* stream) {
    return sober_abs_sym_dmax(C, n, ld, out, ldo, flag, nullptr, stream);
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
