// K1-K3, matrix-core variant for the continuous kernels (RBF, Matern-5/2).
//
//   k(x, y) depends on  a = -1/2 |x~ - y~|^2  =  x~.y~ - 1/2|x~|^2 - 1/2|y~|^2        (x~ = (x - c)/l)
//
// With augmented points  X' = [x~, -1/2|x~|^2, 1, 0..]  and  Y' = [y~, 1, -1/2|y~|^2, 0..]  (the two
// extra slots live in the zero padding of the 4-aligned dimension), a = X'.Y' is a plain GEMM with
// K = DA = roundup4(d + 2): three v_mfma_f64_16x16x4_f64 per 16x16 tile at d = 10.  The matrix pipe
// produces the exponent argument, the vector pipe only does the exponential and the weighted
// accumulation -- the two pipes run concurrently across the waves of a SIMD.
//
// exp(): no v_exp_f64 exists.  a = n ln2/64 + r, |r| <= ln2/128; exp(a) = 2^(n>>6) * T[n&63] * e^r with
// a 64-entry table in LDS and a degree-5 polynomial (truncation 3.5e-17): 11 FP64 + 4 int ops
// (libm's exp is ~2.5x that).  Max relative error ~2.5e-16.
//
// Wave tile: 64 rows (4 MFMA row tiles) x 16 sets; lane l holds column j = l & 15 (one candidate ->
// one set) and rows (l >> 4) + 4*reg of each tile.  Workgroup = 4 waves = 256 rows sharing the staged
// candidate tile.  Accumulators stay in VGPRs over the whole element chunk: fixed order, no atomics.
#include "common.hpp"

namespace sober {

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int LM_RW = 4;     // waves (64 rows each) per workgroup
constexpr int LM_SB = 16;    // sets per workgroup = MFMA N
constexpr int LM_TE = 8;     // elements staged per tile

// exp of four independent arguments, written stage by stage so that the four dependency chains
// interleave (one chain alone leaves the FP64 pipe idle for most of its latency).
//
// The argument arrives PRE-SCALED: y = a * 64/ln2 (the scale is folded into the augmented rows, so the
// GEMM delivers y for free).  Then  n = rint(y)  via the 2^52 magic-number add (its low dword IS n),
// r' = y - n exactly (no Cody-Waite split needed), exp(a) = 2^(n>>6) T[n&63] (1 + p(r')) with the
// polynomial coefficients pre-multiplied by (ln2/64)^k, and the final scaling by 2^(n>>6) is an
// integer add on the exponent field.  10 FP64 ops per value (libm: ~30).
__device__ __forceinline__ void exp_tab4(const double (&y)[4], const double* __restrict__ T, double (&out)[4]) {
    const double MAGIC = 6755399441055744.0;        // 1.5 * 2^52
    const double K1 = 1.0830424696249145e-02;       // (ln2/64)^k / k!
    const double K2 = 5.86490495505617e-05;
    const double K3 = 2.1173137155464776e-07;
    const double K4 = 5.732851688640402e-10;
    const double K5 = 1.2417843701716925e-12;
    double yc[4], u[4], r[4], t[4], q[4];
    int n[4];
    // clamp at about -65000 (e^-704 ~ 1e-306 keeps 2^(n>>6) a normal number) as an UNSIGNED MIN ON THE HIGH DWORD:
    // for negative doubles a larger bit pattern is a more negative value, non-negative ones compare below any
    // negative pattern and pass unchanged (NaN too).  An integer op: it does not take a slot of the FP64 units,
    // which the exponentials and the MFMAs share.
#pragma unroll
    for (int i = 0; i < 4; ++i)
        yc[i] = __hiloint2double((int)min((unsigned)__double2hiint(y[i]), 0xC0EFBD00u), __double2loint(y[i]));
#pragma unroll
    for (int i = 0; i < 4; ++i) u[i] = yc[i] + MAGIC;
#pragma unroll
    for (int i = 0; i < 4; ++i) n[i] = __double2loint(u[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = T[n[i] & 63];
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = yc[i] - (u[i] - MAGIC);
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = fma(r[i], K5, K4);
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = fma(r[i], q[i], K3);
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = fma(r[i], q[i], K2);
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = fma(r[i], q[i], K1);
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = q[i] * r[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double v = fma(t[i], q[i], t[i]);                        // in [1, 2)
        int hi;                                                        // exponent field += n >> 6: shift + fused shift-add
        asm("v_lshl_add_u32 %0, %1, 20, %2" : "=v"(hi) : "v"(n[i] >> 6), "v"(__double2hiint(v)));
        out[i] = __hiloint2double(hi, __double2loint(v));
    }
}

template <int KIND>
__device__ __forceinline__ void kern_from_arg4(const double4_t& c, const double* __restrict__ T, double (&k)[4]) {
    if constexpr (KIND == SOBER_KIND_RBF) {
        const double y[4] = {c[0], c[1], c[2], c[3]};                 // already -sq/2 * 64/ln2
        exp_tab4(y, T, k);
    } else {
        const double s5 = 2.23606797749978969641;
        const double INV_L = 1.0830424696249145e-02;                   // ln2/64: undo the pre-scaling
        const double L = 92.33248261689366;
        double sq[4], rr[4], a[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) sq[i] = fmax((-2.0 * INV_L) * c[i], 1e-30);   // clamp_min(1e-30) before sqrt
#pragma unroll
        for (int i = 0; i < 4; ++i) rr[i] = sqrt(sq[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = (-s5 * L) * rr[i];
        exp_tab4(a, T, k);
#pragma unroll
        for (int i = 0; i < 4; ++i) k[i] *= (s5 * rr[i] + 1.0) + (5.0 / 3.0) * sq[i];
    }
}

template <int KIND, int KT>      // KT = DA / 4 k-steps
__global__ __launch_bounds__(LM_RW * 64) void k_level_reduce_mfma(
    const double* __restrict__ rows, int n_rows,          // n_rows x DA   (A side, augmented)
    const double* __restrict__ cand,                      // N x DA        (B side, augmented)
    const int32_t* __restrict__ idx, int64_t pos0, int64_t count, int S,
    const double* __restrict__ mu, const double* __restrict__ wmul, double os,
    int64_t e_first, int e_total, int e_per_chunk,
    double* __restrict__ partG, int ldg, int col0,
    double* __restrict__ partTot, int64_t tot_limit,
    const int64_t* __restrict__ dR, int S_main, int leftover) {
    constexpr int DA = 4 * KT;
    // queued levels (level_exec.cpp): the launch was sized from an UPPER BOUND of the live positions; the exact
    // number R sits in device memory (written by the previous level's update).  leftover = 0: the main launch over
    // positions [0, R) (S = S_main); leftover = 1: positions [E S_main, R) spread over S pseudo-sets.
    if (dR != nullptr) {
        const int64_t R = __hip_atomic_load(dR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (not through the scalar cache)
        if (R <= S_main) return;                          // nothing to halve (or the chain was stopped: R = -1)
        const int64_t ES = (R / S_main) * S_main;
        if (!leftover) { count = R; tot_limit = ES; }
        else { idx += ES; count = R - ES; tot_limit = count; }
        if (count <= 0) return;
        pos0 = 0; e_first = 0;
        e_total = (int)((count + S - 1) / S);
        const int nch = level_chunks_for(n_rows, e_total, S);
        if ((int)blockIdx.y >= nch) return;
        e_per_chunk = (e_total + nch - 1) / nch;
    }
    constexpr int SB = LM_SB, TE = LM_TE, NT = TE * SB;
    __shared__ double s_pts[2][TE][DA][SB];      // k-major: a B fragment read is 512 contiguous bytes
    __shared__ double s_w[2][NT];
    __shared__ double s_tot[NT];
    __shared__ double s_T[64];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lj = lane & 15, lk = lane >> 4;
    const int s0 = blockIdx.x * SB;
    const int chunk = blockIdx.y;
    const int row0 = blockIdx.z * (LM_RW * 64) + wave * 64;
    const int e0 = chunk * e_per_chunk;
    const int e1 = min(e0 + e_per_chunk, e_total);

    if (tid < 64) s_T[tid] = exp2((double)tid * (1.0 / 64.0));

    // A fragments: lane holds rows[row0 + 16*t + lj][4*ks + lk]
    double afr[4][KT];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int r = row0 + 16 * t + lj;
#pragma unroll
        for (int ks = 0; ks < KT; ++ks) afr[t][ks] = (r < n_rows) ? rows[(size_t)r * DA + 4 * ks + lk] : 0.0;
    }
    double4_t acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (double4_t){0.0, 0.0, 0.0, 0.0};

    // staging: thread tid < NT stages candidate (te = tid / SB, i = tid % SB) of the NEXT tile while the
    // current one is being consumed.  The address chain idx -> (mu, candidate row) is two global round
    // trips, so the index is fetched TWO tiles ahead and the row ONE tile ahead.  Every load is
    // unconditional (clamped position, zero weight for padding): with loads under a branch the compiler
    // can no longer count outstanding loads and falls back to waiting for all of them before the MFMAs.
    const bool stager = tid < NT;
    const int st_te = tid / SB, st_i = tid % SB;
    double st[DA];
    double tot_acc = 0.0, m_raw = 0.0, wv_raw = 0.0;
    bool ok_st = false, ok_tot = false;
    const double* wm_ptr = wmul ? wmul : mu;
    const int64_t p_last = pos0 + count - 1;
    int c_pref = 0;                                   // idx of my candidate two tiles ahead

#define LM_POS(e_tile, P, OK)                                                              \
    const int e_##P = (e_tile) + st_te;                                                    \
    const int s_##P = s0 + st_i;                                                           \
    const int64_t P = (e_first + e_##P) * S + s_##P;                                       \
    const bool OK = stager && (s_##P < S) && (e_##P < e1) && (P >= pos0) && (P <= p_last);
#define LM_PREFETCH_IDX(e_tile)                                                            \
    {                                                                                      \
        LM_POS(e_tile, pp_, okp_)                                                          \
        const int64_t pc_ = okp_ ? pp_ : pos0;                                             \
        c_pref = idx[pc_ - pos0];                                                          \
    }
    // consume the prefetched index: ISSUE the loads of (weight, row) for that tile -- nothing here may touch
    // the loaded values (their first use decides where the compiler waits): that happens in LM_STAGE_WRITE,
    // after the tile in flight has been consumed
#define LM_STAGE_LOAD(e_tile)                                                              \
    {                                                                                      \
        LM_POS(e_tile, pl_, okl_)                                                          \
        const int c_ = okl_ ? c_pref : 0;                                                  \
        m_raw = mu[c_];                                                                    \
        wv_raw = wm_ptr[c_];                                                               \
        const double* src_ = cand + (size_t)c_ * DA;                                       \
        _Pragma("unroll") for (int j = 0; j < DA; ++j) st[j] = src_[j];                    \
        ok_st = okl_;                                                                      \
        ok_tot = okl_ && pl_ < tot_limit;                                                  \
    }
#define LM_STAGE_WRITE(buf)                                                                \
    if (stager) {                                                                          \
        _Pragma("unroll") for (int j = 0; j < DA; ++j) s_pts[buf][st_te][j][st_i] = ok_st ? st[j] : 0.0; \
        s_w[buf][tid] = ok_st ? (wmul ? m_raw * wv_raw : m_raw) * os : 0.0;                \
        tot_acc += ok_tot ? m_raw : 0.0;                                                   \
    }

    int buf = 0;
    LM_PREFETCH_IDX(e0)
    __builtin_amdgcn_s_waitcnt(0);                     // A fragments + first index: nothing pending at loop entry
    LM_STAGE_LOAD(e0)
    LM_PREFETCH_IDX(e0 + TE)
    LM_STAGE_WRITE(0)
    __syncthreads();

    const bool rows_live = row0 < n_rows;               // a wave wholly past the row table only helps staging
    for (int et = e0; et < e1; et += TE) {
        LM_STAGE_LOAD(et + TE)                                          // (padding when there is no next tile)
        LM_PREFETCH_IDX(et + 2 * TE)
        const int te_cnt = rows_live ? min(TE, e1 - et) : 0;
        // software pipeline over the elements of the tile: the MFMAs of element te+1 are issued
        // before the exponentials of element te, so the matrix and vector pipes overlap in-wave
        double4_t cc[4] = {};
        double wc = 0.0;
        if (rows_live) {
            double bfr[KT];
#pragma unroll
            for (int ks = 0; ks < KT; ++ks) bfr[ks] = s_pts[buf][0][4 * ks + lk][lj];
            wc = s_w[buf][lj];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                cc[t] = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < KT; ++ks)
                    cc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(afr[t][ks], bfr[ks], cc[t], 0, 0, 0);
            }
        }
        for (int te = 0; te < te_cnt; ++te) {
            double4_t cn[4];
            double wn = 0.0;
            const int tn = min(te + 1, TE - 1);                 // last iteration: harmless re-read
            {
                double bfr[KT];
#pragma unroll
                for (int ks = 0; ks < KT; ++ks) bfr[ks] = s_pts[buf][tn][4 * ks + lk][lj];
                wn = s_w[buf][tn * SB + lj];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    cn[t] = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int ks = 0; ks < KT; ++ks)
                        cn[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(afr[t][ks], bfr[ks], cn[t], 0, 0, 0);
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                double k[4];
                kern_from_arg4<KIND>(cc[t], s_T, k);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[t][r] = fma(k[r], wc, acc[t][r]);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) cc[t] = cn[t];
            wc = wn;
        }
        LM_STAGE_WRITE(buf ^ 1)                                         // (zeros when there is no next tile)
        __syncthreads();
        buf ^= 1;
    }
#undef LM_STAGE_LOAD
#undef LM_STAGE_WRITE
#undef LM_PREFETCH_IDX
#undef LM_POS

    // C/D map of the f64 MFMA: col = lane & 15, row = (lane >> 4) + 4 * reg
    if (s0 + lj < S) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + 16 * t + lk + 4 * r;
                if (row < n_rows) partG[((size_t)chunk * n_rows + row) * ldg + col0 + s0 + lj] = acc[t][r];
            }
    }
    if (partTot != nullptr && blockIdx.z == 0) {
        if (stager) s_tot[tid] = tot_acc;
        __syncthreads();
        if (tid < SB && s0 + tid < S) {
            double t = 0.0;
#pragma unroll
            for (int te = 0; te < TE; ++te) t += s_tot[te * SB + tid];
            partTot[(size_t)chunk * ldg + col0 + s0 + tid] = t;
        }
    }
}

// points -> augmented, centred, scaled rows.  side 1 (pool): [y~, 1, -|y~|^2/2, 0..];
// side 0 (row table): L * [x~, -|x~|^2/2, 1, 0..] with L = 64/ln2, so that X'.Y' = -|x~ - y~|^2/2 * L is
// directly the table-exp's scaled argument
// One 16-lane DPP row per point (lane l holds coordinates l and l + 16: DA <= 32): the row of X is read and the
// augmented row written as contiguous 128-byte segments, |x~|^2 is an in-row DPP reduction.
template <int CTRL>
__device__ __forceinline__ double aug_dpp(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__global__ __launch_bounds__(256) void k_augment_points(const double* __restrict__ X, int64_t n, int d, int64_t ldx,
                                                        const double* __restrict__ ls, int ls_len,
                                                        const double* __restrict__ center, int side,
                                                        double* __restrict__ out, int da) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int l16 = threadIdx.x & 15;
    const bool live = i < n;
    const int64_t ir = live ? i : n - 1;
    const double sc = (side == 0) ? 92.33248261689366 : 1.0;
    double v[2], nrm = 0.0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int j = l16 + 16 * h, jc = min(j, d - 1);
        const double x = (X[ir * ldx + jc] - center[jc]) / ls[ls_len == 1 ? 0 : jc];
        v[h] = (j < d) ? x : 0.0;
        nrm = fma(v[h], v[h], nrm);
    }
    nrm += aug_dpp<0x128>(nrm);                       // row_ror 8, 4, 2, 1: every lane gets the row total
    nrm += aug_dpp<0x124>(nrm);
    nrm += aug_dpp<0x122>(nrm);
    nrm += aug_dpp<0x121>(nrm);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int j = l16 + 16 * h;
        double o = v[h] * sc;
        if (j == d + side) o = -0.5 * nrm * sc;
        else if (j == d + 1 - side) o = sc;
        else if (j >= d) o = 0.0;
        if (live && j < da) out[i * da + j] = o;
    }
}

template <int KIND, int KT>
static int launch_lm(const double* rows, int n_rows, const double* cand, const int32_t* idx, int64_t pos0,
                     int64_t count, int S, const double* mu, const double* wmul, double os, int n_chunks,
                     double* partG, int ldg, int col0, double* partTot, int64_t tot_limit, hipStream_t st,
                     const int64_t* dR = nullptr, int S_main = 0, int leftover = 0) {
    const int64_t e_first = pos0 / S;
    const int e_total = (int)((pos0 + count + S - 1) / S - e_first);
    const int e_per_chunk = (e_total + n_chunks - 1) / n_chunks;
    dim3 grid((S + LM_SB - 1) / LM_SB, n_chunks, (n_rows + LM_RW * 64 - 1) / (LM_RW * 64));
    hipLaunchKernelGGL((k_level_reduce_mfma<KIND, KT>), grid, dim3(LM_RW * 64), 0, st, rows, n_rows, cand, idx,
                       pos0, count, S, mu, wmul, os, e_first, e_total, e_per_chunk, partG, ldg, col0, partTot,
                       tot_limit, dR, S_main, leftover);
    LAUNCH_CHECK();
    return 0;
}

}  // namespace sober

using namespace sober;

extern "C" int sober_aug_dim(int d) {
    if (d <= 0) return SOBER_E_ARG;
    const int da = ((d + 2 + 3) / 4) * 4;
    return da <= 32 ? da : SOBER_E_DIM;
}

extern "C" int sober_augment_points(const double* X, int64_t n, int d, int64_t ldx, const double* lengthscale,
                                    int ls_len, const double* center, int side, double* out, int da,
                                    void* stream) {
    if (!X || !lengthscale || !center || !out || n <= 0 || d <= 0 || ldx < d || da < d + 2) return SOBER_E_ARG;
    if ((ls_len != 1 && ls_len != d) || (side != 0 && side != 1)) return SOBER_E_ARG;
    if (da > 32) return SOBER_E_DIM;
    hipLaunchKernelGGL(k_augment_points, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, (hipStream_t)stream, X, n,
                       d, ldx, lengthscale, ls_len, center, side, out, da);
    LAUNCH_CHECK();
    return 0;
}

static int level_reduce_mfma_impl(int kind, const double* rows, int n_rows, const double* cand, int da,
                                  const int32_t* idx, int64_t pos0, int64_t count, int S, const double* mu,
                                  const double* wmul, double outputscale, int n_chunks, double* partG,
                                  int ldg, int col0, double* partTot, int64_t tot_limit, void* stream,
                                  const int64_t* dR, int S_main, int leftover);

extern "C" int sober_level_reduce_mfma(int kind, const double* rows, int n_rows, const double* cand, int da,
                                       const int32_t* idx, int64_t pos0, int64_t count, int S, const double* mu,
                                       const double* wmul, double outputscale, int n_chunks, double* partG,
                                       int ldg, int col0, double* partTot, int64_t tot_limit, void* stream) {
    if (n_chunks > (pos0 + count + S - 1) / S - pos0 / S) return SOBER_E_ARG;
    return level_reduce_mfma_impl(kind, rows, n_rows, cand, da, idx, pos0, count, S, mu, wmul, outputscale, n_chunks,
                                  partG, ldg, col0, partTot, tot_limit, stream, nullptr, 0, 0);
}

extern "C" int sober_level_reduce_mfma_queued(int kind, const double* rows, int n_rows, const double* cand, int da,
                                              const int32_t* idx, int64_t count_ub, int S, int S_main,
                                              int leftover, const double* mu, const double* wmul,
                                              double outputscale, int n_chunks_ub, double* partG, int ldg,
                                              double* partTot, const int64_t* dR, void* stream) {
    if (!dR || S_main <= 0 || (!leftover && S != S_main)) return SOBER_E_ARG;
    return level_reduce_mfma_impl(kind, rows, n_rows, cand, da, idx, 0, count_ub, S, mu, wmul, outputscale,
                                  n_chunks_ub, partG, ldg, 0, partTot, 0, stream, dR, S_main, leftover);
}

static int level_reduce_mfma_impl(int kind, const double* rows, int n_rows, const double* cand, int da,
                                  const int32_t* idx, int64_t pos0, int64_t count, int S, const double* mu,
                                  const double* wmul, double outputscale, int n_chunks, double* partG,
                                  int ldg, int col0, double* partTot, int64_t tot_limit, void* stream,
                                  const int64_t* dR, int S_main, int leftover) {
    if (!rows || !cand || !idx || !mu || !partG) return SOBER_E_ARG;
    if (n_rows <= 0 || pos0 < 0 || count <= 0 || S <= 0 || n_chunks <= 0 || ldg < col0 + S) return SOBER_E_ARG;
    hipStream_t st = (hipStream_t)stream;
#define LM_CASE(K, T)                                                                                          \
    case 4 * T:                                                                                                \
        return launch_lm<K, T>(rows, n_rows, cand, idx, pos0, count, S, mu, wmul, outputscale, n_chunks, partG, \
                               ldg, col0, partTot, tot_limit, st, dR, S_main, leftover);
    switch (kind) {
        case SOBER_KIND_RBF:
            switch (da) { LM_CASE(SOBER_KIND_RBF, 1) LM_CASE(SOBER_KIND_RBF, 2) LM_CASE(SOBER_KIND_RBF, 3)
                          LM_CASE(SOBER_KIND_RBF, 4) LM_CASE(SOBER_KIND_RBF, 5) LM_CASE(SOBER_KIND_RBF, 6)
                          LM_CASE(SOBER_KIND_RBF, 7) LM_CASE(SOBER_KIND_RBF, 8) default: return SOBER_E_DIM; }
        case SOBER_KIND_MATERN52:
            switch (da) { LM_CASE(SOBER_KIND_MATERN52, 1) LM_CASE(SOBER_KIND_MATERN52, 2)
                          LM_CASE(SOBER_KIND_MATERN52, 3) LM_CASE(SOBER_KIND_MATERN52, 4)
                          LM_CASE(SOBER_KIND_MATERN52, 5) LM_CASE(SOBER_KIND_MATERN52, 6)
                          LM_CASE(SOBER_KIND_MATERN52, 7) LM_CASE(SOBER_KIND_MATERN52, 8)
                          default: return SOBER_E_DIM; }
        default: return SOBER_E_ARG;
    }
#undef LM_CASE
}
