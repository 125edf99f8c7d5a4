// GP prediction over a candidate pool in ONE launch (SOBER/_gp.py:212-238, SOBER/_pi.py:20-38):
//     mean(x) = c0 + k(x, X_obs) alpha,   var(x) = k(x, x) - k(x, X_obs) W k(X_obs, x) + noise,   pi(x) = Phi((mean - eta) / sqrt(var))
// Round 4 ran it as four launches over the pool -- the posterior mean (one pass of kernel evaluations), K(X_obs, pool)
// materialised (a second pass; 160 MB at 200 x 100k), V = W KX by the latency-oriented k_dgemm (one wave per 16 x 16 tile,
// fragments straight from L2: every row tile of W re-read the whole KX), and the column-wise quadratic form (KX and V read
// again) -- 1.6 ms per call at configuration 2's shapes, two calls per `Sober.next_batch`: 45 % of the acquisition step
// (profiles/r05_funnel.json).  Here a workgroup owns PF_NB = 32 candidates:
//   1. KX tile (n_obs x 32) evaluated ONCE into LDS: thread <-> observation row (its coordinates in registers), the 32
//      candidates' rows broadcast from LDS (a first form had lane <-> candidate and read the observation rows from L2 inside
//      the loop: a dependent round trip per row, 26 us of a workgroup's 100);
//   2. V' = [W; alpha^T] KX on the FP64 matrix cores (the extra row IS the mean's sum): wave w owns the row tiles w, w + 4,
//      w + 8, w + 12 (x 2 candidate tiles = 8 accumulators); the fragments of W (symmetric: read as W^T, 128 contiguous bytes per 16 lanes) come straight from L2
//      three k-steps ahead, the KX fragments from LDS -- no barrier inside (a first form streamed W through LDS panels
//      with a barrier each: 0.76 ms per call, the L2 round trip of every panel exposed behind 16 matrix instructions);
//   3. var = kxx - sum_r KX[r][c] V[r][c] + noise straight from the accumulators, pi(x) with the reference's erfc form.
// Bound: the FP64 matrix cores (2 n_obs^2 flop per candidate: 8 GFLOP at 200 x 100k; half of it from a triangular root, below).
// n_obs <= 511 (the mean's row included: sixteen row tiles over four waves up to 255 observations, thirty-two -- NRT = 8 -- up
// to 511; the KX tile then takes up to 132 KB of LDS); beyond that the caller keeps the materialised route.
#include "common.hpp"

namespace sober {

#include "kern_exp.inc"

constexpr int PF_NB = 32;                 // candidates per workgroup
constexpr int PF_MAX_OBS = 511;           // n_obs + 1 rows of [W; alpha^T] in 4 NRT row tiles over four waves (NRT = 4: n_obs <= 255, 8: <= 511)
typedef double pf_d4 __attribute__((ext_vector_type(4)));

__host__ __device__ inline int pf_obs_pad(int n_obs) { return (n_obs + 15) / 16 * 16; }
// row stride of the KX tile, kept CANDIDATE-major in LDS (kxt[c][r]): = 2 mod 32 doubles, so that the B fragments
// (16 candidates x 2 neighbouring r per half-wave) fall on 32 distinct bank pairs
__host__ __device__ inline int pf_rs(int n_obs) { return (pf_obs_pad(n_obs) + 31) / 32 * 32 + 2; }
__host__ __device__ inline size_t pf_lds_bytes(int n_obs, int dt) {
    return ((size_t)PF_NB * pf_rs(n_obs) + (size_t)PF_NB * dt + 2 * PF_NB + 4 * PF_NB + EXP_TAB) * sizeof(double);
}

// DT: the prepared rows' length (padded dimension of the continuous kernels, 64-bit words of a fingerprint) -- static, so
// that the observation's coordinates stay in registers and the distance loops carry no bounds
// ROOT (round 6): W is not formed -- the table passed as `W` is S^T, the transposed root of W = S S^T (SOBER/_gp.py:277: gpytorch's
// covar_cache, the inverse Cholesky factor's transpose: UPPER triangular, so S^T is lower triangular), the matrix cores form
// V'' = S^T KX and the quadratic form is |V''|^2 by column.  *tri_flag != 0 says that S^T is lower triangular: row tile T then needs
// the observations k <= 16 T + 15 only -- 91 of the 169 tile products at n_obs = 200, and the busiest wave carries 28 of them
// instead of 52.  (The k index of a step's four slices is 8 p + 2 lk + {0, 1} there: consecutive steps walk along k.)
template <int KIND, int DT, bool ROOT, int NRT>
__global__ __launch_bounds__(256) void k_predict_fused(const double* __restrict__ obs, const double* __restrict__ obs_norm,
                                                       int n_obs, const double* __restrict__ cand,
                                                       const double* __restrict__ cand_norm, int64_t N, int dt,
                                                       double outputscale, const double* __restrict__ W, int ldw,
                                                       const double* __restrict__ alpha, double c0, double kxx_const,
                                                       double noise, double* __restrict__ mean_out,
                                                       double* __restrict__ var_out, double eta,
                                                       const double* __restrict__ eta_ptr,
                                                       double* __restrict__ lfi_out, int log_flag,
                                                       const int32_t* __restrict__ tri_flag) {
    extern __shared__ __attribute__((aligned(16))) double pf_lds[];
    if (eta_ptr != nullptr) eta = *eta_ptr;                       // (the threshold still in device memory: no host read-back)
    const int n_pad = pf_obs_pad(n_obs), rs = pf_rs(n_obs);
    double* const kxt = pf_lds;                                  // [PF_NB][rs]: K(X_obs, x_c), candidate-major
    double* const ys = kxt + (size_t)PF_NB * rs;                 // [PF_NB][DT]: the candidates' prepared rows
    double* const s_yn = ys + (size_t)PF_NB * DT;                // [PF_NB]: their popcounts (Tanimoto)
    double* const s_mean = s_yn + PF_NB;                         // [PF_NB]: alpha^T KX, out of the matrix cores
    double* const s_q = s_mean + PF_NB;                          // [4][PF_NB]
    double* const s_T = s_q + 4 * PF_NB;                         // [EXP_TAB]: 2^(j/256), exponent-adjusted (kern_exp.inc)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int64_t c_base = (int64_t)blockIdx.x * PF_NB;

    // ---- 1. the KX tile: thread <-> observation row r = tid (its coordinates in registers), the 32 candidates' rows
    //      broadcast from LDS, four candidates at a time through the table-driven exponential of the level kernel (kern_exp.inc);
    //      kxt[c][r] is written with consecutive lanes on consecutive addresses
    for (int t = tid; t < PF_NB * DT; t += 256) {
        const int cl = t / DT, q = t - cl * DT;
        ys[t] = cand[min(c_base + cl, N - 1) * DT + q];
    }
    if (tid < PF_NB) { s_yn[tid] = (KIND == SOBER_KIND_TANIMOTO) ? cand_norm[min(c_base + tid, N - 1)] : 0.0; s_mean[tid] = 0.0; }
    if (tid < EXP_TAB) s_T[tid] = exp_tab_entry(tid);
    __syncthreads();
    for (int r = tid; r < rs; r += 256) {                         // (one trip up to 255 observations, at most three at 511)
        const bool live = r < n_obs;
        double x[DT];
#pragma unroll
        for (int q = 0; q < DT; ++q) x[q] = obs[(size_t)min(r, n_obs - 1) * DT + q];
        const double nx = (KIND == SOBER_KIND_TANIMOTO) ? obs_norm[min(r, n_obs - 1)] : 0.0;
        {
            for (int c4 = 0; c4 < PF_NB; c4 += 4) {
                double k[4];
                if constexpr (KIND == SOBER_KIND_TANIMOTO) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const double* y = ys + (size_t)(c4 + u) * DT;
                        int dot = 0;
#pragma unroll
                        for (int q = 0; q < DT; ++q) dot += __popcll(__double_as_longlong(x[q]) & __double_as_longlong(y[q]));
                        k[u] = kern_tanimoto((double)dot, nx, s_yn[c4 + u], outputscale);
                    }
                } else {
                    double4_t arg;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const double* y = ys + (size_t)(c4 + u) * DT;
                        double sq0 = 0.0, sq1 = 0.0;
#pragma unroll
                        for (int q = 0; q < DT; q += 2) {
                            const double d0 = x[q] - y[q], d1 = x[q + 1] - y[q + 1];
                            sq0 = fma(d0, d0, sq0);
                            sq1 = fma(d1, d1, sq1);
                        }
                        arg[u] = (sq0 + sq1) * (-0.5 * 369.3299304675746);      // -|x~ - y~|^2 / 2 * 256 / ln2
                    }
                    kern_from_arg4<KIND>(arg, s_T, k);
#pragma unroll
                    for (int u = 0; u < 4; ++u) k[u] *= outputscale;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) kxt[(size_t)(c4 + u) * rs + r] = live ? k[u] : 0.0;   // (rows past n_obs: the k-padding)
            }
        }
    }
    __syncthreads();                                              // the KX tile is complete
    // ---- 2. V' = [W; alpha^T] KX on the FP64 matrix cores: my row tiles T = wave + 4 rt (rt < 4; 16 T <= n_obs), candidate
    //      tiles ct = 0, 1.  Row n_obs of the product is alpha^T KX = mean - c0.  The A fragments (W^T = W: lane (li, lk)
    //      takes W[k = 4 ks + lk][16 T + li], 128 contiguous bytes per 16 lanes) come straight from L2 -- W is 320 KB and
    //      every workgroup reads it -- three k-steps ahead of their matrix instructions; the B fragments from LDS.
    pf_d4 acc[NRT][2];
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) acc[rt][ct] = (pf_d4){0.0, 0.0, 0.0, 0.0};
    // The k index of a matrix instruction's four slices is ANY four values (the sum is order-free as long as A and B agree):
    // lane group lk takes the contiguous quarter [lk Kq, (lk + 1) Kq) of k, so that a lane's fragments of consecutive steps
    // are consecutive in memory -- A (a row of W, or alpha) and B (a candidate's row of the KX tile) arrive two steps per
    // 16-byte load.
    const int Kq = n_pad >> 2;                                    // (a multiple of 4)
    const int n_wrows = n_obs + (alpha != nullptr ? 1 : 0);
    typedef double pf_d2 __attribute__((ext_vector_type(2), aligned(8)));
    const double* arow[NRT];
    bool wlive[NRT], wzero[NRT];
    // which row tile is (wave, rt)'s: wave + 4 rt -- or, with a triangular root (tile T costs T + 1 steps), the live tiles dealt
    // out from the longest in a snake (wave 0 1 2 3 3 2 1 0 0 ...): 24 / 23 / 22 / 22 steps per wave at 13 tiles instead of
    // 28 / 18 / 21 / 24
    const int n_tiles = (n_wrows + 15) >> 4;
    int tile[NRT];
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) {
        if constexpr (ROOT) {
            const int slot = 4 * rt + ((rt & 1) ? 3 - wave : wave);
            tile[rt] = slot < n_tiles ? n_tiles - 1 - slot : 4 * NRT + slot; // (a slot past the live tiles: a tile nobody has)
        } else {
            tile[rt] = wave + 4 * rt;
        }
    }
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) {
        const int T = tile[rt], wr = 16 * T + li;
        wlive[rt] = 16 * T < n_wrows;                             // (wave-uniform)
        wzero[rt] = wr >= n_wrows;                                // rows past [W; alpha^T]: zeros
        arow[rt] = (wr < n_obs) ? W + (size_t)wr * ldw : ((wr == n_obs && alpha != nullptr) ? alpha : W);
    }
    const int kbase = lk * Kq;
    // (unconditional loads at clamped addresses; what lies past the table becomes a zero)
#define PF_LOADA(DST, KP)                                                                          \
    {                                                                                              \
        const int k0_ = ROOT ? 8 * (KP) + 2 * lk : kbase + 2 * (KP);    /* (even) */               \
        const int kc_ = max(min(k0_, n_obs - 2), 0);          /* the pair [kc, kc + 1] lies inside the row */ \
        const bool odd_ = k0_ == n_obs - 1;                   /* the row's last entry sits in the pair's second half */ \
        _Pragma("unroll") for (int rt = 0; rt < NRT; ++rt) {                                       \
            const pf_d2 v_ = *(const pf_d2*)(arow[rt] + kc_);                                      \
            DST[rt][0] = (wzero[rt] || k0_ >= n_obs) ? 0.0 : (odd_ ? v_[1] : v_[0]);               \
            DST[rt][1] = (wzero[rt] || k0_ + 1 >= n_obs) ? 0.0 : v_[1];                            \
        }                                                                                          \
    }
#define PF_STEP2(A_, KP)                                                                           \
    {                                                                                              \
        pf_d2 b_[2];                                                                               \
        const int kb_ = ROOT ? 8 * (KP) + 2 * lk : kbase + 2 * (KP);                               \
        _Pragma("unroll") for (int ct = 0; ct < 2; ++ct) b_[ct] = *(const pf_d2*)(kxt + (size_t)(16 * ct + li) * rs + kb_); \
        _Pragma("unroll") for (int h = 0; h < 2; ++h)                                              \
            _Pragma("unroll") for (int rt = 0; rt < NRT; ++rt) {                                   \
                if (wlive[rt] && (!ROOT || (KP) < kp_end[rt])) {                                   \
                    _Pragma("unroll") for (int ct = 0; ct < 2; ++ct)                               \
                        acc[rt][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(A_[rt][h], b_[ct][h], acc[rt][ct], 0, 0, 0); \
                }                                                                                  \
            }                                                                                      \
    }
    int n_kp = Kq >> 1;                                           // pairs of steps (even: Kq is a multiple of 4)
    int kp_end[NRT];
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) kp_end[rt] = 0;
    if constexpr (ROOT) {
        const int tri = tri_flag != nullptr ? *tri_flag : 0;
        const int n_kp_all = n_pad >> 3;                          // (n_pad is a multiple of 16)
        n_kp = 0;
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) {
            const int T = tile[rt];
            kp_end[rt] = wlive[rt] ? (tri ? min(n_kp_all, 2 * T + 2) : n_kp_all) : 0;     // k <= 16 T + 15 (the mean's row sits in the last tile)
            n_kp = max(n_kp, kp_end[rt]);
        }
        n_kp = (n_kp + 1) & ~1;                                   // (the loop below takes two pair-steps at a time)
    }
    double a0[NRT][2], a1[NRT][2];
    PF_LOADA(a0, 0)
    for (int kp = 0; kp < n_kp; kp += 2) {
        PF_LOADA(a1, min(kp + 1, n_kp - 1))
        PF_STEP2(a0, kp)
        PF_LOADA(a0, min(kp + 2, n_kp - 1))
        PF_STEP2(a1, kp + 1)
    }
#undef PF_LOADA
#undef PF_STEP2
    // ---- 3. q[c] = sum_r KX[r][c] V[r][c]: accumulator (rt, ct)[reg] is V'[16 (wave + 4 rt) + lk + 4 reg][16 ct + li];
    //      row n_obs is the mean's sum (KX is zero there: it drops out of q by itself)
    double qp[2] = {0.0, 0.0};
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int r = 16 * tile[rt] + lk + 4 * reg;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                if constexpr (ROOT) {
                    const double v = (r < n_obs) ? acc[rt][ct][reg] : 0.0;
                    qp[ct] = fma(v, v, qp[ct]);
                } else {
                    const double kv = (r < n_obs) ? kxt[(size_t)(16 * ct + li) * rs + min(r, n_pad - 1)] : 0.0;
                    qp[ct] = fma(kv, (r < n_obs) ? acc[rt][ct][reg] : 0.0, qp[ct]);
                }
                if (r == n_obs && wlive[rt]) s_mean[16 * ct + li] = acc[rt][ct][reg];
            }
        }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        qp[ct] += __shfl_xor(qp[ct], 16, 64);
        qp[ct] += __shfl_xor(qp[ct], 32, 64);
        if (lk == 0) s_q[wave * PF_NB + 16 * ct + li] = qp[ct];
    }
    __syncthreads();
    if (tid < PF_NB) {
        const int64_t c = c_base + tid;
        if (c < N) {
            const double mean = c0 + s_mean[tid];
            const double q = ((s_q[tid] + s_q[PF_NB + tid]) + s_q[2 * PF_NB + tid]) + s_q[3 * PF_NB + tid];
            double kxx = kxx_const;
            if constexpr (KIND == SOBER_KIND_TANIMOTO) {
                const double n2 = s_yn[tid];
                kxx = (n2 + 1e-6) / (1e-6 + n2) * outputscale;
            }
            const double var = kxx - q + noise;
            if (mean_out) mean_out[c] = mean;
            var_out[c] = var;
            if (lfi_out != nullptr) {
                const double z = (mean - eta) / sqrt(var);
                double pr = 0.5 * erfc(-z * 0.70710678118654752440);
                if (log_flag) pr = log(pr + 1.1920928955078125e-07);     // + torch.finfo().eps (FP32 eps, quirk Q5)
                lfi_out[c] = pr;
            }
        }
    }
}

}  // namespace sober

extern "C" int sober_predict_fused_supported(int kind, int n_obs, int dt) {
    if (n_obs < 2 || n_obs > sober::PF_MAX_OBS) return 0;
    if (kind == SOBER_KIND_RBF || kind == SOBER_KIND_MATERN52)
        return (dt == 4 || dt == 8 || dt == 12 || dt == 16 || dt == 20 || dt == 24 || dt == 32) ? 1 : 0;
    if (kind == SOBER_KIND_TANIMOTO) return (dt == 1 || dt == 2 || dt == 4 || dt == 8 || dt == 16 || dt == 32) ? 1 : 0;
    return 0;
}

static int predict_fused_launch(int kind, const void* obs, const double* obs_norm, int n_obs, const void* cand,
                                const double* cand_norm, int64_t N, int dt, double outputscale, const double* W, int ldw,
                                const double* alpha, double c0, double kxx_const, double noise, double* mean_out,
                                double* var_out, double eta, const double* eta_ptr, double* lfi_out, int log_flag,
                                bool root, const int32_t* tri_flag, void* stream) {
    if (!obs || !cand || !W || !var_out || N <= 0 || ldw < n_obs) return SOBER_E_ARG;
    if (!root) tri_flag = nullptr;
    if (!sober_predict_fused_supported(kind, n_obs, dt)) return SOBER_E_DIM;
    if (kind == SOBER_KIND_TANIMOTO && (!obs_norm || !cand_norm)) return SOBER_E_ARG;
    const size_t bytes = sober::pf_lds_bytes(n_obs, dt);
    if (bytes > 160 * 1024 - 512) return SOBER_E_DIM;
    const dim3 grid((unsigned)((N + sober::PF_NB - 1) / sober::PF_NB));
    hipStream_t st = (hipStream_t)stream;
    // (the dynamic-LDS attribute per instantiation and device: set on every call -- a few hundred ns of host time)
#define PF_ONE(K, D, R, NRT_)                                                                                                  \
    {                                                                                                                          \
        HIP_TRY(hipFuncSetAttribute((const void*)sober::k_predict_fused<K, D, R, NRT_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512)); \
        hipLaunchKernelGGL((sober::k_predict_fused<K, D, R, NRT_>), grid, dim3(256), bytes, st, (const double*)obs, obs_norm, n_obs, \
                           (const double*)cand, cand_norm, N, dt, outputscale, W, ldw, alpha, c0, kxx_const, noise, mean_out,  \
                           var_out, eta, eta_ptr, lfi_out, log_flag, tri_flag);                                                \
    }
    // (eight row tiles per wave beyond 255 observations, four up to there: the accumulators of the tiles a wave does not have
    //  would cost it registers for nothing)
#define PF_LAUNCH(K, D)                                                                                                        \
    case D: {                                                                                                                  \
        if (root) {                                                                                                            \
            if (n_obs <= 255) PF_ONE(K, D, true, 4) else PF_ONE(K, D, true, 8)                                                 \
        } else {                                                                                                               \
            if (n_obs <= 255) PF_ONE(K, D, false, 4) else PF_ONE(K, D, false, 8)                                               \
        }                                                                                                                      \
        break;                                                                                                                 \
    }
#define PF_DIMS(K) switch (dt) { PF_LAUNCH(K, 4) PF_LAUNCH(K, 8) PF_LAUNCH(K, 12) PF_LAUNCH(K, 16) PF_LAUNCH(K, 20) PF_LAUNCH(K, 24) \
                                 PF_LAUNCH(K, 32) default: return SOBER_E_DIM; }
    switch (kind) {
        case SOBER_KIND_RBF: PF_DIMS(SOBER_KIND_RBF) break;
        case SOBER_KIND_MATERN52: PF_DIMS(SOBER_KIND_MATERN52) break;
        default:
            switch (dt) { PF_LAUNCH(SOBER_KIND_TANIMOTO, 1) PF_LAUNCH(SOBER_KIND_TANIMOTO, 2) PF_LAUNCH(SOBER_KIND_TANIMOTO, 4)
                          PF_LAUNCH(SOBER_KIND_TANIMOTO, 8) PF_LAUNCH(SOBER_KIND_TANIMOTO, 16) PF_LAUNCH(SOBER_KIND_TANIMOTO, 32)
                          default: return SOBER_E_DIM; }
            break;
    }
#undef PF_DIMS
#undef PF_LAUNCH
#undef PF_ONE
    LAUNCH_CHECK();
    return 0;
}

extern "C" int sober_predict_fused(int kind, const void* obs, const double* obs_norm, int n_obs, const void* cand,
                                   const double* cand_norm, int64_t N, int dt, double outputscale, const double* W, int ldw,
                                   const double* alpha, double c0, double kxx_const, double noise, double* mean_out,
                                   double* var_out, double eta, const double* eta_ptr, double* lfi_out, int log_flag,
                                   void* stream) {
    return predict_fused_launch(kind, obs, obs_norm, n_obs, cand, cand_norm, N, dt, outputscale, W, ldw, alpha, c0, kxx_const, noise,
                                mean_out, var_out, eta, eta_ptr, lfi_out, log_flag, false, nullptr, stream);
}

// ... with the ROOT of W instead of W: St = S^T (n_obs x n_obs, row stride ldst), W = S S^T; *tri_flag (device, may be NULL = 0)
// != 0 promises that St is lower triangular (gpytorch's covar_cache of an exact GP): the tile products above the diagonal
// are then skipped.  Same outputs; var = kxx - |St k|^2 + noise.
extern "C" int sober_predict_fused_root(int kind, const void* obs, const double* obs_norm, int n_obs, const void* cand,
                                        const double* cand_norm, int64_t N, int dt, double outputscale, const double* St, int ldst,
                                        const int32_t* tri_flag, const double* alpha, double c0, double kxx_const, double noise,
                                        double* mean_out, double* var_out, double eta, const double* eta_ptr, double* lfi_out,
                                        int log_flag, void* stream) {
    return predict_fused_launch(kind, obs, obs_norm, n_obs, cand, cand_norm, N, dt, outputscale, St, ldst, alpha, c0, kxx_const,
                                noise, mean_out, var_out, eta, eta_ptr, lfi_out, log_flag, true, tri_flag, stream);
}
