// K1-K3 for a kernel matrix that is already resident in HBM (row f4 and the callable-kernel protocol of
// SOBER/_rchq.py:9,20): Kmat[c][r] = k(x_r, cand_c), candidate-major (one candidate = one contiguous
// row of n_rows doubles, 4 KB at N_nys = 500).  A level is then a weighted gather-sum
//
//     G[r][s] = sum_e Kmat[idx[e*S + s]][r] * mu[idx[e*S + s]]                     (SOBER/_rchq.py:124-150)
//
// which is HBM-bound (8 B read per kernel entry, no arithmetic to speak of): one WAVE per set s, lanes
// across the rows r, so every load is a fully coalesced 512 B segment; the element loop is unrolled so
// that several candidate rows are in flight per wave.  Same partial-sum contract as sober_level_reduce:
// fixed summation order (e ascending inside a chunk, chunks added by sober_sum_partials), no atomics.
//
// k_gspace_finish is the epilogue of BASQ's g-space kernel (SOBER/BASQ/_scale_mmlt.py:256-275):
//     Kg[c][r] = mu_g(cand_c) mu_g(x_r) (exp(C_h) - 1),   C_h = K[c][r] - corr[c][r]
// with K = os*k(cand, X_nys) and corr = k(cand, X) W k(X, X_nys) produced by sober_pairwise / sober_dgemm.
#include "common.hpp"

namespace sober {

constexpr int LG_WAVES = 4;

template <int KR>      // rows per lane: n_rows <= 64 * KR
__global__ __launch_bounds__(LG_WAVES * 64) void k_level_gather(
    const double* __restrict__ Kmat, int n_rows, int64_t ldk,
    const int32_t* __restrict__ idx, int64_t pos0, int64_t count, int S,
    const double* __restrict__ mu, const double* __restrict__ wmul,
    int64_t e_first, int e_total, int e_per_chunk,
    double* __restrict__ partG, int ldg, int col0,
    double* __restrict__ partTot, int64_t tot_limit) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = blockIdx.x * LG_WAVES + wave;
    const int chunk = blockIdx.y;
    if (s >= S) return;                                  // whole wave
    const int e0 = chunk * e_per_chunk;
    const int e1 = min(e0 + e_per_chunk, e_total);
    double acc[KR];
#pragma unroll
    for (int k = 0; k < KR; ++k) acc[k] = 0.0;
    double tot = 0.0;
    constexpr int UN = 4;
    for (int e = e0; e < e1; e += UN) {
        double w[UN];
        const double* row[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int64_t p = (e_first + e + u) * (int64_t)S + s;
            const bool ok = (e + u < e1) && (p >= pos0) && (p < pos0 + count);
            const int c = ok ? idx[p - pos0] : 0;
            const double m = ok ? mu[c] : 0.0;
            w[u] = (ok && wmul) ? m * wmul[c] : m;
            if (ok && p < tot_limit) tot += m;
            row[u] = Kmat + (size_t)c * ldk;
        }
        double v[UN][KR];
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int k = 0; k < KR; ++k) {
                const int r = lane + 64 * k;
                v[u][k] = (r < n_rows) ? row[u][r] : 0.0;
            }
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int k = 0; k < KR; ++k) acc[k] = fma(v[u][k], w[u], acc[k]);
    }
#pragma unroll
    for (int k = 0; k < KR; ++k) {
        const int r = lane + 64 * k;
        if (r < n_rows) partG[((size_t)chunk * n_rows + r) * ldg + col0 + s] = acc[k];
    }
    if (partTot != nullptr && lane == 0) partTot[(size_t)chunk * ldg + col0 + s] = tot;
}

__global__ void k_gspace_finish(double* __restrict__ K, const double* __restrict__ corr, int64_t n, int m,
                                int64_t ldk, int64_t ldc, const double* __restrict__ mug_cand,
                                const double* __restrict__ mug_rows) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * m) return;
    const int64_t c = t / m;
    const int r = (int)(t - c * m);
    const double ch = K[c * ldk + r] - corr[c * ldc + r];
    K[c * ldk + r] = mug_rows[r] * mug_cand[c] * (exp(ch) - 1.0);           // cov.exp() - 1 as written (:269)
}

template <int KR>
static int launch_lg(const double* Kmat, int n_rows, int64_t ldk, const int32_t* idx, int64_t pos0, int64_t count,
                     int S, const double* mu, const double* wmul, int n_chunks, double* partG, int ldg, int col0,
                     double* partTot, int64_t tot_limit, hipStream_t st) {
    const int64_t e_first = pos0 / S;
    const int e_total = (int)((pos0 + count + S - 1) / S - e_first);
    const int e_per_chunk = (e_total + n_chunks - 1) / n_chunks;
    dim3 grid((S + LG_WAVES - 1) / LG_WAVES, n_chunks);
    hipLaunchKernelGGL((k_level_gather<KR>), grid, dim3(LG_WAVES * 64), 0, st, Kmat, n_rows, ldk, idx, pos0, count,
                       S, mu, wmul, e_first, e_total, e_per_chunk, partG, ldg, col0, partTot, tot_limit);
    LAUNCH_CHECK();
    return 0;
}

}  // namespace sober

using namespace sober;

extern "C" int sober_level_gather(const double* Kmat, int n_rows, int64_t ldk, const int32_t* idx, int64_t pos0,
                                  int64_t count, int S, const double* mu, const double* wmul, int n_chunks,
                                  double* partG, int ldg, int col0, double* partTot, int64_t tot_limit,
                                  void* stream) {
    if (!Kmat || !idx || !mu || !partG) return SOBER_E_ARG;
    if (n_rows <= 0 || ldk < n_rows || pos0 < 0 || count <= 0 || S <= 0 || n_chunks <= 0 || ldg < col0 + S)
        return SOBER_E_ARG;
    if (n_chunks > (pos0 + count + S - 1) / S - pos0 / S) return SOBER_E_ARG;
    if (n_rows > 1024) return SOBER_E_DIM;
    hipStream_t st = (hipStream_t)stream;
#define LG_CASE(KR)                                                                                              \
    return launch_lg<KR>(Kmat, n_rows, ldk, idx, pos0, count, S, mu, wmul, n_chunks, partG, ldg, col0, partTot, \
                         tot_limit, st)
    const int kr = (n_rows + 63) / 64;
    if (kr <= 1) LG_CASE(1);
    if (kr <= 2) LG_CASE(2);
    if (kr <= 4) LG_CASE(4);
    if (kr <= 8) LG_CASE(8);
    if (kr <= 12) LG_CASE(12);
    LG_CASE(16);
#undef LG_CASE
}

extern "C" int sober_gspace_finish(double* K, const double* corr, int64_t n, int m, int64_t ldk, int64_t ldc,
                                   const double* mug_cand, const double* mug_rows, void* stream) {
    if (!K || !corr || !mug_cand || !mug_rows || n <= 0 || m <= 0 || ldk < m || ldc < m) return SOBER_E_ARG;
    hipLaunchKernelGGL(k_gspace_finish, dim3((unsigned)((n * m + 255) / 256)), dim3(256), 0, (hipStream_t)stream, K,
                       corr, n, m, ldk, ldc, mug_cand, mug_rows);
    LAUNCH_CHECK();
    return 0;
}
