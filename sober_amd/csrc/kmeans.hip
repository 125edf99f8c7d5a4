// K10: the Nystrom column subsample of continuous priors, KMeans of SOBER/_weights.py:100-126.
// Lloyd's algorithm exactly as the reference runs it: centroids = first K rows, `iters` iterations
// with no convergence test, argmin with first-index tie break (a NaN distance wins, like
// torch.argmin), centroid = cluster sum / count (empty cluster -> 0/0 = NaN).
//
// E step: lanes <-> points (coordinates in VGPRs), centroid tiles broadcast from LDS.
// M step: one workgroup per cluster scans the labels and sums its members in a fixed order
//         (no atomics -> bit-reproducible; the reference's scatter_add_ is sequential too).
#include "common.hpp"

namespace sober {

constexpr int KM_TILE_BYTES = 48 * 1024;

template <int DT>
__global__ __launch_bounds__(256) void k_kmeans_assign(const double* __restrict__ X, int64_t N, int d,
                                                       const double* __restrict__ cent, int K,
                                                       int32_t* __restrict__ labels) {
    constexpr int KT = KM_TILE_BYTES / (DT * 8);
    __shared__ double s_c[KT][DT];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double x[DT];
#pragma unroll
    for (int j = 0; j < DT; ++j) x[j] = (i < N && j < d) ? X[i * d + j] : 0.0;

    double best = __builtin_inf();
    int bi = 0;
    bool best_nan = false;
    for (int k0 = 0; k0 < K; k0 += KT) {
        const int cnt = min(KT, K - k0);
        __syncthreads();
        for (int t = threadIdx.x; t < cnt * DT; t += blockDim.x) {
            const int kk = t / DT, j = t % DT;
            s_c[kk][j] = (j < d) ? cent[(size_t)(k0 + kk) * d + j] : 0.0;
        }
        __syncthreads();
        for (int kk = 0; kk < cnt; ++kk) {
            double dist = 0.0;
#pragma unroll
            for (int j = 0; j < DT; ++j) {
                const double df = x[j] - s_c[kk][j];
                dist = fma(df, df, dist);
            }
            const bool isn = dist != dist;
            if (!best_nan && (isn || dist < best)) {
                best = dist;
                bi = k0 + kk;
                best_nan = isn;
            }
        }
    }
    if (i < N) labels[i] = bi;
}

template <int DT>
__global__ __launch_bounds__(256) void k_kmeans_update(const double* __restrict__ X, int64_t N, int d,
                                                       const int32_t* __restrict__ labels,
                                                       double* __restrict__ cent) {
    __shared__ double s_sum[4][DT + 1];
    const int k = blockIdx.x;
    double acc[DT + 1];                       // [DT] = member count
#pragma unroll
    for (int j = 0; j <= DT; ++j) acc[j] = 0.0;
    for (int64_t i = threadIdx.x; i < N; i += 256) {
        if (labels[i] == k) {
#pragma unroll
            for (int j = 0; j < DT; ++j)
                if (j < d) acc[j] += X[i * d + j];
            acc[DT] += 1.0;
        }
    }
    // fixed-order butterfly inside each wave, then the four waves in order
#pragma unroll
    for (int j = 0; j <= DT; ++j) {
        double v = acc[j];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        acc[j] = v;
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int j = 0; j <= DT; ++j) s_sum[threadIdx.x >> 6][j] = acc[j];
    }
    __syncthreads();
    if (threadIdx.x < d) {
        const int j = threadIdx.x;
        const double sum = ((s_sum[0][j] + s_sum[1][j]) + s_sum[2][j]) + s_sum[3][j];
        const double cnt = ((s_sum[0][DT] + s_sum[1][DT]) + s_sum[2][DT]) + s_sum[3][DT];
        cent[(size_t)k * d + j] = sum / cnt;
    }
}

__global__ void k_copy_rows(const double* __restrict__ X, int64_t cnt, double* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < cnt) out[t] = X[t];
}

template <int DT>
static int run_kmeans(const double* X, int64_t N, int d, int K, int iters, double* cent,
                      int32_t* labels, hipStream_t st) {
    hipLaunchKernelGGL(k_copy_rows, dim3((unsigned)(((int64_t)K * d + 255) / 256)), dim3(256), 0, st, X,
                       (int64_t)K * d, cent);
    LAUNCH_CHECK();
    for (int it = 0; it < iters; ++it) {
        hipLaunchKernelGGL((k_kmeans_assign<DT>), dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, X,
                           N, d, cent, K, labels);
        LAUNCH_CHECK();
        hipLaunchKernelGGL((k_kmeans_update<DT>), dim3(K), dim3(256), 0, st, X, N, d, labels, cent);
        LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace sober

extern "C" int64_t sober_kmeans_ws_bytes(int64_t N, int d, int K) {
    (void)N; (void)d; (void)K;
    return 8;   // the deterministic M step needs no scratch; kept in the ABI for future variants
}

extern "C" int sober_kmeans_lloyd(const double* X, int64_t N, int d, int K, int iters,
                                  double* centroids, int32_t* labels, void* ws, int64_t ws_bytes,
                                  void* stream) {
    (void)ws; (void)ws_bytes;
    if (!X || !centroids || !labels || N <= 0 || d <= 0 || K <= 0 || K > N || iters < 0)
        return SOBER_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int dt = sober_padded_dim(d);
    switch (dt) {
        case 4: return sober::run_kmeans<4>(X, N, d, K, iters, centroids, labels, st);
        case 8: return sober::run_kmeans<8>(X, N, d, K, iters, centroids, labels, st);
        case 12: return sober::run_kmeans<12>(X, N, d, K, iters, centroids, labels, st);
        case 16: return sober::run_kmeans<16>(X, N, d, K, iters, centroids, labels, st);
        case 20: return sober::run_kmeans<20>(X, N, d, K, iters, centroids, labels, st);
        case 24: return sober::run_kmeans<24>(X, N, d, K, iters, centroids, labels, st);
        case 32: return sober::run_kmeans<32>(X, N, d, K, iters, centroids, labels, st);
        default: return SOBER_E_DIM;
    }
}
