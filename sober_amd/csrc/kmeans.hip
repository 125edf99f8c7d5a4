// K10: the Nystrom column subsample of continuous priors, KMeans of SOBER/_weights.py:100-126.
// Lloyd's algorithm exactly as the reference runs it: centroids = first K rows, `iters` iterations
// with no convergence test, argmin with first-index tie break (a NaN distance wins, like
// torch.argmin), centroid = cluster sum / count (empty cluster -> 0/0 = NaN).
//
// E step: lanes <-> points (coordinates in VGPRs), centroid tiles broadcast from LDS.
// M step: the points are brought into cluster order by a STABLE radix sort of (label, index) -- ascending indices inside
//         a cluster -- and one workgroup per cluster sums its contiguous segment in a fixed order (no floating-point
//         atomics -> bit-reproducible; the reference's scatter_add_ is sequential too).  The cluster sizes come from
//         integer atomics in the E step.  (Round 1: every cluster's workgroup scanned ALL labels, O(K N): 2.7 ms per
//         iteration at 1M x 20, K = 500.)  Without a workspace the O(K N) form is used.
#include "common.hpp"
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

namespace sober {

constexpr int KM_TILE_BYTES = 48 * 1024;

template <int DT>
__global__ __launch_bounds__(256) void k_kmeans_assign(const double* __restrict__ X, int64_t N, int d,
                                                       const double* __restrict__ cent, int K,
                                                       int32_t* __restrict__ labels, int32_t* __restrict__ counts) {
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
    if (i < N) {
        labels[i] = bi;
        if (counts != nullptr) atomicAdd(counts + bi, 1);              // (integer: the result does not depend on the order)
    }
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

// M step on cluster-sorted points: workgroup k sums rows X[order[lo .. lo + n_k)] (ascending point indices)
template <int DT>
__global__ __launch_bounds__(256) void k_kmeans_update_sorted(const double* __restrict__ X, int d, int K,
                                                              const int32_t* __restrict__ counts,
                                                              const int32_t* __restrict__ order,
                                                              double* __restrict__ cent) {
    __shared__ double s_sum[4][DT + 1];
    __shared__ int s_lo[4];
    const int k = blockIdx.x, tid = threadIdx.x;
    // lo = sum of the sizes of the clusters in front of mine
    int part = 0;
    for (int j = tid; j < k; j += 256) part += counts[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
    if ((tid & 63) == 0) s_lo[tid >> 6] = part;
    __syncthreads();
    const int lo = s_lo[0] + s_lo[1] + s_lo[2] + s_lo[3];
    const int n_k = counts[k];
    double acc[DT + 1];
#pragma unroll
    for (int j = 0; j <= DT; ++j) acc[j] = 0.0;
    for (int p = tid; p < n_k; p += 256) {
        const int64_t i = order[lo + p];
#pragma unroll
        for (int j = 0; j < DT; ++j)
            if (j < d) acc[j] += X[i * d + j];
        acc[DT] += 1.0;
    }
#pragma unroll
    for (int j = 0; j <= DT; ++j) {
        double v = acc[j];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        acc[j] = v;
    }
    __syncthreads();
    if ((tid & 63) == 0) {
#pragma unroll
        for (int j = 0; j <= DT; ++j) s_sum[tid >> 6][j] = acc[j];
    }
    __syncthreads();
    if (tid < d) {
        const int j = tid;
        const double sum = ((s_sum[0][j] + s_sum[1][j]) + s_sum[2][j]) + s_sum[3][j];
        const double cnt = ((s_sum[0][DT] + s_sum[1][DT]) + s_sum[2][DT]) + s_sum[3][DT];
        cent[(size_t)k * d + j] = sum / cnt;
    }
}

__global__ void k_copy_rows(const double* __restrict__ X, int64_t cnt, double* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < cnt) out[t] = X[t];
}

static inline unsigned km_bits(int K) { unsigned b = 1; while ((1u << b) < (unsigned)K) ++b; return b; }
static inline size_t km_sort_bytes(int64_t N, int K) {
    size_t bytes = 0;
    rocprim::counting_iterator<int32_t> ids(0);
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const int32_t*)nullptr, (int32_t*)nullptr, ids, (int32_t*)nullptr,
                                    (size_t)N, 0u, km_bits(K));
    return (bytes + 255) / 256 * 256;
}
// workspace: [counts K int32 | keys_out N int32 | order N int32 | radix sort scratch]
static inline size_t km_off_keys(int K) { return ((size_t)K * 4 + 255) / 256 * 256; }
static inline size_t km_off_order(int64_t N, int K) { return km_off_keys(K) + ((size_t)N * 4 + 255) / 256 * 256; }
static inline size_t km_off_sort(int64_t N, int K) { return km_off_order(N, K) + ((size_t)N * 4 + 255) / 256 * 256; }

template <int DT>
static int run_kmeans(const double* X, int64_t N, int d, int K, int iters, double* cent,
                      int32_t* labels, void* ws, int64_t ws_bytes, hipStream_t st) {
    hipLaunchKernelGGL(k_copy_rows, dim3((unsigned)(((int64_t)K * d + 255) / 256)), dim3(256), 0, st, X,
                       (int64_t)K * d, cent);
    LAUNCH_CHECK();
    const bool sorted = ws != nullptr && N < 0x7fffffffLL &&
                        ws_bytes >= (int64_t)(km_off_sort(N, K) + km_sort_bytes(N, K));
    int32_t* counts = sorted ? (int32_t*)ws : nullptr;
    int32_t* keys_out = sorted ? (int32_t*)((char*)ws + km_off_keys(K)) : nullptr;
    int32_t* order = sorted ? (int32_t*)((char*)ws + km_off_order(N, K)) : nullptr;
    void* scratch = sorted ? (void*)((char*)ws + km_off_sort(N, K)) : nullptr;
    size_t sbytes = sorted ? km_sort_bytes(N, K) : 0;
    for (int it = 0; it < iters; ++it) {
        if (sorted) HIP_TRY(hipMemsetAsync(counts, 0, (size_t)K * 4, st));
        hipLaunchKernelGGL((k_kmeans_assign<DT>), dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, X,
                           N, d, cent, K, labels, counts);
        LAUNCH_CHECK();
        if (sorted) {
            rocprim::counting_iterator<int32_t> ids(0);
            HIP_TRY(rocprim::radix_sort_pairs(scratch, sbytes, (const int32_t*)labels, keys_out, ids, order, (size_t)N, 0u,
                                              km_bits(K), st));
            hipLaunchKernelGGL((k_kmeans_update_sorted<DT>), dim3(K), dim3(256), 0, st, X, d, K, counts, order, cent);
        } else {
            hipLaunchKernelGGL((k_kmeans_update<DT>), dim3(K), dim3(256), 0, st, X, N, d, labels, cent);
        }
        LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace sober

extern "C" int64_t sober_kmeans_ws_bytes(int64_t N, int d, int K) {
    (void)d;
    if (N <= 0 || K <= 0 || N >= 0x7fffffffLL) return 8;
    return (int64_t)(sober::km_off_sort(N, K) + sober::km_sort_bytes(N, K));
}

extern "C" int sober_kmeans_lloyd(const double* X, int64_t N, int d, int K, int iters,
                                  double* centroids, int32_t* labels, void* ws, int64_t ws_bytes,
                                  void* stream) {
    if (!X || !centroids || !labels || N <= 0 || d <= 0 || K <= 0 || K > N || iters < 0)
        return SOBER_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int dt = sober_padded_dim(d);
    switch (dt) {
        case 4: return sober::run_kmeans<4>(X, N, d, K, iters, centroids, labels, ws, ws_bytes, st);
        case 8: return sober::run_kmeans<8>(X, N, d, K, iters, centroids, labels, ws, ws_bytes, st);
        case 12: return sober::run_kmeans<12>(X, N, d, K, iters, centroids, labels, ws, ws_bytes, st);
        case 16: return sober::run_kmeans<16>(X, N, d, K, iters, centroids, labels, ws, ws_bytes, st);
        case 20: return sober::run_kmeans<20>(X, N, d, K, iters, centroids, labels, ws, ws_bytes, st);
        case 24: return sober::run_kmeans<24>(X, N, d, K, iters, centroids, labels, ws, ws_bytes, st);
        case 32: return sober::run_kmeans<32>(X, N, d, K, iters, centroids, labels, ws, ws_bytes, st);
        default: return SOBER_E_DIM;
    }
}
