// K10: the Nystrom column subsample of continuous priors, KMeans of SOBER/_weights.py:100-126.
// Lloyd's algorithm exactly as the reference runs it: centroids = first K rows, `iters` iterations
// with no convergence test, argmin with first-index tie break (a NaN distance wins, like
// torch.argmin), centroid = cluster sum / count (empty cluster -> 0/0 = NaN).
//
// E step (round 3): distances in GEMM form on the FP64 matrix cores, with an exact re-check where it matters.
//         |x - c|^2 - |x|^2 = |c|^2 - 2 x.c is a (d + 1)-term dot product of augmented rows [x, 1] . [-2c, |c|^2]:
//         v_mfma_f64_16x16x4 on 16 centroids x 16 points tiles; every lane keeps the two smallest values (and the index
//         of the smallest) of its point over the centroids it sees.  The label is the reference's argmin of
//         ((x - c)^2).sum(-1) PROVIDED the two smallest are further apart than any rounding of either evaluation can
//         bridge (margin = 2^-40 (|x|^2 + max|c|^2), > 300x the error bounds of both); otherwise -- near-ties, exact
//         ties (duplicated centroids), NaN / Inf anywhere -- the point goes on a list and k_kmeans_assign_list runs the
//         reference-ordered (x - c)^2 arithmetic of k_kmeans_assign on it: labels bit-equal to that kernel's.
//         (k_kmeans_assign: lanes <-> points, centroid tiles broadcast from LDS -- the form without a workspace.)
// M step: the points are brought into cluster order by a STABLE radix sort of (label, index) -- ascending indices inside
//         a cluster -- and one workgroup per cluster sums its contiguous segment in a fixed order (no floating-point
//         atomics -> bit-reproducible; the reference's scatter_add_ is sequential too).  The cluster's segment comes
//         from two binary searches in the sorted labels (no counters, no atomics).  (Round 1: every cluster's workgroup scanned ALL labels, O(K N): 2.7 ms per
//         iteration at 1M x 20, K = 500.)  Without a workspace the O(K N) form is used.
#include "common.hpp"
#include <cstdlib>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

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
    if (i < N) {
        labels[i] = bi;
    }
}

// the same arithmetic for the points on the re-check list (grid-stride over the list)
template <int DT>
__global__ __launch_bounds__(256) void k_kmeans_assign_list(const double* __restrict__ X, int d, const double* __restrict__ cent,
                                                            int K, const int32_t* __restrict__ list,
                                                            const int32_t* __restrict__ n_list, int32_t* __restrict__ labels) {
    const int n = *n_list;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
        const int64_t i = list[t];
        double x[DT];
#pragma unroll
        for (int j = 0; j < DT; ++j) x[j] = (j < d) ? X[i * d + j] : 0.0;
        double best = __builtin_inf();
        int bi = 0;
        bool best_nan = false;
        for (int k = 0; k < K; ++k) {
            double dist = 0.0;
#pragma unroll
            for (int j = 0; j < DT; ++j) {
                const double df = x[j] - ((j < d) ? cent[(size_t)k * d + j] : 0.0);
                dist = fma(df, df, dist);
            }
            const bool isn = dist != dist;
            if (!best_nan && (isn || dist < best)) {
                best = dist;
                bi = k;
                best_nan = isn;
            }
        }
        labels[i] = bi;
    }
}

// augmented centroid rows for the matrix-core E step: Caug[k] = [-2 c_k, |c_k|^2, 0..] (k < K), [0.., 1e300, 0..] for the
// padding rows up to Kp (they never win); meta[0] = max_k |c_k|^2, meta[1] != 0: some centroid is not finite (then every
// point goes to the re-check: the reference's argmin lets the first NaN distance win); the re-check list is emptied.
__global__ __launch_bounds__(256) void k_kmeans_prep(const double* __restrict__ cent, int K, int d, int Kp, int DA,
                                                     double* __restrict__ Caug, double* __restrict__ meta,
                                                     int32_t* __restrict__ n_list) {
    __shared__ double s_max[256];
    __shared__ int s_bad[256];
    double mx = 0.0;
    int bad = 0;
    for (int k = threadIdx.x; k < Kp; k += 256) {
        double n2 = 0.0;
        for (int j = 0; j < d; ++j) {
            const double c = (k < K) ? cent[(size_t)k * d + j] : 0.0;
            Caug[(size_t)k * DA + j] = -2.0 * c;
            n2 = fma(c, c, n2);
        }
        Caug[(size_t)k * DA + d] = (k < K) ? n2 : 1e300;
        for (int j = d + 1; j < DA; ++j) Caug[(size_t)k * DA + j] = 0.0;
        if (k < K) {
            if (!(n2 <= 1e300)) bad = 1;                        // NaN or overflow
            else mx = fmax(mx, n2);
        }
    }
    s_max[threadIdx.x] = mx;
    s_bad[threadIdx.x] = bad;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) {
            s_max[threadIdx.x] = fmax(s_max[threadIdx.x], s_max[threadIdx.x + h]);
            s_bad[threadIdx.x] |= s_bad[threadIdx.x + h];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { meta[0] = s_max[0]; meta[1] = s_bad[0] ? 1.0 : 0.0; *n_list = 0; }
}

#ifndef KM_PB_
#define KM_PB_ 4
#endif
constexpr int KM_PB = KM_PB_;      // 16-point blocks per wave
// v_min_f64 / v_max_f64 as the hardware has them (fmin / fmax put a canonicalising v_max_f64 v, v, v in front of every
// operand that comes out of an MFMA: 24 of the 131 vector instructions per tile); a NaN operand is dropped, which the
// caller wants (it leaves the two smallest equal and sends the point to the re-check)
__device__ __forceinline__ double km_min(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double km_max(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

template <int KT>                  // DA = 4 KT >= d + 1
__global__ __launch_bounds__(256) void k_kmeans_assign_mfma(const double* __restrict__ X, int64_t N, int d,
                                                            const double* __restrict__ Caug, int Kp,
                                                            const double* __restrict__ meta, int32_t* __restrict__ labels,
                                                            int32_t* __restrict__ list,
                                                            int32_t* __restrict__ n_list) {
    typedef double d4 __attribute__((ext_vector_type(4)));
    constexpr int DA = 4 * KT, PB = KM_PB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lj = lane & 15, lg = lane >> 4;
    const int64_t p0 = ((int64_t)blockIdx.x * 4 + wave) * (16 * PB);
    if (p0 >= N) return;
    // B fragments: point p0 + 16 bl + lj, contraction slices kk = lg KT + ks (the same permutation on both operands)
    double b[PB][KT], x2[PB];
#pragma unroll
    for (int bl = 0; bl < PB; ++bl) {
        const int64_t i = min(p0 + 16 * bl + lj, N - 1);
        double part = 0.0;
#pragma unroll
        for (int ks = 0; ks < KT; ++ks) {
            const int kk = lg * KT + ks;
            const double v = (kk < d) ? X[i * d + min(kk, d - 1)] : ((kk == d) ? 1.0 : 0.0);
            b[bl][ks] = v;
            part = (kk < d) ? fma(v, v, part) : part;
        }
        part += __shfl_xor(part, 16, 64);
        part += __shfl_xor(part, 32, 64);
        x2[bl] = part;
    }
    double b1[PB], b2[PB];
    int i1[PB];
#pragma unroll
    for (int bl = 0; bl < PB; ++bl) { b1[bl] = __builtin_inf(); b2[bl] = __builtin_inf(); i1[bl] = 0; }
    const int n_tiles = Kp >> 4;
    // A fragments one tile ahead
    double a[KT], an[KT];
#pragma unroll
    for (int ks = 0; ks < KT; ++ks) a[ks] = Caug[(size_t)lj * DA + lg * KT + ks];
    for (int t = 0; t < n_tiles; ++t) {
        // the next tile's A fragments are requested FIRST (pinned: the compiler otherwise sinks the loads to the end of
        // the body and waits for them there -- an exposed L2 round trip per tile); they are taken over after the body
        const int tn = min(t + 1, n_tiles - 1);
#pragma unroll
        for (int ks = 0; ks < KT; ++ks) an[ks] = Caug[(size_t)(16 * tn + lj) * DA + lg * KT + ks];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int bl = 0; bl < PB; ++bl) {
            d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
#ifdef KM_X_NOMFMA       // timing-only builds (wrong results): what each part of the tile costs
            for (int ks = 0; ks < KT; ++ks) { acc[0] += a[ks]; acc[1] += b[bl][ks]; acc[2] -= a[ks]; acc[3] -= b[bl][ks]; }
#else
            for (int ks = 0; ks < KT; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], b[bl][ks], acc, 0, 0, 0);
#endif
            // lane (lj, lg): centroid rows lg + 4 r of the tile against point lj: the two smallest so far
#ifdef KM_X_NOSEL
            b1[bl] += (acc[0] + acc[1]) + (acc[2] + acc[3]);
#else
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double v = acc[r];
                const bool lt = v < b1[bl];
                b2[bl] = km_min(b2[bl], km_max(b1[bl], v));
                b1[bl] = km_min(b1[bl], v);
                i1[bl] = lt ? (4 * t + r) : i1[bl];
            }
#endif
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < KT; ++ks) a[ks] = an[ks];
    }
    const double cmax2 = meta[0];
    const bool all_recheck = meta[1] != 0.0;
#pragma unroll
    for (int bl = 0; bl < PB; ++bl) {
        // the four lane groups hold disjoint centroid subsets of the same point: merge their two smallest
        double m1 = b1[bl], m2 = b2[bl];
        int k1 = 16 * (i1[bl] >> 2) + lg + 4 * (i1[bl] & 3);
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
            const double o1 = __shfl_xor(m1, o, 64), o2 = __shfl_xor(m2, o, 64);
            const int ok = __shfl_xor(k1, o, 64);
            const bool take = o1 < m1;
            m2 = fmin(fmin(m2, o2), fmax(m1, o1));
            k1 = take ? ok : k1;
            m1 = fmin(m1, o1);
        }
        const int64_t i = p0 + 16 * bl + lj;
        if (lg == 0 && i < N) {
            const double margin = 0x1p-40 * (x2[bl] + cmax2);
            if (!all_recheck && (m2 - m1 > margin)) {       // (NaN anywhere fails this test)
                labels[i] = k1;
            } else {
                list[atomicAdd(n_list, 1)] = (int32_t)i;
            }
        }
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

// M step on cluster-sorted points: workgroup k sums rows X[order[lo .. lo + n_k)] (ascending point indices); the segment
// comes from the sorted labels themselves
template <int DT>
__global__ __launch_bounds__(256) void k_kmeans_update_sorted(const double* __restrict__ X, int d, int K,
                                                              const int32_t* __restrict__ keys, int N,
                                                              const int32_t* __restrict__ order,
                                                              double* __restrict__ cent) {
    __shared__ double s_sum[4][DT + 1];
    const int k = blockIdx.x, tid = threadIdx.x;
    // my segment of the sorted labels: [first position with label >= k, first position with label >= k + 1) -- two
    // binary searches per thread instead of one integer atomic per point in the E step (100k atomics on the 16 cache
    // lines of the counters took ~80 us of every iteration: the L2 serialises them per line)
    int lo = 0, hi = N;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (keys[mid] < k) lo = mid + 1; else hi = mid; }
    int lo2 = lo, hi2 = N;
    while (lo2 < hi2) { const int mid = (lo2 + hi2) >> 1; if (keys[mid] <= k) lo2 = mid + 1; else hi2 = mid; }
    const int n_k = lo2 - lo;
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
// workspace: [counts K int32 | keys_out N int32 | order N int32 | radix sort scratch | matrix-core E step: Caug Kp x 32
//             doubles, meta 2 doubles, list length 1 int32 (+ pad), list N int32]
static inline size_t km_off_keys(int K) { return ((size_t)K * 4 + 255) / 256 * 256; }
static inline size_t km_off_order(int64_t N, int K) { return km_off_keys(K) + ((size_t)N * 4 + 255) / 256 * 256; }
static inline size_t km_off_sort(int64_t N, int K) { return km_off_order(N, K) + ((size_t)N * 4 + 255) / 256 * 256; }
static inline int km_kp(int K) { return (K + 15) / 16 * 16; }
static inline size_t km_off_caug(int64_t N, int K) { return km_off_sort(N, K) + km_sort_bytes(N, K); }
static inline size_t km_off_meta(int64_t N, int K) { return km_off_caug(N, K) + (size_t)km_kp(K) * 32 * 8; }
static inline size_t km_off_list(int64_t N, int K) { return km_off_meta(N, K) + 256; }
static inline size_t km_total(int64_t N, int K) { return km_off_list(N, K) + ((size_t)N * 4 + 255) / 256 * 256; }

template <int DT>
static int run_kmeans(const double* X, int64_t N, int d, int K, int iters, double* cent,
                      int32_t* labels, void* ws, int64_t ws_bytes, hipStream_t st) {
    hipLaunchKernelGGL(k_copy_rows, dim3((unsigned)(((int64_t)K * d + 255) / 256)), dim3(256), 0, st, X,
                       (int64_t)K * d, cent);
    LAUNCH_CHECK();
    const bool sorted = ws != nullptr && N < 0x7fffffffLL && ws_bytes >= (int64_t)km_total(N, K);
    const int kt = (d + 4) / 4;                              // augmented rows [x, 1]: d + 1 entries, 4 per MFMA step
    static const bool valu_e = getenv("SOBER_KMEANS_VALU") != nullptr;     // (same-box A/B of the E step)
    const bool mfma_e = sorted && !valu_e && kt <= 8;
    double* Caug = sorted ? (double*)((char*)ws + km_off_caug(N, K)) : nullptr;
    double* meta = sorted ? (double*)((char*)ws + km_off_meta(N, K)) : nullptr;
    int32_t* n_list = sorted ? (int32_t*)((char*)ws + km_off_meta(N, K) + 64) : nullptr;
    int32_t* list = sorted ? (int32_t*)((char*)ws + km_off_list(N, K)) : nullptr;
    int32_t* keys_out = sorted ? (int32_t*)((char*)ws + km_off_keys(K)) : nullptr;
    int32_t* order = sorted ? (int32_t*)((char*)ws + km_off_order(N, K)) : nullptr;
    void* scratch = sorted ? (void*)((char*)ws + km_off_sort(N, K)) : nullptr;
    size_t sbytes = sorted ? km_sort_bytes(N, K) : 0;
    for (int it = 0; it < iters; ++it) {
        if (mfma_e) {
            const int Kp = km_kp(K);
            hipLaunchKernelGGL(k_kmeans_prep, dim3(1), dim3(256), 0, st, cent, K, d, Kp, 4 * kt, Caug, meta, n_list);
            LAUNCH_CHECK();
            const int64_t n_waves = (N + 16 * KM_PB - 1) / (16 * KM_PB);
            const dim3 grid((unsigned)((n_waves + 3) / 4));
#define KM_CASE(T) case T: hipLaunchKernelGGL((k_kmeans_assign_mfma<T>), grid, dim3(256), 0, st, X, N, d, Caug, Kp, meta, \
                                              labels, list, n_list); break;
            switch (kt) { KM_CASE(1) KM_CASE(2) KM_CASE(3) KM_CASE(4) KM_CASE(5) KM_CASE(6) KM_CASE(7) KM_CASE(8) default: break; }
#undef KM_CASE
            LAUNCH_CHECK();
            hipLaunchKernelGGL((k_kmeans_assign_list<DT>), dim3(256), dim3(256), 0, st, X, d, cent, K, list, n_list,
                               labels);
        } else {
            hipLaunchKernelGGL((k_kmeans_assign<DT>), dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, X,
                               N, d, cent, K, labels);
        }
        LAUNCH_CHECK();
        if (sorted) {
            rocprim::counting_iterator<int32_t> ids(0);
            HIP_TRY(rocprim::radix_sort_pairs(scratch, sbytes, (const int32_t*)labels, keys_out, ids, order, (size_t)N, 0u,
                                              km_bits(K), st));
            hipLaunchKernelGGL((k_kmeans_update_sorted<DT>), dim3(K), dim3(256), 0, st, X, d, K, keys_out, (int)N, order, cent);
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
    return (int64_t)sober::km_total(N, K);
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
