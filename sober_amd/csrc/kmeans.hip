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
//         ties (duplicated centroids), Inf anywhere -- the lane runs the reference-ordered (x - c)^2 arithmetic of
//         k_kmeans_assign inline (a handful of points per million on continuous data).  A NaN centroid (an empty
//         cluster's 0 / 0) gives every point a NaN distance: the first such index is every finite point's label, as with
//         torch.argmin.  Labels bit-equal to k_kmeans_assign's.  The augmented rows of the NEXT E step are written by
//         the M step itself (one launch less per iteration).
//         (k_kmeans_assign: lanes <-> points, centroid tiles broadcast from LDS -- the form without a workspace.)
// M step: the points are brought into cluster order by a STABLE counting sort of (label, index) -- ascending indices
//         inside a cluster -- and one workgroup per cluster sums its contiguous segment in a fixed order (no
//         floating-point atomics -> bit-reproducible; the reference's scatter_add_ is sequential too).  The sort is three
//         launches, deterministic without any atomic (round 3; rocPRIM's radix sort took ten launches per iteration,
//         54 us at 100k points, 184 us at 1M): every WAVE counts its 256 points per label (k_km_count: the lanes of one
//         label found with ballots, in lane order), one workgroup per label scans the waves' counts (k_km_scan), and
//         the waves place their points (k_km_place: cluster offset + the wave's offset + rank inside the wave).
//         (Round 1: every cluster's workgroup scanned ALL labels, O(K N): 2.7 ms per iteration at 1M x 20, K = 500.)
//         Without a workspace (or K > 4096) the O(K N) form is used.
#include "common.hpp"
#include <cstdlib>
#include <cstring>

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

// augmented centroid rows for the matrix-core E step: Caug[k] = [-2 c_k, |c_k|^2, 0..] (k < K), [0.., 1e300, 0..] for the
// padding rows up to Kp (they never win).  Runs once, for the initial centroids; afterwards the M step writes the row of
// the centroid it has just computed (k_kmeans_update_sorted).
__global__ __launch_bounds__(256) void k_kmeans_prep(const double* __restrict__ cent, int K, int d, int Kp, int DA,
                                                     double* __restrict__ Caug) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= Kp) return;
    double n2 = 0.0;
    for (int j = 0; j < d; ++j) {
        const double c = (k < K) ? cent[(size_t)k * d + j] : 0.0;
        Caug[(size_t)k * DA + j] = -2.0 * c;
        n2 = fma(c, c, n2);
    }
    Caug[(size_t)k * DA + d] = (k < K) ? n2 : 1e300;
    for (int j = d + 1; j < DA; ++j) Caug[(size_t)k * DA + j] = 0.0;
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

__device__ __forceinline__ void km_round_groups(int label, int lane, int& rank, int& cnt, bool& leader);

// ucount != NULL: the workgroup also COUNTS its 256 points per label (the counting sort's first pass, see below: unit =
// workgroup; every wave counts its 64 points into its own K counters in dynamic LDS -- 4 K ints --, the four are added
// at the end): the labels never travel to a counting kernel and back
template <int KT>                  // DA = 4 KT >= d + 1
__global__ __launch_bounds__(256) void k_kmeans_assign_mfma(const double* __restrict__ X, int64_t N, int d,
                                                            const double* __restrict__ cent,
                                                            const double* __restrict__ Caug, int K, int Kp,
                                                            int32_t* __restrict__ labels,
                                                            int32_t* __restrict__ ucount, int64_t n_units) {
    extern __shared__ int km_hist[];
    typedef double d4 __attribute__((ext_vector_type(4)));
    constexpr int DA = 4 * KT, PB = KM_PB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lj = lane & 15, lg = lane >> 4;
    const int64_t p0 = ((int64_t)blockIdx.x * 4 + wave) * (16 * PB);
    if (p0 >= N && ucount == nullptr) return;               // (a counting workgroup keeps all its waves for the final sum)
    int* const hist = km_hist + wave * K;
    if (ucount != nullptr)
        for (int k = lane; k < K; k += 64) hist[k] = 0;     // (my own counters: no barrier needed before I use them)
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
    // |c|^2 sits in slice kk = d of the A fragments: the lanes that hold it keep max |c|^2 (the margin's scale), the first
    // centroid whose |c|^2 is NaN (a NaN centroid -- an empty cluster's 0 / 0 -- gives every point a NaN distance, and the
    // reference's argmin returns the first NaN) and whether any is infinite (then everything is re-checked)
    const int ksn = d - lg * KT;                             // my slice index of kk = d, if 0 <= ksn < KT
    double cmax2 = 0.0;
    int first_nan = 0x7fffffff;
    bool any_inf = false;
    // A fragments one tile ahead
    double a[KT], an[KT];
#pragma unroll
    for (int ks = 0; ks < KT; ++ks) a[ks] = Caug[(size_t)lj * DA + lg * KT + ks];
    for (int t = 0; t < n_tiles; ++t) {
#pragma unroll
        for (int ks = 0; ks < KT; ++ks) {
            if (ks == ksn && 16 * t + lj < K) {
                const double n2 = a[ks];
                first_nan = (n2 != n2) ? min(first_nan, 16 * t + lj) : first_nan;
                any_inf |= n2 > 1e300;
                cmax2 = (n2 <= 1e300) ? fmax(cmax2, n2) : cmax2;
            }
        }
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
            for (int ks = 0; ks < KT; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], b[bl][ks], acc, 0, 0, 0);
            // lane (lj, lg): centroid rows lg + 4 r of the tile against point lj: the two smallest so far
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double v = acc[r];
                const bool lt = v < b1[bl];
                b2[bl] = km_min(b2[bl], km_max(b1[bl], v));
                b1[bl] = km_min(b1[bl], v);
                i1[bl] = lt ? (4 * t + r) : i1[bl];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < KT; ++ks) a[ks] = an[ks];
    }
    // wave-wide: the scale, the first NaN centroid, any infinite one
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        cmax2 = fmax(cmax2, __shfl_xor(cmax2, o, 64));
        first_nan = min(first_nan, __shfl_xor(first_nan, o, 64));
    }
    const bool all_recheck = __ballot(any_inf) != 0ull;
    int lab[PB];
#pragma unroll
    for (int bl = 0; bl < PB; ++bl) {
        lab[bl] = -1;
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
            const bool x_nan = x2[bl] != x2[bl];
            if (x_nan) {                                     // every distance is NaN: the first index
                lab[bl] = 0;
            } else if (first_nan != 0x7fffffff && !all_recheck && x2[bl] <= 1e300) {
                lab[bl] = first_nan;                         // finite point, NaN centroid: that distance is the first NaN
            } else if (!all_recheck && (m2 - m1 > margin)) { // (an infinite coordinate fails this test)
                lab[bl] = k1;
            } else {
                // near-tie, exact tie or non-finite data: the reference-ordered (x - c)^2 arithmetic of k_kmeans_assign,
                // inline (rare: a handful of points per million on continuous data)
                double best = __builtin_inf();
                int bi = 0;
                bool best_nan = false;
                for (int k = 0; k < K; ++k) {
                    double dist = 0.0;
                    for (int j = 0; j < d; ++j) {
                        const double df = X[i * d + j] - cent[(size_t)k * d + j];
                        dist = fma(df, df, dist);
                    }
                    const bool isn = dist != dist;
                    if (!best_nan && (isn || dist < best)) {
                        best = dist;
                        bi = k;
                        best_nan = isn;
                    }
                }
                lab[bl] = bi;
            }
            labels[i] = lab[bl];
        }
    }
    if (ucount == nullptr) return;
    // point p0 + L's label into lane L (it sits in lane L & 15 of block L >> 4), then the lanes of one label are found
    // with ballots and their number goes to the wave's counter of that label
    int l64 = -1;
#pragma unroll
    for (int bl = 0; bl < PB; ++bl) {
        const int v = __shfl(lab[bl], lane & 15, 64);
        l64 = ((lane >> 4) == bl) ? v : l64;
    }
    int rank, cnt;
    bool leader;
    km_round_groups(l64, lane, rank, cnt, leader);
    if (leader) hist[l64] += cnt;
    __syncthreads();
    if (blockIdx.x < n_units)
        for (int k = threadIdx.x; k < K; k += 256)
            ucount[(size_t)k * n_units + blockIdx.x] = (km_hist[k] + km_hist[K + k]) + (km_hist[2 * K + k] + km_hist[3 * K + k]);
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

// ---- stable counting sort of the points by label, no atomics (see the header) --------------------------------------
// A wave owns 256 consecutive points (unit u = 4 * workgroup + wave), four rounds of 64.  hist = the wave's own K
// counters in LDS.  The lanes that share a label are found with ballots; within a round they are ranked in lane order,
// across rounds by the running counter: the order inside a cluster is ascending point index.
constexpr int KM_UNIT = 256;
// one round of 64 points: rank of every lane among the lanes of its label (lane order), the group's size, its leader --
// ballots only, nothing touches memory inside the loop (one trip per distinct label of the round)
__device__ __forceinline__ void km_round_groups(int label, int lane, int& rank, int& cnt, bool& leader) {
    const unsigned long long below = (1ull << lane) - 1ull;
    unsigned long long todo = __ballot(label >= 0);
    rank = 0; cnt = 0; leader = false;
    while (todo != 0ull) {                                   // (uniform)
        const int src = __ffsll((long long)todo) - 1;
        const int l = __builtin_amdgcn_readlane(label, src);
        const unsigned long long m = __ballot(label == l);
        const bool mine = label == l;
        rank = mine ? __popcll(m & below) : rank;
        cnt = mine ? __popcll(m) : cnt;
        leader = mine ? (lane == src) : leader;
        todo &= ~m;
    }
}
// hist[k]: the wave's running position for label k (counting: starts at 0; placing: starts at the cluster's offset + the
// wave's offset inside the cluster).  Per round ONE gather read and one scatter write by the group leaders.
template <bool PLACE>
__device__ __forceinline__ void km_wave_pass(const int32_t* __restrict__ labels, int64_t N, int64_t unit, int* hist,
                                             int32_t* __restrict__ order) {
    volatile int* hv = hist;                                 // (the leaders write what the whole wave reads next round)
    const int lane = threadIdx.x & 63;
    for (int r = 0; r < KM_UNIT / 64; ++r) {
        const int64_t i = unit * KM_UNIT + r * 64 + lane;
        const int label = (i < N) ? labels[i] : -1;
        int rank, cnt;
        bool leader;
        km_round_groups(label, lane, rank, cnt, leader);
        const int before = (label >= 0) ? hv[label] : 0;
        if (PLACE) {
            if (label >= 0) order[before + rank] = (int32_t)i;
        }
        if (leader) hv[label] = before + cnt;                // (after every read of the round: a wave's LDS operations are in order)
    }
}

__global__ __launch_bounds__(256) void k_km_count(const int32_t* __restrict__ labels, int64_t N, int K, int64_t n_units,
                                                  int32_t* __restrict__ ucount) {
    extern __shared__ int km_hist[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int* hist = km_hist + wave * K;
    for (int k = lane; k < K; k += 64) hist[k] = 0;
    const int64_t unit = (int64_t)blockIdx.x * 4 + wave;
    if (unit >= n_units) return;
    km_wave_pass<false>(labels, N, unit, hist, nullptr);
    for (int k = lane; k < K; k += 64) ucount[(size_t)k * n_units + unit] = hist[k];
}

// workgroup k: exclusive scan of ucount[k][0 .. n_units) in place (-> the unit's offset inside cluster k), tot[k] = sum
__global__ __launch_bounds__(256) void k_km_scan(int32_t* __restrict__ ucount, int64_t n_units, int32_t* __restrict__ tot) {
    __shared__ int s_part[256];
    __shared__ int s_carry;
    // (counts kept label by label, [K][n_units]; unit by unit -- contiguous for the counting and placing passes, a
    //  strided walk here -- measured slower: 1.05 vs 1.02 ms at 100k x 10, 7.0 vs 6.8 ms at 1M x 20)
    int32_t* row = ucount + (size_t)blockIdx.x * n_units;
    const int tid = threadIdx.x;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < n_units; base += 1024) {
        int v[4], sum = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t u = base + 4 * tid + j;
            v[j] = (u < n_units) ? row[u] : 0;
            sum += v[j];
        }
        s_part[tid] = sum;
        __syncthreads();
        for (int h = 1; h < 256; h <<= 1) {                  // inclusive scan of the 256 thread sums
            const int t = (tid >= h) ? s_part[tid - h] : 0;
            __syncthreads();
            s_part[tid] += t;
            __syncthreads();
        }
        int run = s_carry + s_part[tid] - sum;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t u = base + 4 * tid + j;
            if (u < n_units) row[u] = run;
            run += v[j];
        }
        __syncthreads();
        if (tid == 255) s_carry += s_part[255];
        __syncthreads();
    }
    if (tid == 0) tot[blockIdx.x] = s_carry;
}

__global__ __launch_bounds__(256) void k_km_place(const int32_t* __restrict__ labels, int64_t N, int K, int64_t n_units,
                                                  const int32_t* __restrict__ uoff, const int32_t* __restrict__ tot,
                                                  int32_t* __restrict__ order) {
    extern __shared__ int km_hist[];                         // [4][K] counters, then [K] cluster offsets
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int* coff = km_hist + 4 * K;
    int* hist = km_hist + wave * K;
    // cluster offsets = exclusive scan of the K totals (every workgroup for itself: K is a few hundred)
    if (wave == 0) {
        int carry = 0;
        for (int k0 = 0; k0 < K; k0 += 64) {
            const int k = k0 + lane;
            const int t = (k < K) ? tot[k] : 0;
            int incl = t;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int up = __shfl_up(incl, o, 64);
                if (lane >= o) incl += up;
            }
            if (k < K) coff[k] = carry + incl - t;
            carry += __shfl(incl, 63, 64);
        }
    }
    __syncthreads();
    const int64_t unit = (int64_t)blockIdx.x * 4 + wave;
    if (unit >= n_units) return;
    // where this wave's points of cluster k start: K independent loads up front, nothing global inside the pass
    for (int k = lane; k < K; k += 64) hist[k] = coff[k] + uoff[(size_t)k * n_units + unit];
    km_wave_pass<true>(labels, N, unit, hist, order);
}

// M step on cluster-sorted points: workgroup k sums rows X[order[lo .. lo + n_k)] (ascending point indices)
template <int DT>
__global__ __launch_bounds__(256) void k_kmeans_update_sorted(const double* __restrict__ X, int d, int K,
                                                              const int32_t* __restrict__ tot,
                                                              const int32_t* __restrict__ order,
                                                              double* __restrict__ cent, double* __restrict__ Caug, int DA) {
    __shared__ double s_sum[4][DT + 1];
    __shared__ double s_c[DT];
    const int k = blockIdx.x, tid = threadIdx.x;
    // lo = sum of the sizes of the clusters in front of mine
    __shared__ int s_lo[4];
    int part = 0;
    for (int j = tid; j < k; j += 256) part += tot[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
    if ((tid & 63) == 0) s_lo[tid >> 6] = part;
    __syncthreads();
    const int lo = s_lo[0] + s_lo[1] + s_lo[2] + s_lo[3];
    const int n_k = tot[k];
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
        const double c = sum / cnt;
        cent[(size_t)k * d + j] = c;
        if (Caug != nullptr) { Caug[(size_t)k * DA + j] = -2.0 * c; s_c[j] = c; }
    }
    if (Caug == nullptr) return;                             // (uniform)
    __syncthreads();
    // the next E step's augmented row of this centroid (k_kmeans_prep's arithmetic: |c|^2 summed in index order)
    if (tid == 0) {
        double n2 = 0.0;
        for (int j = 0; j < d; ++j) n2 = fma(s_c[j], s_c[j], n2);
        Caug[(size_t)k * DA + d] = n2;
    }
    if (tid > d && tid < DA) Caug[(size_t)k * DA + tid] = 0.0;
}

__global__ void k_copy_rows(const double* __restrict__ X, int64_t cnt, double* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < cnt) out[t] = X[t];
}

// workspace: [tot K int32 | order N int32 | ucount K x n_units int32 | matrix-core E step: Caug Kp x 32 doubles]
constexpr int KM_MAX_K_SORT = 4096;                          // (4 waves' counters + the offsets in LDS)
static inline size_t km_al(size_t b) { return (b + 255) / 256 * 256; }
static inline int64_t km_units(int64_t N) { return (N + KM_UNIT - 1) / KM_UNIT; }
static inline int km_kp(int K) { return (K + 15) / 16 * 16; }
static inline size_t km_off_order(int K) { return km_al((size_t)K * 4); }
static inline size_t km_off_ucount(int64_t N, int K) { return km_off_order(K) + km_al((size_t)N * 4); }
static inline size_t km_off_caug(int64_t N, int K) { return km_off_ucount(N, K) + km_al((size_t)K * km_units(N) * 4); }
static inline size_t km_total(int64_t N, int K) { return km_off_caug(N, K) + (size_t)km_kp(K) * 32 * 8; }

template <int DT>
static int run_kmeans(const double* X, int64_t N, int d, int K, int iters, double* cent,
                      int32_t* labels, void* ws, int64_t ws_bytes, hipStream_t st) {
    hipLaunchKernelGGL(k_copy_rows, dim3((unsigned)(((int64_t)K * d + 255) / 256)), dim3(256), 0, st, X,
                       (int64_t)K * d, cent);
    LAUNCH_CHECK();
    const bool sorted = ws != nullptr && N < 0x7fffffffLL && K <= KM_MAX_K_SORT && ws_bytes >= (int64_t)km_total(N, K);
    const int kt = (d + 4) / 4;                              // augmented rows [x, 1]: d + 1 entries, 4 per MFMA step
    const bool mfma_e = sorted && kt <= 8;
    double* Caug = mfma_e ? (double*)((char*)ws + km_off_caug(N, K)) : nullptr;
    const int Kp = km_kp(K);
    if (mfma_e) {
        hipLaunchKernelGGL(k_kmeans_prep, dim3((unsigned)((Kp + 255) / 256)), dim3(256), 0, st, cent, K, d, Kp, 4 * kt, Caug);
        LAUNCH_CHECK();
    }
    int32_t* tot = sorted ? (int32_t*)ws : nullptr;
    int32_t* order = sorted ? (int32_t*)((char*)ws + km_off_order(K)) : nullptr;
    int32_t* ucount = sorted ? (int32_t*)((char*)ws + km_off_ucount(N, K)) : nullptr;
    const int64_t n_units = km_units(N);
    const size_t lds_count = (size_t)4 * K * sizeof(int), lds_place = (size_t)5 * K * sizeof(int);
    const bool fuse_count = mfma_e && sorted && lds_count <= 48 * 1024;   // the E step counts its own labels
    if (sorted && lds_place > 48 * 1024) {
        static std::atomic<unsigned long long> attr_set{0};
        if (sober_attr_needed(attr_set)) {
            HIP_TRY(hipFuncSetAttribute((const void*)k_km_count, hipFuncAttributeMaxDynamicSharedMemorySize, 5 * KM_MAX_K_SORT * 4));
            HIP_TRY(hipFuncSetAttribute((const void*)k_km_place, hipFuncAttributeMaxDynamicSharedMemorySize, 5 * KM_MAX_K_SORT * 4));
            sober_attr_done(attr_set);
        }
    }
    for (int it = 0; it < iters; ++it) {
        if (mfma_e) {
            const int64_t n_waves = (N + 16 * KM_PB - 1) / (16 * KM_PB);
            const dim3 grid((unsigned)((n_waves + 3) / 4));
#define KM_CASE(T) case T: hipLaunchKernelGGL((k_kmeans_assign_mfma<T>), grid, dim3(256), fuse_count ? lds_count : 0, st, X, N, d, \
                                              cent, Caug, K, Kp, labels, fuse_count ? ucount : (int32_t*)nullptr, n_units); break;
            switch (kt) { KM_CASE(1) KM_CASE(2) KM_CASE(3) KM_CASE(4) KM_CASE(5) KM_CASE(6) KM_CASE(7) KM_CASE(8) default: break; }
#undef KM_CASE
        } else {
            hipLaunchKernelGGL((k_kmeans_assign<DT>), dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, X,
                               N, d, cent, K, labels);
        }
        LAUNCH_CHECK();
        if (sorted) {
            const dim3 ugrid((unsigned)((n_units + 3) / 4));
            if (!fuse_count) {                               // (else the matrix-core E step has counted already)
                hipLaunchKernelGGL(k_km_count, ugrid, dim3(256), lds_count, st, labels, N, K, n_units, ucount);
                LAUNCH_CHECK();
            }
            hipLaunchKernelGGL(k_km_scan, dim3(K), dim3(256), 0, st, ucount, n_units, tot);
            LAUNCH_CHECK();
            hipLaunchKernelGGL(k_km_place, ugrid, dim3(256), lds_place, st, labels, N, K, n_units, ucount, tot, order);
            LAUNCH_CHECK();
            hipLaunchKernelGGL((k_kmeans_update_sorted<DT>), dim3(K), dim3(256), 0, st, X, d, K, tot, order, cent, Caug, 4 * kt);
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
