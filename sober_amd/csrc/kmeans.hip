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
// E step, SCREENED (round 4; pools of >= KM_SCREEN_MIN_N points, d <= 31, the centroid image fits LDS): the FP64 form is
//         0.6 of the FP64 matrix peak at 1M x 20 x 500 (0.50 ms of the 0.69 ms an iteration takes) -- the arithmetic
//         itself is the bound.  But the label only needs the ORDER of the distances, and an exact evaluation only where two
//         of them are close: a first pass evaluates |c'|^2 - 2 x'.c' (x' = x - mu, c' = c - mu: centred on the initial
//         centroids' mean, which leaves every distance as it is and takes the common offset out of the magnitudes) on the
//         BF16 matrix cores from two-piece splits x' = xh + xl, -2c' = ah + al -- products xh ah + xh al + xl ah side by side
//         along K, |c'|^2 as three pieces against 1.0, FP32 accumulation: v_mfma_f32_16x16x32_bf16, 3 d + 3 <= 32 NK
//         slots -- keeps the two smallest per point (a value carries its index in its 7 low mantissa bits, 8 beyond
//         K = 512), and takes the smallest as the label when the gap exceeds 2^-12 (|x'|^2 + max |c'|^2): the split
//         (<= 2^-16.4 of that scale), the dropped xl al products (2^-18), 64 FP32 accumulations (<= 2^-15) and the index bits
//         (<= 2^-16; 2^-15 with 8) move a value by at most 2^-14, both together by half the margin (0.62 of it with 8 bits);
//         measured: no label differs down to a margin of 2^-16 in 78 M point-iterations (profiles/r04_kmeans_margin.txt).  Every other point --
//         near-ties, non-finite data or centroids, scales outside 1e-20 .. 1e30 -- goes to a list, and the FP64 kernel
//         above runs on the list (its own exact re-check included): labels bit-equal to k_kmeans_assign's by construction.
//         The points' BF16 image is built once per call (X does not change over the iterations), the centroids' by the M
//         step; the first pass counts its labels for the sort, the list pass corrects the counts it changes.
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
// caller wants (it leaves the two smallest equal and sends the point to the re-check).
// `after`: the compare mask of the same MFMA result.  The hazard recogniser does not look inside inline asm: it pads the
// compiler's own first read of an MFMA result (that compare) with the wait states the result needs, but an asm statement
// that reads the result was free to be scheduled right behind the MFMA (round 3's listing had the v_max_f64 there, in
// front of the s_nops: a stale operand for the SECOND smallest value -- the margin test's input).  Taking the mask as an
// operand orders the asm behind the compare.
__device__ __forceinline__ double km_min(double a, double b, unsigned long long after) {
    double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b), "s"(after)); return r;
}
__device__ __forceinline__ double km_max(double a, double b, unsigned long long after) {
    double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b), "s"(after)); return r;
}

__device__ __forceinline__ void km_round_groups(int label, int lane, int& rank, int& cnt, bool& leader);

// ucount != NULL: the workgroup also COUNTS its 256 points per label (the counting sort's first pass, see below: unit =
// workgroup; every wave counts its 64 points into its own K counters in dynamic LDS -- 4 K ints --, the four are added
// at the end): the labels never travel to a counting kernel and back
// LIST: the second pass of the screened E step -- the points are flist[0 .. *n_list), the waves walk the list with a grid
// stride, a label that differs from the first pass's moves one count of the sort (ucount: [K][n_units], unit = 256 points)
// (the list is short -- a few points per thousand -- and a wave is alone with its share: 16 points per wave there, 64 in
//  the full pass)
template <int KT, bool LIST>       // DA = 4 KT >= d + 1
__global__ __launch_bounds__(256) void k_kmeans_assign_mfma(const double* __restrict__ X, int64_t N, int d,
                                                            const double* __restrict__ cent,
                                                            const double* __restrict__ Caug, int K, int Kp,
                                                            int32_t* __restrict__ labels,
                                                            int32_t* __restrict__ ucount, int64_t n_units,
                                                            const int32_t* __restrict__ flist,
                                                            const unsigned* __restrict__ n_list,
                                                            const int* __restrict__ nan_first_p) {
    extern __shared__ int km_hist[];
    typedef double d4 __attribute__((ext_vector_type(4)));
    constexpr int DA = 4 * KT, PB = LIST ? 1 : KM_PB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lj = lane & 15, lg = lane >> 4;
    const int64_t NP = LIST ? (int64_t)*n_list : N;           // points to label
    if constexpr (LIST) {
        // A NaN centroid (an empty cluster's 0 / 0) sends EVERY point to this list, and every finite point's answer is known
        // without a single tile: its first NaN distance is that centroid (the reference's argmin returns the first NaN).
        // Walking the 32 centroid tiles for a million points sixteen at a time took this launch from 15 us to 10 ms per
        // iteration (the advisor's "silent cliff", tests/test_hip_round6.py times it); now one thread per listed point:
        // finite -> that centroid, a NaN coordinate -> 0 (all its distances are NaN), an infinite one -> the exact loop.
        const int nf = nan_first_p != nullptr ? *nan_first_p : 0x7fffffff;
        if (nf != 0x7fffffff) {                               // (uniform)
            for (int64_t ip = (int64_t)blockIdx.x * 256 + threadIdx.x; ip < NP; ip += (int64_t)gridDim.x * 256) {
                const int64_t i = flist[ip];
                bool x_nan = false, x_inf = false;
                for (int j = 0; j < d; ++j) { const double v = X[i * d + j]; x_nan |= v != v; x_inf |= fabs(v) > 1e150; }
                int lab = x_nan ? 0 : nf;
                if (x_inf && !x_nan) {
                    double best = __builtin_inf();
                    int bi = 0;
                    bool best_nan = false;
                    for (int k = 0; k < K; ++k) {
                        double dist = 0.0;
                        for (int j = 0; j < d; ++j) { const double df = X[i * d + j] - cent[(size_t)k * d + j]; dist = fma(df, df, dist); }
                        const bool isn = dist != dist;
                        if (!best_nan && (isn || dist < best)) { best = dist; bi = k; best_nan = isn; }
                    }
                    lab = bi;
                }
                const int was = labels[i];
                if (was != lab) {
                    atomicAdd(ucount + (size_t)was * n_units + i / 256, -1);
                    atomicAdd(ucount + (size_t)lab * n_units + i / 256, 1);
                    labels[i] = lab;
                }
            }
            return;
        }
    }
  for (int64_t wu = (int64_t)blockIdx.x * 4 + wave; !LIST || wu * (16 * PB) < NP; wu += (int64_t)gridDim.x * 4) {
    const int64_t p0 = wu * (16 * PB);
    if (p0 >= NP && (LIST || ucount == nullptr)) return;      // (a counting workgroup keeps all its waves for the final sum)
    int* const hist = km_hist + wave * K;
    if (!LIST && ucount != nullptr)
        for (int k = lane; k < K; k += 64) hist[k] = 0;     // (my own counters: no barrier needed before I use them)
    // B fragments: point p0 + 16 bl + lj, contraction slices kk = lg KT + ks (the same permutation on both operands)
    double b[PB][KT], x2[PB];
#pragma unroll
    for (int bl = 0; bl < PB; ++bl) {
        const int64_t ip = min(p0 + 16 * bl + lj, NP - 1);
        const int64_t i = LIST ? (int64_t)flist[ip] : ip;
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
    // A fragments G tiles ahead (G = 1 in the full pass: four point blocks per tile and several waves per SIMD cover an L2
    // round trip; the list pass has one block per tile and the wave is alone on its SIMD -- one tile ahead left ~0.4 us of
    // every tile exposed, 15-19 us per launch: four tiles per request there)
    constexpr int G = LIST ? 4 : 1;
    double a[G][KT], an[G][KT];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int ks = 0; ks < KT; ++ks) a[g][ks] = Caug[(size_t)(16 * min(g, n_tiles - 1) + lj) * DA + lg * KT + ks];
    for (int t0 = 0; t0 < n_tiles; t0 += G) {
        // the next tiles' A fragments are requested FIRST (pinned: the compiler otherwise sinks the loads to the end of
        // the body and waits for them there -- an exposed L2 round trip per tile); they are taken over after the body
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int tn = min(t0 + G + g, n_tiles - 1);
#pragma unroll
            for (int ks = 0; ks < KT; ++ks) an[g][ks] = Caug[(size_t)(16 * tn + lj) * DA + lg * KT + ks];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int t = t0 + g;
            if (t >= n_tiles) break;                         // (uniform)
#pragma unroll
            for (int ks = 0; ks < KT; ++ks) {
                if (ks == ksn && 16 * t + lj < K) {
                    const double n2 = a[g][ks];
                    first_nan = (n2 != n2) ? min(first_nan, 16 * t + lj) : first_nan;
                    any_inf |= n2 > 1e300;
                    cmax2 = (n2 <= 1e300) ? fmax(cmax2, n2) : cmax2;
                }
            }
#pragma unroll
            for (int bl = 0; bl < PB; ++bl) {
                d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < KT; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[g][ks], b[bl][ks], acc, 0, 0, 0);
                // lane (lj, lg): centroid rows lg + 4 r of the tile against point lj: the two smallest so far
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double v = acc[r];
                    const bool lt = v < b1[bl];
                    const unsigned long long ltm = __builtin_amdgcn_ballot_w64(lt);
                    b2[bl] = km_min(b2[bl], km_max(b1[bl], v, ltm), ltm);
                    b1[bl] = km_min(b1[bl], v, ltm);
                    i1[bl] = lt ? (4 * t + r) : i1[bl];
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int ks = 0; ks < KT; ++ks) a[g][ks] = an[g][ks];
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
        const int64_t ip = p0 + 16 * bl + lj;
        if (lg == 0 && ip < NP) {
            const int64_t i = LIST ? (int64_t)flist[ip] : ip;
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
            if constexpr (LIST) {
                const int was = labels[i];                   // (the first pass's candidate: counted under that label)
                if (was != lab[bl]) {
                    atomicAdd(ucount + (size_t)was * n_units + i / 256, -1);
                    atomicAdd(ucount + (size_t)lab[bl] * n_units + i / 256, 1);
                    labels[i] = lab[bl];
                }
            } else {
                labels[i] = lab[bl];
            }
        }
    }
    if constexpr (LIST) continue;
    if (ucount == nullptr) return;
    // point p0 + L's label into lane L (it sits in lane L & 15 of block L >> 4), then one LDS add per point
    int l64 = -1;
#pragma unroll
    for (int bl = 0; bl < PB; ++bl) {
        const int v = __shfl(lab[bl], lane & 15, 64);
        l64 = ((lane >> 4) == bl) ? v : l64;
    }
    if (l64 >= 0) atomicAdd(hist + l64, 1);                   // (counts only: integer adds in any order)
    __syncthreads();
    if (blockIdx.x < n_units)
        for (int k = threadIdx.x; k < K; k += 256)
            ucount[(size_t)k * n_units + blockIdx.x] = (km_hist[k] + km_hist[K + k]) + (km_hist[2 * K + k] + km_hist[3 * K + k]);
    return;
  }
}

// ---- the screened E step's first pass (see the header) -----------------------------------------------------------------
// K slots: s = 3 j + {0, 1, 2} for coordinate j < d: points {xh, xh, xl}, centroids {ah, al, ah} (a = -2 (c - mu));
// s = 3 d + {0, 1, 2}: points 1.0, centroids the three pieces of |c - mu|^2; the rest zero.  KS = 32 NK slots per row.
struct KmStat {
    unsigned long long cmax2_bits[2];   // max |c - mu|^2 over the finite centroids (the bits of a non-negative double order
    unsigned bad[2];                    // like integers); bad: a centroid that is not finite.  [parity of the iteration]
    unsigned n_list;                    // points the first pass sent to the list
    unsigned listed;                    // ... summed over the iterations of the call (sober_kmeans_stat_offset)
    int nan_first[2];                   // the first centroid with a NaN coordinate (0x7fffffff: none).  [parity of the iteration]
    unsigned big_ticket;                // the M step of a cluster that holds more than half the pool: arrivals of its KM_BIG_SPLIT
};                                      // workgroups (k_kmeans_update_sorted; their partial sums borrow the E step's list)
static_assert(sizeof(KmStat) <= 256, "the workspace keeps 256 bytes for it (km_off_mu)");
constexpr int KM_BIG_SPLIT = 64;
constexpr int KM_SCREEN_MIN_N = 4096;          // (eligible from here on; RECOMMENDED from KM_SCREEN_MIN_WORK on)
// pool size x FP64 contraction steps from which the screened E step is faster than the FP64 one (same box, K = 500:
// 20k x 10 0.54 vs 0.49 ms, 50k x 6 0.53 vs 0.46, 50k x 20 0.65 vs 0.65, 100k x 10 0.70 vs 0.73, 200k x 10 0.86 vs 1.12,
// 1M x 20 2.8 vs 5.8): below it sober_kmeans_ws_bytes asks for the FP64 form's workspace only
constexpr long long KM_SCREEN_MIN_WORK = 300000;
// the margin's exponent: 2^-12 in every build that ships.  A diagnostic build with a smaller margin measures how far the
// BF16 values really are from the exact ones: the smallest margin that still leaves every label right
// (scripts/kmeans_margin.sh -> profiles/r04_kmeans_margin.txt)
#ifndef KM_MARGIN_LOG2
#define KM_MARGIN_LOG2 12
#elif !defined(SOBER_DIAG_BUILD)
#error "KM_MARGIN_LOG2 is a diagnostic switch: build with -DSOBER_DIAG_BUILD"
#endif
constexpr int KM_SCREEN_T = 512;        // threads of the first pass: 8 waves x 64 points = two units of the counting sort
constexpr int KM_SCREEN_LDS = 78 * 1024;    // two workgroups per compute unit

__device__ __forceinline__ unsigned short km_bf16(double v) {           // round to nearest even (NaN stays NaN, Inf stays Inf)
    unsigned u = __float_as_uint((float)v);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ double km_bf16_val(unsigned short h) { return (double)__uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ void km_split2(double v, unsigned short& h, unsigned short& l) {
    h = km_bf16(v);
    l = km_bf16(v - km_bf16_val(h));
}

// mu = mean of the first K rows (the initial centroids): any vector would do -- it only has to sit inside the cloud
__global__ __launch_bounds__(1024) void k_km_mu(const double* __restrict__ X, int K, int d, double* __restrict__ mu,
                                                KmStat* __restrict__ st) {
    __shared__ double s_part[32][33];
    const int tid = threadIdx.x, j = tid & 31, g = tid >> 5;     // 32 row groups x 32 coordinates (a chain of K / 32 loads each)
    if (tid == 0) { st->cmax2_bits[0] = 0ull; st->cmax2_bits[1] = 0ull; st->bad[0] = 0u; st->bad[1] = 0u; st->n_list = 0u; st->listed = 0u;
                    st->nan_first[0] = 0x7fffffff; st->nan_first[1] = 0x7fffffff; st->big_ticket = 0u; }
    double acc = 0.0;
    if (j < d)
        for (int k = g; k < K; k += 32) acc += X[(size_t)k * d + j];
    s_part[g][j] = acc;
    __syncthreads();
    if (tid >= d) return;
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < 32; ++q) t += s_part[q][tid];
    const double m = t / (double)K;
    mu[tid] = (m == m && fabs(m) < 1e300) ? m : 0.0;
}

// the centroid's row of the BF16 image, by the threads of one workgroup (>= KS of them); c: the centroid (LDS or global)
__device__ __forceinline__ void km_centroid_row(const double* c, const double* __restrict__ mu, int d, int KS,
                                                unsigned short* __restrict__ row, KmStat* __restrict__ st, int par, int tid,
                                                int k) {
    if (tid < d) {
        unsigned short h, l;
        km_split2(-2.0 * (c[tid] - mu[tid]), h, l);
        row[3 * tid] = h; row[3 * tid + 1] = l; row[3 * tid + 2] = h;
    } else if (tid == d) {
        double n2 = 0.0;
        for (int j = 0; j < d; ++j) { const double cp = c[j] - mu[j]; n2 = fma(cp, cp, n2); }
        const unsigned short p0 = km_bf16(n2);
        const double r1 = n2 - km_bf16_val(p0);
        const unsigned short p1 = km_bf16(r1);
        const unsigned short p2 = km_bf16(r1 - km_bf16_val(p1));
        row[3 * d] = p0; row[3 * d + 1] = p1; row[3 * d + 2] = p2;
        if (n2 < 1e300) atomicMax(&st->cmax2_bits[par], (unsigned long long)__double_as_longlong(n2));
        else atomicOr(&st->bad[par], 1u);                    // (NaN or Inf: an empty cluster's 0 / 0, overflowing data)
        if (n2 != n2) atomicMin(&st->nan_first[par], k);     // (a NaN centroid: every finite point's first NaN distance)
    } else if (tid >= 3 * d + 3 && tid < KS) {
        row[tid] = 0;
    }
}

// the image of the initial centroids (rows < K) and of the padding rows up to Kp (|c|^2 = 1e30: they never win)
__global__ __launch_bounds__(128) void k_km_cprep(const double* __restrict__ cent, int K, int d, int KS,
                                                  const double* __restrict__ mu, unsigned short* __restrict__ Cb,
                                                  KmStat* __restrict__ st) {
    const int k = blockIdx.x, tid = threadIdx.x;
    unsigned short* row = Cb + (size_t)k * KS;
    if (k < K) { km_centroid_row(cent + (size_t)k * d, mu, d, KS, row, st, 0, tid, k); return; }
    if (tid < KS) row[tid] = (tid == 3 * d) ? km_bf16(1e30) : (unsigned short)0;
}

// the points' image, once per call: Xb[i][KS] and |x - mu|^2 (FP32: it only scales the margin).  One thread per 16-byte
// piece (8 slots: the coordinates floor(8 q / 3) .. + 3), 4 NK threads per point: rows are read and written as whole lines
template <int NK>
__global__ __launch_bounds__(256) void k_km_xprep(const double* __restrict__ X, int64_t N, int d,
                                                  const double* __restrict__ mu, uint4* __restrict__ Xb,
                                                  float* __restrict__ xn2) {
    constexpr int LP = 4 * NK, PPW = 256 / LP;
    __shared__ double s_x2[PPW][LP];
    const int p = threadIdx.x / LP, q = threadIdx.x - p * LP;
    const int64_t i = (int64_t)blockIdx.x * PPW + p;
    const bool active = p < PPW && i < N;
    const int s0 = 8 * q, j0 = s0 / 3;
    unsigned short h[4], l[4];
    double x2 = 0.0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int j = j0 + t;
        h[t] = 0; l[t] = 0;
        if (active && j < d) {
            const double xv = X[i * d + j] - mu[j];
            km_split2(xv, h[t], l[t]);
            if (3 * j >= s0 && 3 * j < s0 + 8) x2 = fma(xv, xv, x2);      // (the piece that holds the coordinate's first slot)
        } else if (j == d) {
            h[t] = 0x3f80; l[t] = 0x3f80;                    // 1.0 against the three pieces of |c'|^2
        }
    }
    unsigned w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int sl = s0 + e, jj = sl / 3 - j0, pz = sl - 3 * (sl / 3);
        const unsigned short hh = jj == 0 ? h[0] : (jj == 1 ? h[1] : (jj == 2 ? h[2] : h[3]));
        const unsigned short ll = jj == 0 ? l[0] : (jj == 1 ? l[1] : (jj == 2 ? l[2] : l[3]));
        w[e >> 1] |= (unsigned)(pz == 2 ? ll : hh) << (16 * (e & 1));
    }
    if (active) Xb[i * LP + q] = make_uint4(w[0], w[1], w[2], w[3]);
    if (p < PPW) s_x2[p][q] = x2;
    __syncthreads();
    if (active && q == 0) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < LP; ++k) t += s_x2[p][k];
        xn2[i] = (float)t;
    }
}

typedef __bf16 km_bf16x8 __attribute__((ext_vector_type(8)));
typedef float km_f4 __attribute__((ext_vector_type(4)));

// first pass: labels (a candidate where the point goes to the list), the list, the sort's counts per unit of 256 points.
// Dynamic LDS: the centroid image, Kp rows of KS * 2 + 16 bytes (the pad spreads the 16-byte fragment reads over the
// banks), then 2 K counters.  par: parity of the iteration (which KmStat slot the M step filled; the other one is cleared).
// PB: 16-point blocks per wave -- 4: a trip of the workgroup covers 512 points (two units of the sort); 2: 256 points (one
// unit), twice the waves per point for pools too small to fill the chip with the first form
// IB: bits of a value's place inside the lane's stream (4 n_tiles <= 2^IB: 7 up to Kp = 512, 8 beyond)
template <int NK, int PB, int IB>
__global__ __launch_bounds__(KM_SCREEN_T) void k_kmeans_screen(const uint4* __restrict__ Xb, const float* __restrict__ xn2,
                                                                 int64_t N, const uint4* __restrict__ Cb, int K, int Kp,
                                                                 KmStat* __restrict__ st, int par,
                                                                 int32_t* __restrict__ labels, int32_t* __restrict__ ucount,
                                                                 int64_t n_units, int32_t* __restrict__ flist) {
    extern __shared__ int km_hist[];
    constexpr int KS = 32 * NK, ROWB = KS * 2 + 16, PTS = 128 * PB, NU = PB / 2;     // points, units per trip
    char* const cimg = (char*)km_hist;
    int* const hist = (int*)(cimg + (size_t)Kp * ROWB);      // [NU][K]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lj = lane & 15, lg = lane >> 4;
    for (int q = tid; q < Kp * NK * 4; q += KM_SCREEN_T) {
        const int row = q / (NK * 4), part = q % (NK * 4);
        *(uint4*)(cimg + (size_t)row * ROWB + part * 16) = Cb[q];
    }
    const float cmax2 = (float)__longlong_as_double((long long)st->cmax2_bits[par]);
    const bool bad = st->bad[par] != 0u;
    if (blockIdx.x == 0 && tid == 0) { st->cmax2_bits[par ^ 1] = 0ull; st->bad[par ^ 1] = 0u; st->nan_first[par ^ 1] = 0x7fffffff; }   // (the M step fills it next)
    const int n_tiles = Kp >> 4;
    const int64_t n_pairs = (N + PTS - 1) / PTS;
    for (int64_t pair = blockIdx.x; pair < n_pairs; pair += gridDim.x) {
        __syncthreads();                                      // (the image is in place; the last trip's counters are out)
        for (int k = tid; k < NU * K; k += KM_SCREEN_T) hist[k] = 0;
        __syncthreads();
        const int64_t p0 = pair * PTS + wave * (16 * PB);
        km_bf16x8 b[PB][NK];
        float x2[PB], bias[PB];
        unsigned b1[PB], b2[PB];
        unsigned keep = ~((1u << IB) - 1u);
        asm volatile("" : "+v"(keep));                        // (a VGPR: v_and_or takes one scalar operand, the place)
#pragma unroll
        for (int bl = 0; bl < PB; ++bl) {
            const int64_t i = min(p0 + 16 * bl + lj, N - 1);
#pragma unroll
            for (int c = 0; c < NK; ++c) b[bl][c] = __builtin_bit_cast(km_bf16x8, Xb[(i * NK + c) * 4 + lg]);
            x2[bl] = xn2[i];
            b1[bl] = 0x7f800000u; b2[bl] = 0x7f800000u;        // +inf
            bias[bl] = x2[bl] + 0x1p-10f * (x2[bl] + cmax2);  // |x' - c'|^2 + bias: positive whatever the rounding
        }
        km_bf16x8 a[NK], an[NK];
#pragma unroll
        for (int c = 0; c < NK; ++c) a[c] = __builtin_bit_cast(km_bf16x8, *(const uint4*)(cimg + (size_t)lj * ROWB + c * 64 + lg * 16));
        for (int t = 0; t < n_tiles; ++t) {
            const int tn = min(t + 1, n_tiles - 1);
#pragma unroll
            for (int c = 0; c < NK; ++c)
                an[c] = __builtin_bit_cast(km_bf16x8, *(const uint4*)(cimg + (size_t)(16 * tn + lj) * ROWB + c * 64 + lg * 16));
            __builtin_amdgcn_sched_barrier(0);                // (the next tile's fragments are requested first, used a trip later)
#pragma unroll
            for (int bl = 0; bl < PB; ++bl) {
                km_f4 acc = (km_f4){bias[bl], bias[bl], bias[bl], bias[bl]};
#pragma unroll
                for (int c = 0; c < NK; ++c) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[c], b[bl][c], acc, 0, 0, 0);
                // lane (lj, lg): centroid rows 4 lg + r of the tile against point lj: the two smallest so far
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // Three INTEGER instructions per value.  The accumulator starts at |x'|^2 plus a small bias, so every
                    // value is a positive float (a squared distance + bias - error > 0) and unsigned integer order IS float
                    // order; the value carries its place (4 t + r) in its low IB mantissa bits (<= 2^-16 of the scale at 7
                    // bits, 2^-15 at 8: inside the margin's budget, see the header): v_and_or, the new second smallest = the
                    // middle of (b1, b2, v), the new smallest = min(b1, v).  (A float min puts a canonicalising v_max in
                    // front of an MFMA result; an inline-asm v_min is not seen by the hazard recogniser -- it read the
                    // MFMA's result early; a separate index takes a compare and two selects.)
                    unsigned place = (unsigned)(4 * t + r);
                    asm("" : "+s"(place));                    // (its own SGPR: v_and_or instead of v_and + v_or3 with r inline)
                    const unsigned v = (__float_as_uint(acc[r]) & keep) | place;
                    // (v_med3_u32 has no builtin; its operands are VALU results the compiler made -- the v_and_or above --,
                    //  not the MFMA's registers: no hazard hides in this asm)
                    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(b2[bl]) : "v"(b1[bl]), "v"(b2[bl]), "v"(v));   // (b1 <= b2: the middle one)
                    b1[bl] = min(b1[bl], v);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < NK; ++c) a[c] = an[c];
        }
        int lab[PB];
#pragma unroll
        for (int bl = 0; bl < PB; ++bl) {
            lab[bl] = -1;
            float m1 = __uint_as_float(b1[bl]), m2 = __uint_as_float(b2[bl]);
            const int code = (int)(b1[bl] & ((1u << IB) - 1u));                  // (+inf, nothing seen: 0)
            int k1 = 16 * (code >> 2) + 4 * lg + (code & 3);
#pragma unroll
            for (int o = 16; o <= 32; o <<= 1) {
                const float o1 = __shfl_xor(m1, o, 64), o2 = __shfl_xor(m2, o, 64);
                const int ok = __shfl_xor(k1, o, 64);
                const bool take = o1 < m1;
                m2 = fminf(fminf(m2, o2), fmaxf(m1, o1));
                k1 = take ? ok : k1;
                m1 = fminf(m1, o1);
            }
            const int64_t i = p0 + 16 * bl + lj;
            const bool valid = lg == 0 && i < N;
            const float scale = x2[bl] + cmax2;
            const bool sure = !bad && scale > 1e-20f && scale < 1e30f && (m2 - m1) > __builtin_ldexpf(1.0f, -KM_MARGIN_LOG2) * scale && k1 < K;
            if (valid) { lab[bl] = min(k1, K - 1); labels[i] = lab[bl]; }
            const unsigned long long fl = __ballot(valid && !sure);
            if (fl != 0ull) {                                 // (uniform)
                unsigned base = 0;
                if (lane == 0) base = atomicAdd(&st->n_list, (unsigned)__popcll(fl));
                base = __builtin_amdgcn_readfirstlane(base);
                if (valid && !sure) flist[base + __popcll(fl & ((1ull << lane) - 1ull))] = (int32_t)i;
            }
        }
        // the counting sort's first pass: point p0 + L's label into lane L (the unit's counters are integer adds: any order)
        int l64 = -1;
#pragma unroll
        for (int bl = 0; bl < PB; ++bl) {
            const int v = __shfl(lab[bl], lane & 15, 64);
            l64 = ((lane >> 4) == bl) ? v : l64;
        }
        if (l64 >= 0) atomicAdd(hist + (wave * PB / 16) * K + l64, 1);     // (counts only: one LDS add per point, no ranks needed)
        __syncthreads();
        for (int q = tid; q < NU * K; q += KM_SCREEN_T) {
            const int u = q / K, k = q - u * K;
            const int64_t unit = pair * NU + u;
            if (unit < n_units) ucount[(size_t)k * n_units + unit] = hist[q];
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

// ---- stable counting sort of the points by label, no atomics (see the header) --------------------------------------
// A wave owns 256 consecutive points (unit u = 4 * workgroup + wave), four rounds of 64.  hist = the wave's own K
// counters in LDS.  The lanes that share a label are found with ballots; within a round they are ranked in lane order,
// across rounds by the running counter: the order inside a cluster is ascending point index.
constexpr int KM_UNIT = 256;
// one round of 64 points: rank of every lane among the lanes of its label (lane order), the group's size, its leader --
// ballots only, nothing touches memory inside the loop (one trip per distinct label of the round)
__device__ __forceinline__ void km_round_groups(int label, int lane, int& rank, int& cnt, bool& leader, unsigned long long todo) {
    const unsigned long long below = (1ull << lane) - 1ull;
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
__device__ __forceinline__ void km_round_groups(int label, int lane, int& rank, int& cnt, bool& leader) {
    km_round_groups(label, lane, rank, cnt, leader, __ballot(label >= 0));
}
// hist[k]: the wave's running position for label k (counting: starts at 0; placing: starts at the cluster's offset + the
// wave's offset inside the cluster).  Counting is one LDS add per point (integer sums: any order).  Placing needs the
// rank of a point among the round's points of its label: with K in the hundreds most of a round's 64 labels occur once
// (rank 0) -- a scratch counter per label (tmp, zero between rounds) tells which, and only the labels that occur twice or
// more go through the ballot loop (~4 trips per round at K = 500 instead of ~60: 46 -> 2x us at 1M points).
template <bool PLACE>
__device__ __forceinline__ void km_wave_pass(const int32_t* __restrict__ labels, int64_t N, int64_t unit, int* hist, int* tmp,
                                             int32_t* __restrict__ order) {
    volatile int* hv = hist;                                 // (the leaders write what the whole wave reads next round)
    volatile int* tv = tmp;
    const int lane = threadIdx.x & 63;
    for (int r = 0; r < KM_UNIT / 64; ++r) {
        const int64_t i = unit * KM_UNIT + r * 64 + lane;
        const int label = (i < N) ? labels[i] : -1;
        if (!PLACE) {
            if (label >= 0) atomicAdd(hist + label, 1);
            continue;
        }
        // (a wave's LDS operations are executed in order: the adds, then the reads, then the clears)
        if (label >= 0) atomicAdd(tmp + label, 1);
        const int occ = (label >= 0) ? tv[label] : 0;
        if (label >= 0) tv[label] = 0;
        int rank = 0, cnt = 1;
        bool leader = label >= 0;
        const unsigned long long dup = __ballot(occ > 1);
        if (dup != 0ull) {                                   // (uniform)
            int r2, c2;
            bool l2;
            km_round_groups(occ > 1 ? label : -1, lane, r2, c2, l2, dup);
            if (occ > 1) { rank = r2; cnt = c2; leader = l2; }
        }
        const int before = (label >= 0) ? hv[label] : 0;
        if (label >= 0) order[before + rank] = (int32_t)i;
        if (leader) hv[label] = before + cnt;                // (after every read of the round)
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
    km_wave_pass<false>(labels, N, unit, hist, nullptr, nullptr);
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
    extern __shared__ int km_hist[];                         // [4][K] counters, [K] cluster offsets, [4][K] scratch
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
    int* tmp = km_hist + (5 + wave) * K;
    for (int k = lane; k < K; k += 64) { hist[k] = coff[k] + uoff[(size_t)k * n_units + unit]; tmp[k] = 0; }
    km_wave_pass<true>(labels, N, unit, hist, tmp, order);
}

// M step on cluster-sorted points: workgroup k sums rows X[order[lo .. lo + n_k)] (ascending point indices)
// (512 threads: twice the rows in flight per cluster -- a thread's share of a 2000-point cluster is a chain of ~20 row
//  fetches instead of ~40: 88 -> 67 us at 1M x 20; 1024 threads: 72)
constexpr int KM_UPD_T = 512;
template <int DT>
__global__ __launch_bounds__(KM_UPD_T) void k_kmeans_update_sorted(const double* __restrict__ X, int d, int K,
                                                              const int32_t* __restrict__ tot,
                                                              const int32_t* __restrict__ order,
                                                              double* __restrict__ cent, double* __restrict__ Caug, int DA,
                                                              const double* __restrict__ mu, unsigned short* __restrict__ Cb,
                                                              int KS, KmStat* __restrict__ st, int par_next, int64_t N_pool,
                                                              double* __restrict__ big_part) {
    __shared__ double s_c[DT];
    const int tid = threadIdx.x;
    int k = blockIdx.x, split = -1;
    // A cluster that holds more than half the pool (the degenerate clustering behind an empty cluster: the reference's argmin
    // sends EVERY point to the first NaN centroid) is not summed by one workgroup -- 9.6 ms per iteration at 1M x 20, the
    // second half of the advisor's "cliff" -- but by the KM_BIG_SPLIT workgroups launched behind the K regular ones: each sums
    // a contiguous share in this kernel's own order, the last to arrive adds the shares in share order (fixed: run-to-run
    // bit-equal).  Every other cluster is summed exactly as before; without the screen's statistics block (N_pool = 0) so is this one.
    if (k >= K) {                                            // one of the extra workgroups: is there such a cluster?
        __shared__ int s_g;
        if (tid == 0) s_g = -1;
        __syncthreads();
        for (int j = tid; j < K; j += KM_UPD_T)
            if (2 * (int64_t)tot[j] > N_pool) s_g = j;       // (at most one)
        __syncthreads();
        if (s_g < 0) return;
        split = k - K;
        k = s_g;
    } else if (N_pool > 0 && 2 * (int64_t)tot[k] > N_pool) {
        return;                                              // (the extra workgroups' cluster)
    }
    // lo = sum of the sizes of the clusters in front of mine
    __shared__ int s_lo[KM_UPD_T / 64];
    int part = 0;
    for (int j = tid; j < k; j += KM_UPD_T) part += tot[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
    if ((tid & 63) == 0) s_lo[tid >> 6] = part;
    __syncthreads();
    int lo = 0;
#pragma unroll
    for (int q = 0; q < KM_UPD_T / 64; ++q) lo += s_lo[q];
    const int n_k = tot[k];
    const int share = (n_k + KM_BIG_SPLIT - 1) / KM_BIG_SPLIT;
    const int p_lo = split < 0 ? 0 : min(n_k, split * share), p_hi = split < 0 ? n_k : min(n_k, p_lo + share);
    // a row is read by DT / 4 neighbouring lanes, 32 bytes each (one thread per row walked it with 8-byte loads, 64 rows
    // -- 128 cache lines -- per load instruction, every line touched again by the next 19: 119 us at 1M x 20, x 10
    // iterations); lane (rs, c) sums chunk c of rows rs, rs + RP, ...; the RP partial sums of a coordinate are added in
    // order at the end -- a fixed order, like the one before
    constexpr int CH = DT / 4, RP = KM_UPD_T / CH;
    __shared__ double s_acc[RP][DT];
    const int ch = tid % CH, rs = tid / CH;
    double a4[4] = {0.0, 0.0, 0.0, 0.0};
    if (rs < RP) {
#pragma unroll 4
        for (int p = p_lo + rs; p < p_hi; p += RP) {
            const double* row = X + (size_t)order[lo + p] * d + 4 * ch;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * ch + e < d) a4[e] += row[e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) s_acc[rs][4 * ch + e] = a4[e];
    }
    __syncthreads();
    double sum = 0.0;
    if (tid < d) {
        const int j = tid;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;       // four interleaved chains, then one fixed combination
        int r = 0;
        for (; r + 3 < RP; r += 4) { s0 += s_acc[r][j]; s1 += s_acc[r + 1][j]; s2 += s_acc[r + 2][j]; s3 += s_acc[r + 3][j]; }
        for (; r < RP; ++r) s0 += s_acc[r][j];
        sum = (s0 + s1) + (s2 + s3);
    }
    if (split >= 0) {                                        // my share is out; the last share to arrive adds them all
        if (tid < d) big_part[split * 32 + tid] = sum;
        __threadfence();
        __syncthreads();
        __shared__ int s_last;
        if (tid == 0) {
            const unsigned t = __hip_atomic_fetch_add(&st->big_ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            s_last = (t == (unsigned)KM_BIG_SPLIT - 1u) ? 1 : 0;
            if (s_last) __hip_atomic_store(&st->big_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (!s_last) return;                                 // (uniform)
        __threadfence();
        if (tid < d) {
            sum = 0.0;
            for (int e = 0; e < KM_BIG_SPLIT; ++e)
                sum += __hip_atomic_load(big_part + e * 32 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (tid < d) {
        const int j = tid;
        const double c = sum / (double)n_k;                  // (an empty cluster: 0 / 0 = NaN, as in the reference)
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
    if (Cb == nullptr) return;
    // (screened E step) the BF16 image of this centroid for the next first pass; the list is consumed by now
    km_centroid_row(s_c, mu, d, KS, Cb + (size_t)k * KS, st, par_next, tid, k);
    if (k == 0 && tid == KM_UPD_T - 1) { st->listed += st->n_list; st->n_list = 0u; }
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
// screened E step: + [KmStat | mu 32 doubles | Cb Kp x KS bf16 | |x'|^2 N floats | list N int32 | Xb N x KS bf16]
static inline int km_ks(int d) { return 3 * d + 3 <= 32 ? 32 : (3 * d + 3 <= 64 ? 64 : 96); }
static inline size_t km_screen_lds(int K, int d) { return (size_t)km_kp(K) * (km_ks(d) * 2 + 16) + (size_t)2 * K * 4; }
static inline bool km_screen_shape(int64_t N, int d, int K) {
    return N >= KM_SCREEN_MIN_N && N < 0x7fffffffLL && K <= KM_MAX_K_SORT && (d + 4) / 4 <= 8 && 3 * d + 3 <= 96 &&
           km_screen_lds(K, d) <= (size_t)KM_SCREEN_LDS;
}
static inline size_t km_off_stat(int64_t N, int K) { return km_al(km_total(N, K)); }
static inline size_t km_off_mu(int64_t N, int K) { return km_off_stat(N, K) + 256; }
static inline size_t km_off_cb(int64_t N, int K) { return km_off_mu(N, K) + 256; }
static inline size_t km_off_xn2(int64_t N, int K, int d) { return km_off_cb(N, K) + km_al((size_t)km_kp(K) * km_ks(d) * 2); }
static inline size_t km_off_list(int64_t N, int K, int d) { return km_off_xn2(N, K, d) + km_al((size_t)N * 4); }
static inline size_t km_off_xb(int64_t N, int K, int d) { return km_off_list(N, K, d) + km_al((size_t)N * 4); }
static inline size_t km_total_screen(int64_t N, int K, int d) { return km_off_xb(N, K, d) + km_al((size_t)N * km_ks(d) * 2); }

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
    const size_t lds_count = (size_t)4 * K * sizeof(int), lds_place = (size_t)9 * K * sizeof(int);
    const bool fuse_count = mfma_e && sorted && lds_count <= 48 * 1024;   // the E step counts its own labels
    if (sorted && lds_place > 48 * 1024) {
        static std::atomic<unsigned long long> attr_set{0};
        if (sober_attr_needed(attr_set)) {
            HIP_TRY(hipFuncSetAttribute((const void*)k_km_count, hipFuncAttributeMaxDynamicSharedMemorySize, 5 * KM_MAX_K_SORT * 4));
            HIP_TRY(hipFuncSetAttribute((const void*)k_km_place, hipFuncAttributeMaxDynamicSharedMemorySize, 9 * KM_MAX_K_SORT * 4));
            sober_attr_done(attr_set);
        }
    }
    // the screened E step (header): needs the larger workspace sober_kmeans_ws_bytes asks for at this shape
    const bool screen = mfma_e && fuse_count && km_screen_shape(N, d, K) && ws_bytes >= (int64_t)km_total_screen(N, K, d);
    const int ks = km_ks(d), nk = ks / 32;
    KmStat* stat = screen ? (KmStat*)((char*)ws + km_off_stat(N, K)) : nullptr;
    double* mu = screen ? (double*)((char*)ws + km_off_mu(N, K)) : nullptr;
    unsigned short* Cb = screen ? (unsigned short*)((char*)ws + km_off_cb(N, K)) : nullptr;
    float* xn2 = screen ? (float*)((char*)ws + km_off_xn2(N, K, d)) : nullptr;
    int32_t* flist = screen ? (int32_t*)((char*)ws + km_off_list(N, K, d)) : nullptr;
    uint4* Xb = screen ? (uint4*)((char*)ws + km_off_xb(N, K, d)) : nullptr;
    const size_t lds_screen = km_screen_lds(K, d);
    if (screen) {
        static std::atomic<unsigned long long> attr_screen{0};
        if (sober_attr_needed(attr_screen)) {
#define KM_SATTR(NK_, PB_, IB_) HIP_TRY(hipFuncSetAttribute((const void*)k_kmeans_screen<NK_, PB_, IB_>, hipFuncAttributeMaxDynamicSharedMemorySize, KM_SCREEN_LDS))
            KM_SATTR(1, 4, 7); KM_SATTR(2, 4, 7); KM_SATTR(3, 4, 7); KM_SATTR(1, 2, 7); KM_SATTR(2, 2, 7); KM_SATTR(3, 2, 7);
            KM_SATTR(1, 4, 8); KM_SATTR(2, 4, 8); KM_SATTR(3, 4, 8); KM_SATTR(1, 2, 8); KM_SATTR(2, 2, 8); KM_SATTR(3, 2, 8);
#undef KM_SATTR
            sober_attr_done(attr_screen);
        }
        hipLaunchKernelGGL(k_km_mu, dim3(1), dim3(1024), 0, st, X, K, d, mu, stat);
        LAUNCH_CHECK();
        hipLaunchKernelGGL(k_km_cprep, dim3((unsigned)Kp), dim3(128), 0, st, cent, K, d, ks, mu, Cb, stat);
        LAUNCH_CHECK();
        const int ppw = 256 / (4 * nk);                      // points per workgroup of k_km_xprep
        const dim3 xgrid((unsigned)((N + ppw - 1) / ppw));
        switch (nk) {
            case 1: hipLaunchKernelGGL((k_km_xprep<1>), xgrid, dim3(256), 0, st, X, N, d, mu, Xb, xn2); break;
            case 2: hipLaunchKernelGGL((k_km_xprep<2>), xgrid, dim3(256), 0, st, X, N, d, mu, Xb, xn2); break;
            default: hipLaunchKernelGGL((k_km_xprep<3>), xgrid, dim3(256), 0, st, X, N, d, mu, Xb, xn2); break;
        }
        LAUNCH_CHECK();
    }
    for (int it = 0; it < iters; ++it) {
        if (screen) {
            const int pb = N >= 512 * 512 ? 4 : 2;              // (fewer points per workgroup for a pool that would not fill the chip)
            const int64_t n_pairs = (N + 128 * pb - 1) / (128 * pb);
            const dim3 sgrid((unsigned)(n_pairs < 512 ? n_pairs : 512));
#define KM_SCASE(NK_, PB_) do { if (Kp <= 512) hipLaunchKernelGGL((k_kmeans_screen<NK_, PB_, 7>), sgrid, dim3(KM_SCREEN_T), lds_screen, st, Xb, xn2, N, \
                                              (const uint4*)Cb, K, Kp, stat, it & 1, labels, ucount, n_units, flist); \
                                 else hipLaunchKernelGGL((k_kmeans_screen<NK_, PB_, 8>), sgrid, dim3(KM_SCREEN_T), lds_screen, st, Xb, xn2, N, \
                                              (const uint4*)Cb, K, Kp, stat, it & 1, labels, ucount, n_units, flist); } while (0)
            if (pb == 4) { if (nk == 1) KM_SCASE(1, 4); else if (nk == 2) KM_SCASE(2, 4); else KM_SCASE(3, 4); }
            else { if (nk == 1) KM_SCASE(1, 2); else if (nk == 2) KM_SCASE(2, 2); else KM_SCASE(3, 2); }
#undef KM_SCASE
            LAUNCH_CHECK();
            // the list: the FP64 kernel with its exact re-check, on a grid that strides over however many there are
#define KM_CASE(T) case T: hipLaunchKernelGGL((k_kmeans_assign_mfma<T, true>), dim3(256), dim3(256), 0, st, X, N, d, cent, Caug, K, Kp, \
                                              labels, ucount, n_units, flist, &stat->n_list, &stat->nan_first[it & 1]); break;
            switch (kt) { KM_CASE(1) KM_CASE(2) KM_CASE(3) KM_CASE(4) KM_CASE(5) KM_CASE(6) KM_CASE(7) KM_CASE(8) default: break; }
#undef KM_CASE
        } else if (mfma_e) {
            const int64_t n_waves = (N + 16 * KM_PB - 1) / (16 * KM_PB);
            const dim3 grid((unsigned)((n_waves + 3) / 4));
#define KM_CASE(T) case T: hipLaunchKernelGGL((k_kmeans_assign_mfma<T, false>), grid, dim3(256), fuse_count ? lds_count : 0, st, X, N, d, \
                                              cent, Caug, K, Kp, labels, fuse_count ? ucount : (int32_t*)nullptr, n_units, \
                                              (const int32_t*)nullptr, (const unsigned*)nullptr, (const int*)nullptr); break;
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
            // (+ KM_BIG_SPLIT workgroups for a cluster that holds more than half the pool; they leave at once otherwise)
            hipLaunchKernelGGL((k_kmeans_update_sorted<DT>), dim3(K + (stat ? KM_BIG_SPLIT : 0)), dim3(KM_UPD_T), 0, st, X, d, K, tot,
                               order, cent, Caug, 4 * kt, (const double*)mu, Cb, ks, stat, (it + 1) & 1, stat ? N : (int64_t)0,
                               (double*)flist);      // (the list is consumed by now: KM_BIG_SPLIT x 32 doubles of it)
        } else {
            hipLaunchKernelGGL((k_kmeans_update<DT>), dim3(K), dim3(256), 0, st, X, N, d, labels, cent);
        }
        LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace sober

extern "C" int64_t sober_kmeans_ws_bytes(int64_t N, int d, int K) {
    if (N <= 0 || K <= 0 || d <= 0 || N >= 0x7fffffffLL) return 8;
    if (sober::km_screen_shape(N, d, K) && N * ((d + 4) / 4) >= sober::KM_SCREEN_MIN_WORK)
        return (int64_t)sober::km_total_screen(N, K, d);
    return (int64_t)sober::km_total(N, K);
}

// the workspace that selects the screened E step wherever the shape allows it (sober_kmeans_stat_offset >= 0), also
// below the size from which it pays -- what sober_kmeans_ws_bytes returns there; 0: not a screened shape
extern "C" int64_t sober_kmeans_ws_bytes_screened(int64_t N, int d, int K) {
    if (N <= 0 || K <= 0 || d <= 0 || !sober::km_screen_shape(N, d, K)) return 0;
    return (int64_t)sober::km_total_screen(N, K, d);
}

extern "C" int64_t sober_kmeans_stat_offset(int64_t N, int d, int K) {
    if (N <= 0 || K <= 0 || d <= 0 || !sober::km_screen_shape(N, d, K)) return -1;
    return (int64_t)sober::km_off_stat(N, K) + 28;
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
