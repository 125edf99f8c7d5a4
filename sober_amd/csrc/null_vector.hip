// The direction of the acquisition-guided branch's second elimination (SOBER/_rchq.py:87-106, :177-196) in ONE launch.
//
// After the Caratheodory step with the objective as one more test function, n1 = b + 1 points are left; the reference stacks
// their b - 1 Nystrom features on a row of ones -- A2, b x (b + 1) -- takes the last right singular vector of its SVD (:88-91)
// and moves the weights along it.  A2 has a one-dimensional null space and the reference fixes the vector's sign by the
// objective and its scale by the ratio test (:92-100): ANY null vector gives its result.  Rounds 2-5 took it from a second
// run of the Caratheodory kernels on the survivors (a 100-step bidiagonalisation + Phi + a pivot launch per level: as long as
// the level's own step).  Here: Gauss-Jordan elimination with partial pivoting on A2 itself, one workgroup of 16 waves --
//   lane <-> row of A2 (function; two 64-row slots: b <= 128), a wave owns the points (columns) w, w + 16, ... (7 slots:
//   n1 <= 112); step k clears point k: its owner picks the unused row with the largest |entry| (wave argmax), publishes the
//   multipliers entry / pivot of ALL other rows and the row's index through LDS (two alternating slots, ONE barrier per
//   step); every wave subtracts multiplier x (the pivot row's entry, one v_readlane pair) from its points BEHIND k -- the
//   points before k are unit vectors already and are never touched again, so the work halves as the steps go;
//   after b steps the last point was never a pivot: v[last] = 1, v[k] = -A[r_k, last] / pivot_k.
// No back substitution (Gauss-Jordan clears the pivot point from earlier rows too), no square roots; ~25 vector
// instructions per wave and step behind the owner's ~50.  The first form (lane <-> point, rows dealt over the waves, every
// row updated at every step) took 1.0 us a step: 16 waves x 63 instructions on the four SIMDs of one compute unit.
// A zero pivot column (the first b points' matrix is singular) reports failure; the caller's host route takes that level.
#include "common.hpp"

namespace sober {

#ifndef NV_WAVES
#define NV_WAVES 16
#endif
constexpr int NV_W = NV_WAVES, NV_RS = 112 / NV_W, NV_Q = 2;          // waves, point slots per wave, 64-row slots

template <int CTRL>
__device__ __forceinline__ double nv_dpp(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double nv_rdlane(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
// the 64 lanes' maximum of a non-negative value, in every lane of row 3 at least; read out from lane 63
__device__ __forceinline__ double nv_wave_max(double v) {
    v = fmax(v, nv_dpp<0x128>(v));                     // row_ror 8, 4, 2, 1: every lane of a 16-lane row holds the row's maximum
    v = fmax(v, nv_dpp<0x124>(v));
    v = fmax(v, nv_dpp<0x122>(v));
    v = fmax(v, nv_dpp<0x121>(v));
    v = fmax(v, nv_dpp<0x142>(v));                     // row_bcast15: rows 1 and 3 take in rows 0 and 2
    v = fmax(v, nv_dpp<0x143>(v));                     // row_bcast31: row 3 takes in rows 0-1
    return nv_rdlane(v, 63);
}

__global__ __launch_bounds__(NV_W * 64) void k_null_vector(const double* __restrict__ X, int ldx, int Nsets, int nfun,
                                                           const int32_t* __restrict__ rank1,
                                                           const int32_t* __restrict__ n_keep1, int n1,
                                                           double* __restrict__ null_row, int32_t* __restrict__ status) {
    __shared__ int s_set[128];                                    // rank (= point) -> set
    __shared__ double s_m[2][128];                                // the step's multipliers, by row
    __shared__ int s_piv[2];                                      // the step's pivot row (-1: none)
    __shared__ double s_pvrow[128];                               // row -> its pivot entry
    __shared__ int s_krow[128];                                   // row -> the point it cleared
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = nfun + 1;                                       // rows of A2: the features, then the ones
    if (*n_keep1 != n1) { if (tid == 0) *status = (*n_keep1 < 0) ? -1 : -2; return; }      // (uniform: the first step's verdict)
    for (int s = tid; s < Nsets; s += NV_W * 64) {
        null_row[s] = 0.0;
        const int r = rank1[s];
        if (r >= 0 && r < n1) s_set[r] = s;
    }
    __syncthreads();
    double a[NV_RS][NV_Q];                                        // a[j][q]: row lane + 64 q of point w + 16 j
    unsigned long long alive[NV_Q];                               // rows not used as a pivot yet
#pragma unroll
    for (int q = 0; q < NV_Q; ++q) {
        const int row = lane + 64 * q;
        const int nq = min(max(b - 64 * q, 0), 64);
        alive[q] = nq >= 64 ? ~0ull : ((1ull << nq) - 1ull);
#pragma unroll
        for (int j = 0; j < NV_RS; ++j) {
            const int p = w + NV_W * j;
            double v = 0.0;
            if (p < n1 && row < b) v = row < nfun ? X[(size_t)s_set[p] * ldx + row] : 1.0;
            a[j][q] = v;
        }
    }
    // a pivot below 1e-13 of the matrix' largest entry is a rounding residue of a rank-deficient A2, not a pivot
    __shared__ double s_amax[NV_W];
    {
        double mx = 0.0;
#pragma unroll
        for (int j = 0; j < NV_RS; ++j) mx = fmax(mx, fmax(fabs(a[j][0]), fabs(a[j][1])));
        mx = nv_wave_max(mx);
        if (lane == 0) s_amax[w] = mx;
    }
    __syncthreads();
    double tiny = 0.0;
#pragma unroll
    for (int i = 0; i < NV_W; ++i) tiny = fmax(tiny, s_amax[i]);
    tiny *= 1e-13;
    bool dead = false;
#pragma unroll
    for (int J = 0; J < NV_RS; ++J) {                             // the pivot point's register slot: static inside this body
        for (int kk = 0; kk < NV_W; ++kk) {
            const int k = NV_W * J + kk;
            if (k >= b || dead) break;                            // (uniform)
            const int sl = k & 1;
            if (w == kk) {                                        // the owner of point k: pivot row, multipliers
                const double c0 = a[J][0], c1 = a[J][1];
                const double m0 = __builtin_amdgcn_inverse_ballot_w64(alive[0]) ? fabs(c0) : 0.0;
                const double m1 = __builtin_amdgcn_inverse_ballot_w64(alive[1]) ? fabs(c1) : 0.0;
                const double M = nv_wave_max(fmax(m0, m1));
                const unsigned long long b0 = __ballot(m0 == M), b1 = __ballot(m1 == M);
                const int lq = ((b0 ? __ffsll((long long)b0) : __ffsll((long long)b1)) - 1) & 63;
                const double p0 = nv_rdlane(c0, lq), p1 = nv_rdlane(c1, lq);
                const double pv = b0 ? p0 : p1;
                double y = __builtin_amdgcn_rcp(pv);                          // 1 / pivot to an ulp: seed + two Newton steps
                double e = fma(-pv, y, 1.0); y = fma(y, e, y);
                e = fma(-pv, y, 1.0); y = fma(y, e, y);
                const bool ok = M > tiny && M < 1e300;                        // (uniform; a NaN fails both)
                const int rr = ok ? (b0 ? lq : 64 + lq) : -1;
                s_m[sl][lane] = (lane == rr) ? 0.0 : c0 * y;                  // (the pivot row itself stays)
                s_m[sl][lane + 64] = (lane + 64 == rr) ? 0.0 : c1 * y;
                if (lane == 0) { s_piv[sl] = rr; if (ok) { s_pvrow[rr] = pv; s_krow[rr] = k; } }
            }
            __syncthreads();
            const int r = s_piv[sl];
            if (r < 0) { dead = true; break; }                    // a zero column: reported below (uniform)
            const double mq0 = s_m[sl][lane], mq1 = s_m[sl][lane + 64];
            const int lp = r & 63;
            const unsigned long long bit = 1ull << lp;
            // my points behind k (slots before J are unit vectors already; in slot J the waves up to the owner's are).
            // ONE uniform branch per step on the pivot row's 64-row slot: two v_readlane per point instead of four and a select.
            // (Variants measured on the 101 x 100 matrix of cfg-2's levels, rocprofv3: rows dealt over the waves and every row
            //  updated at every step 104 us; this layout 76 us; with the next point's owner searching ahead of everybody's
            //  updates 86 us; 8 waves x 14 points 73 us, 4 x 28 85 us -- ~0.7 us a step whatever the split: the step is the
            //  owner's search between two LDS round trips and a barrier)
            if (r < 64) {
                alive[0] &= ~bit;
#pragma unroll
                for (int j = J; j < NV_RS; ++j) {
                    const double pr = (j == J && w <= kk) ? 0.0 : nv_rdlane(a[j][0], lp);
                    a[j][0] = fma(-mq0, pr, a[j][0]);
                    a[j][1] = fma(-mq1, pr, a[j][1]);
                }
            } else {
                alive[1] &= ~bit;
#pragma unroll
                for (int j = J; j < NV_RS; ++j) {
                    const double pr = (j == J && w <= kk) ? 0.0 : nv_rdlane(a[j][1], lp);
                    a[j][0] = fma(-mq0, pr, a[j][0]);
                    a[j][1] = fma(-mq1, pr, a[j][1]);
                }
            }
        }
    }
    __syncthreads();
    int left = __popcll(alive[0]) + __popcll(alive[1]);
    if (dead || left != 0) { if (tid == 0) *status = -3; return; }
    // the point that was never a pivot (index b = n1 - 1): v[k] = -A[r_k, b] / pivot_k by row, v[b] = 1
    const int pb = b;
    if (w == (pb & (NV_W - 1))) {
        const int jb = pb / NV_W;
#pragma unroll
        for (int q = 0; q < NV_Q; ++q) {
            double v = 0.0;
#pragma unroll
            for (int j = 0; j < NV_RS; ++j) v = (j == jb) ? a[j][q] : v;
            const int row = lane + 64 * q;
            if (row < b) null_row[s_set[s_krow[row]]] = -v / s_pvrow[row];
        }
        if (lane == 0) { null_row[s_set[pb]] = 1.0; *status = 0; }
    }
}

}  // namespace sober

using namespace sober;

extern "C" int sober_null_vector_supported(int nfun) { return (nfun >= 1 && nfun + 2 <= NV_W * NV_RS && nfun + 1 <= 64 * NV_Q) ? 1 : 0; }

extern "C" int sober_null_vector(const double* X, int ldx, int Nsets, int nfun, const int32_t* rank1, const int32_t* n_keep1,
                                 int n1, double* null_row, int32_t* status, void* stream) {
    if (!X || !rank1 || !n_keep1 || !null_row || !status || Nsets <= 0 || ldx < nfun || n1 != nfun + 2 || Nsets < n1)
        return SOBER_E_ARG;
    if (!sober_null_vector_supported(nfun)) return SOBER_E_DIM;
    hipLaunchKernelGGL(k_null_vector, dim3(1), dim3(NV_W * 64), 0, (hipStream_t)stream, X, ldx, Nsets, nfun, rank1, n_keep1, n1,
                       null_row, status);
    LAUNCH_CHECK();
    return 0;
}
