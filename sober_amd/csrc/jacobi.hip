// Left singular vectors of a small square matrix (q <= 128) on the device: one-sided Jacobi (Hestenes).
//
// torch.svd_lowrank (SOBER/_rchq.py:37; torch/_lowrank.py:165-171) ends with the SVD of the small matrix
// B = Q^H A.  The range finder leaves its q x q triangular factor T (B = T Q2^T, Q2 orthonormal) on the device;
// the left singular vectors of B are those of T.  A LAPACK call on the host costs ~0.9 ms for q = 99 plus two
// PCIe round trips and a stream synchronisation in the middle of the step.  Here: right rotations J make the
// columns of T J mutually orthogonal, T J = U Sigma, so U is read off the normalised columns -- no accumulation
// of J.  A triangular, graded T (this one is a product of Cholesky factors) is the favourable case for the
// method: a handful of sweeps, and the small singular values come out to high RELATIVE accuracy.
//
// One workgroup of 16 waves; the matrix lives column-major in LDS.  A sweep is q_e - 1 rounds of q_e / 2 disjoint
// column pairs (round-robin tournament); a pair belongs to one 16-lane DPP row -- four pairs per wave -- so the three
// dot products of a rotation are in-row DPP reductions and a round costs one workgroup barrier.
#include "common.hpp"

namespace sober {

constexpr int JS_T = 1024;
constexpr int JS_QMAX = 128;
constexpr int JS_E = JS_QMAX / 16;       // row slots per lane

template <int CTRL>
__device__ __forceinline__ double js_dpp(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double js_row16_sum(double v) {      // every lane of a 16-lane row gets the row total
    v += js_dpp<0x128>(v);                                     // row_ror 8, 4, 2, 1
    v += js_dpp<0x124>(v);
    v += js_dpp<0x122>(v);
    v += js_dpp<0x121>(v);
    return v;
}

__global__ __launch_bounds__(JS_T) void k_jacobi_left(const double* __restrict__ T, int q, int ldt,
                                                     double* __restrict__ U, int ldu, double* __restrict__ sigma,
                                                     int32_t* __restrict__ sweeps_out, int max_sweeps) {
    extern __shared__ double js[];                    // column j at js[j * LD + i]
    const int LD = q | 1;                             // odd stride: the four pairs of a wave spread over the banks
    __shared__ double s_sig[JS_QMAX];
    __shared__ int s_rank[JS_QMAX];
    const int tid = threadIdx.x, l16 = tid & 15, grp = tid >> 4;          // 64 groups of 16 lanes
    for (int t = tid; t < q * q; t += JS_T) {
        const int i = t / q, j = t % q;
        js[j * LD + i] = T[(size_t)i * ldt + j];
    }
    __syncthreads();
    const int qe = (q + 1) & ~1, half = qe / 2, nround = qe - 1;
    int sweep = 0;
    for (; sweep < max_sweeps; ++sweep) {
        int rotated = 0;
        for (int t = 0; t < nround; ++t) {
            // round-robin tournament: player qe-1 stays, the others rotate
            int a, b;
            if (grp == 0) { a = qe - 1; b = t; }
            else { a = (t + grp) % nround; b = (t - grp + nround) % nround; }
            if (grp < half && a < q && b < q) {                            // (a padding column never rotates)
                double x[JS_E], y[JS_E];
                double al = 0.0, be = 0.0, ga = 0.0;
#pragma unroll
                for (int e = 0; e < JS_E; ++e) {
                    const int i = l16 + 16 * e;
                    x[e] = (i < q) ? js[a * LD + i] : 0.0;
                    y[e] = (i < q) ? js[b * LD + i] : 0.0;
                    al = fma(x[e], x[e], al);
                    be = fma(y[e], y[e], be);
                    ga = fma(x[e], y[e], ga);
                }
                al = js_row16_sum(al); be = js_row16_sum(be); ga = js_row16_sum(ga);
                if (ga * ga > 1e-30 * (al * be)) {                          // |ga| > 1e-15 sqrt(al be); uniform in the row
                    // tan of the rotation angle, t = sign(d) g / (|d| + sqrt(d^2 + g^2)) with d = be - al, g = 2 ga
                    // (the smaller root: |t| <= 1).  Only c has to be accurate -- s = c t makes c^2 + s^2 = 1 to
                    // rounding whatever t is -- so t comes from the hardware reciprocal / reciprocal square root
                    // with one Newton step each instead of two divisions and a square root.
                    const double d = be - al, g = 2.0 * ga;
                    const double h = fma(d, d, g * g);
                    double r = __builtin_amdgcn_rsq(h);
                    r = fma(0.5 * r, fma(-h * r, r, 1.0), r);
                    const double den = fabs(d) + h * r;
                    double ri = __builtin_amdgcn_rcp(den);
                    ri = fma(fma(-den, ri, 1.0), ri, ri);
                    const double tn = copysign(g, d * g) * ri;            // sign(d) g / den
                    const double c = rsqrt(fma(tn, tn, 1.0)), s = c * tn;
#pragma unroll
                    for (int e = 0; e < JS_E; ++e) {
                        const int i = l16 + 16 * e;
                        if (i < q) {
                            js[a * LD + i] = c * x[e] - s * y[e];
                            js[b * LD + i] = fma(s, x[e], c * y[e]);
                        }
                    }
                    rotated = 1;
                }
            }
            __syncthreads();
        }
        if (!__syncthreads_or(rotated)) { ++sweep; break; }
    }
    // singular values = column norms; descending order by counting; U = normalised columns
    for (int j = grp; j < q; j += JS_T / 16) {
        double nn = 0.0;
#pragma unroll
        for (int e = 0; e < JS_E; ++e) {
            const int i = l16 + 16 * e;
            const double v = (i < q) ? js[j * LD + i] : 0.0;
            nn = fma(v, v, nn);
        }
        nn = js_row16_sum(nn);
        if (l16 == 0) s_sig[j] = sqrt(nn);
    }
    __syncthreads();
    if (tid < q) {
        const double sj = s_sig[tid];
        int r = 0;
        for (int k = 0; k < q; ++k) {
            const double sk = s_sig[k];
            r += (sk > sj) || (sk == sj && k < tid);
        }
        s_rank[tid] = r;
        sigma[r] = sj;
    }
    __syncthreads();
    for (int j = grp; j < q; j += JS_T / 16) {
        const double sj = s_sig[j];
        const double inv = sj > 0.0 ? 1.0 / sj : 0.0;
        const int r = s_rank[j];
#pragma unroll
        for (int e = 0; e < JS_E; ++e) {
            const int i = l16 + 16 * e;
            if (i < q) U[(size_t)i * ldu + r] = js[j * LD + i] * inv;
        }
    }
    if (tid == 0 && sweeps_out) *sweeps_out = sweep;
}

}  // namespace sober

extern "C" int sober_jacobi_left(const double* T, int q, int ldt, double* U, int ldu, double* sigma,
                                 int32_t* sweeps, int max_sweeps, void* stream) {
    if (!T || !U || !sigma || q <= 0 || q > sober::JS_QMAX || ldt < q || ldu < q || max_sweeps <= 0) return SOBER_E_ARG;
    const size_t bytes = (size_t)q * (q | 1) * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void*)sober::k_jacobi_left, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    150 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(sober::k_jacobi_left, dim3(1), dim3(sober::JS_T), bytes, (hipStream_t)stream, T, q, ldt, U, ldu,
                       sigma, sweeps, max_sweeps);
    LAUNCH_CHECK();
    return 0;
}
