// Row-major FP64 GEMM on the CDNA4 matrix cores: v_mfma_f64_16x16x4_f64.
//
// The dense contractions on the recombination path are small (<= ~1000^3): W = S S^T, T = KxX W,
// the posterior correction and the Nystrom projection P G.  One wave computes one 16x16 tile of C;
// fragments are loaded straight from global memory (the operands are L2 resident).
//
// Fragment maps (cdna_hip_programming.md section 3): A: lane l holds A[i = l & 15][k = l >> 4];
// B: lane l holds B[k = l >> 4][j = l & 15]; C/D (f64 form): col = l & 15, row = (l >> 4) + 4 * reg.
#include "common.hpp"

namespace sober {

typedef double double4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double ld_op(const double* M, int ld, int trans, int r, int c, int R, int C) {
    // element (r, c) of op(M) with shape R x C; zero outside.  The load itself is unconditional (clamped
    // index): a load under a branch would make the compiler wait for every outstanding load at the next use.
    const bool ok = (r < R) & (c < C);
    const int rc = min(r, R - 1), cc = min(c, C - 1);
    const double v = trans ? M[(size_t)cc * ld + rc] : M[(size_t)rc * ld + cc];
    return ok ? v : 0.0;
}

// SPLIT = 1: workgroup tile 32 x 32, each wave owns one 16 x 16 MFMA tile over the whole K.
// SPLIT = 4: the workgroup owns ONE 16 x 16 tile and its four waves each take a quarter of K (partial tiles
//            are added through LDS).  The GEMMs on this path are small and latency bound -- a wave's K loop is
//            a chain of L2 round trips -- so for few tiles the chain is cut in four instead.
// Fragments of the next 16 k are in flight while the MFMAs of the current 16 k issue.
template <int SPLIT>
__global__ __launch_bounds__(256) void k_dgemm(int transa, int transb, int m, int n, int k,
                                               double alpha, const double* __restrict__ A, int lda,
                                               const double* __restrict__ B, int ldb, double beta,
                                               double* __restrict__ C, int ldc,
                                               const double* __restrict__ coldiv, double* __restrict__ Ct, int ldct) {
    __shared__ double red[SPLIT == 4 ? 3 * 256 : 1];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int i0, j0, kbeg, kend;
    if constexpr (SPLIT == 4) {
        i0 = blockIdx.y * 16;
        j0 = blockIdx.x * 16;
        const int kq = (((k + 3) / 4) + 15) / 16 * 16;                // quarter of K, multiple of 16
        kbeg = min(wave * kq, k);
        kend = min(kbeg + kq, k);
    } else {
        i0 = blockIdx.y * 32 + (wave >> 1) * 16;
        j0 = blockIdx.x * 32 + (wave & 1) * 16;
        kbeg = 0;
        kend = k;
        if (i0 >= m || j0 >= n) return;
    }
    const int li = lane & 15, lk = lane >> 4;

    double4_t acc = (double4_t){0.0, 0.0, 0.0, 0.0};
    double af[4], bf[4], an[4], bn[4];
#define DG_LOAD(AF, BF, K0)                                                                   \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                           \
        AF[u] = ld_op(A, lda, transa, i0 + li, (K0) + 4 * u + lk, m, kend);                   \
        BF[u] = ld_op(B, ldb, transb, (K0) + 4 * u + lk, j0 + li, kend, n);                   \
    }
    DG_LOAD(af, bf, kbeg)
    for (int k0 = kbeg; k0 < kend; k0 += 16) {
        DG_LOAD(an, bn, k0 + 16)                                      // (zeros beyond kend)
#pragma unroll
        for (int u = 0; u < 4; ++u)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af[u], bf[u], acc, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 4; ++u) { af[u] = an[u]; bf[u] = bn[u]; }
    }
#undef DG_LOAD
    if constexpr (SPLIT == 4) {
        if (wave > 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(wave - 1) * 256 + r * 64 + lane] = acc[r];
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = ((acc[r] + red[r * 64 + lane]) + red[256 + r * 64 + lane]) + red[512 + r * 64 + lane];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = i0 + lk + 4 * r;
        const int col = j0 + li;
        if (row < m && col < n) {
            const double v = alpha * acc[r];
            if (Ct != nullptr) {                                       // transposed, column j divided by coldiv[j]
                Ct[(size_t)col * ldct + row] = coldiv ? v / coldiv[col] : v;
            } else {
                double* c = C + (size_t)row * ldc + col;
                *c = (beta == 0.0) ? v : fma(beta, *c, v);
            }
        }
    }
}

}  // namespace sober

extern "C" int sober_dgemm(int transa, int transb, int m, int n, int k, double alpha, const double* A,
                           int lda, const double* B, int ldb, double beta, double* C, int ldc,
                           void* stream) {
    if (!A || !B || !C || m <= 0 || n <= 0 || k <= 0) return SOBER_E_ARG;
    if (lda < (transa ? m : k) || ldb < (transb ? k : n) || ldc < n) return SOBER_E_ARG;
    const long tiles = (long)((n + 15) / 16) * ((m + 15) / 16);
    if (tiles <= 2048 && k >= 64) {                  // few tiles: cut the K chain in four
        dim3 grid((n + 15) / 16, (m + 15) / 16);
        hipLaunchKernelGGL(sober::k_dgemm<4>, grid, dim3(256), 0, (hipStream_t)stream, transa, transb, m, n, k,
                           alpha, A, lda, B, ldb, beta, C, ldc, (const double*)nullptr, (double*)nullptr, 0);
    } else {
        dim3 grid((n + 31) / 32, (m + 31) / 32);
        hipLaunchKernelGGL(sober::k_dgemm<1>, grid, dim3(256), 0, (hipStream_t)stream, transa, transb, m, n, k,
                           alpha, A, lda, B, ldb, beta, C, ldc, (const double*)nullptr, (double*)nullptr, 0);
    }
    LAUNCH_CHECK();
    return 0;
}

// Ct[j][i] = (A B)[i][j] / coldiv[j]  (A: m x k, B: k x n, both row-major; coldiv may be null): the projection P G and the
// barycentres of SOBER/_rchq.py:151,166 -- division by the set masses, transposed store -- in one launch; the value
// divided is the same alpha = 1 product sober_dgemm would have stored, so the result is bit-identical to the two steps.
extern "C" int sober_dgemm_coldiv_t(int m, int n, int k, const double* A, int lda, const double* B, int ldb,
                                    const double* coldiv, double* Ct, int ldct, void* stream) {
    if (!A || !B || !Ct || m <= 0 || n <= 0 || k <= 0 || lda < k || ldb < n || ldct < m) return SOBER_E_ARG;
    const long tiles = (long)((n + 15) / 16) * ((m + 15) / 16);
    if (tiles <= 2048 && k >= 64) {
        dim3 grid((n + 15) / 16, (m + 15) / 16);
        hipLaunchKernelGGL(sober::k_dgemm<4>, grid, dim3(256), 0, (hipStream_t)stream, 0, 0, m, n, k, 1.0, A, lda, B, ldb,
                           0.0, (double*)nullptr, 0, coldiv, Ct, ldct);
    } else {
        dim3 grid((n + 31) / 32, (m + 31) / 32);
        hipLaunchKernelGGL(sober::k_dgemm<1>, grid, dim3(256), 0, (hipStream_t)stream, 0, 0, m, n, k, 1.0, A, lda, B, ldb,
                           0.0, (double*)nullptr, 0, coldiv, Ct, ldct);
    }
    LAUNCH_CHECK();
    return 0;
}
