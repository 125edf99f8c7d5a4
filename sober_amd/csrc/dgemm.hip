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
    // element (r, c) of op(M) with shape R x C; zero outside
    if (r >= R || c >= C) return 0.0;
    return trans ? M[(size_t)c * ld + r] : M[(size_t)r * ld + c];
}

__global__ __launch_bounds__(256) void k_dgemm(int transa, int transb, int m, int n, int k,
                                               double alpha, const double* __restrict__ A, int lda,
                                               const double* __restrict__ B, int ldb, double beta,
                                               double* __restrict__ C, int ldc) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // workgroup tile 32 x 32; each wave owns one 16 x 16 MFMA tile (the matrices here are small:
    // favour many waves over register blocking)
    const int i0 = blockIdx.y * 32 + (wave >> 1) * 16;
    const int j0 = blockIdx.x * 32 + (wave & 1) * 16;
    if (i0 >= m || j0 >= n) return;
    const int li = lane & 15, lk = lane >> 4;

    double4_t acc = (double4_t){0.0, 0.0, 0.0, 0.0};
    int k0 = 0;
    for (; k0 + 16 <= k; k0 += 16) {             // 4 k-steps per trip: loads issued ahead of the MFMAs
        double af[4], bf[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            af[u] = ld_op(A, lda, transa, i0 + li, k0 + 4 * u + lk, m, k);
            bf[u] = ld_op(B, ldb, transb, k0 + 4 * u + lk, j0 + li, k, n);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af[u], bf[u], acc, 0, 0, 0);
    }
    for (; k0 < k; k0 += 4) {
        const double af = ld_op(A, lda, transa, i0 + li, k0 + lk, m, k);
        const double bf = ld_op(B, ldb, transb, k0 + lk, j0 + li, k, n);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af, bf, acc, 0, 0, 0);
    }

#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = i0 + lk + 4 * r;
        const int col = j0 + li;
        if (row < m && col < n) {
            double* c = C + (size_t)row * ldc + col;
            const double v = alpha * acc[r];
            *c = (beta == 0.0) ? v : fma(beta, *c, v);
        }
    }
}

}  // namespace sober

extern "C" int sober_dgemm(int transa, int transb, int m, int n, int k, double alpha, const double* A,
                           int lda, const double* B, int ldb, double beta, double* C, int ldc,
                           void* stream) {
    if (!A || !B || !C || m <= 0 || n <= 0 || k <= 0) return SOBER_E_ARG;
    if (lda < (transa ? m : k) || ldb < (transb ? k : n) || ldc < n) return SOBER_E_ARG;
    dim3 grid((n + 31) / 32, (m + 31) / 32);
    hipLaunchKernelGGL(sober::k_dgemm, grid, dim3(256), 0, (hipStream_t)stream, transa, transb, m, n, k,
                       alpha, A, lda, B, ldb, beta, C, ldc);
    LAUNCH_CHECK();
    return 0;
}
