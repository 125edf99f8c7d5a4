// Dense FP64 Cholesky for the Nystrom side of the path, as ONE persistent workgroup (blocked,
// right-looking; panel in LDS, trailing update register-tiled), plus the row-wise triangular solve
// that turns it into a QR factorisation (CholeskyQR).
//
//  * SOBER/_utils.py:117-157 (is_psd / make_cov_psd) probes `torch.linalg.cholesky(cov + jitter I)` on
//    the N_nys x N_nys Gram up to 12 times; only success/failure is used.  k_chol answers that on
//    the device (info = 0: positive definite; info = j+1: the leading minor of order j+1 is not).
//  * torch.svd_lowrank (SOBER/_rchq.py:37) orthonormalises five M x q blocks with Householder QR; an
//    orthonormal basis of the same flag of subspaces comes from Q = Y R^-1 with R^T R = Y^T Y
//    (applied twice for orthogonality to rounding); Q differs from LAPACK's only by column signs,
//    which cancel in U = Q U_B.
#include "common.hpp"

namespace sober {

constexpr int CH_NB = 32;            // panel width
constexpr int CH_T = 1024;
constexpr int CH_MAXN = 608;         // panel (n - NB) x NB doubles + diagonal block must fit in 160 KiB LDS

// A (n x n, row-major, ld) is overwritten by L in its lower triangle (upper triangle untouched).
// shift is added to the diagonal on the fly (the jitter rung), so the caller's matrix can be probed
// at several rungs from a scratch copy.
// Batched form (gridDim.x > 1): workgroup b factorises (src + shifts[b] I) in its own n x n slab of
// `A` (slab stride n*ld doubles) and reports info[b]; used to probe every rung of the jitter ladder of
// SOBER/_utils.py:145-156 in ONE launch (the rungs are independent).
__global__ __launch_bounds__(CH_T) void k_chol(double* __restrict__ A, int n, int ld, double shift,
                                              int32_t* __restrict__ info, double* __restrict__ min_pivot,
                                              const double* __restrict__ src, int lds_src,
                                              const double* __restrict__ shifts) {
    extern __shared__ double lds[];
    if (src != nullptr) {                     // batched: copy the lower triangle into my slab first
        A += (size_t)blockIdx.x * n * ld;
        info += blockIdx.x;
        if (min_pivot) min_pivot += blockIdx.x;
        shift = shifts[blockIdx.x];
        for (int t = threadIdx.x; t < n * n; t += CH_T) {
            const int i = t / n, j = t % n;
            if (j <= i) A[(size_t)i * ld + j] = src[(size_t)i * lds_src + j];
        }
        __threadfence_block();
        __syncthreads();
    }
    double* D = lds;                         // NB x (NB+1)
    double* P = lds + CH_NB * (CH_NB + 1);   // (n - kb - nb) x (NB+1) panel, padded against bank conflicts
    __shared__ int s_fail;
    __shared__ double s_minp;
    const int tid = threadIdx.x;
    constexpr int LDP = CH_NB + 1;
    if (tid == 0) { s_fail = 0; s_minp = __builtin_inf(); }
    __syncthreads();

    for (int kb = 0; kb < n; kb += CH_NB) {
        const int nb = min(CH_NB, n - kb);
        const int nr = n - kb - nb;                       // rows below the diagonal block
        // ---- diagonal block -> LDS (lower part), + shift
        for (int t = tid; t < nb * nb; t += CH_T) {
            const int i = t / nb, j = t % nb;
            double v = (j <= i) ? A[(size_t)(kb + i) * ld + kb + j] : 0.0;
            if (i == j) v += shift;
            D[i * LDP + j] = v;
        }
        __syncthreads();
        // ---- unblocked Cholesky of D by the first wave, IN REGISTERS: lane = row, d[c] = column c of my row.
        // Column j: pivot and the multipliers L[c][j] are wave-uniform v_readlane broadcasts, the rank-1 update
        // is one FMA per remaining column -- no LDS round trips inside the 32-step dependency chain.
        if (tid < 64) {
            const int i = tid;
            double d[CH_NB];
#pragma unroll
            for (int c = 0; c < CH_NB; ++c) d[c] = (i < nb && c <= i) ? D[i * LDP + c] : 0.0;
            int fail = 0;
            double minp = s_minp;
#pragma unroll
            for (int j = 0; j < CH_NB; ++j) {
                if (j < nb && fail == 0) {                // uniform
                    const double djj = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(d[j]), j),
                                                        __builtin_amdgcn_readlane(__double2loint(d[j]), j));
                    if (!(djj > 0.0)) {                   // also catches NaN
                        fail = kb + j + 1;
                    } else {
                        const double l = sqrt(djj);
                        minp = fmin(minp, djj);
                        d[j] = (i == j) ? l : ((i > j) ? d[j] / l : 0.0);
#pragma unroll
                        for (int c = j + 1; c < CH_NB; ++c) {
                            const double lcj = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(d[j]), c),
                                                                __builtin_amdgcn_readlane(__double2loint(d[j]), c));
                            d[c] = fma(-d[j], lcj, d[c]);  // rows i < c hold unused upper-triangle values
                        }
                    }
                }
            }
            if (i < nb) {
#pragma unroll
                for (int c = 0; c < CH_NB; ++c)
                    if (c <= i) D[i * LDP + c] = d[c];
            }
            if (i == 0) { s_fail = fail; s_minp = minp; }
        }
        __syncthreads();
        if (s_fail != 0) break;                            // uniform
        // ---- write L_kk back
        for (int t = tid; t < nb * nb; t += CH_T) {
            const int i = t / nb, j = t % nb;
            if (j <= i) A[(size_t)(kb + i) * ld + kb + j] = D[i * LDP + j];
        }
        if (nr == 0) break;
        // ---- panel: rows below solve x L_kk^T = a  (one row per thread), kept in LDS
        for (int r = tid; r < nr; r += CH_T) {
            double x[CH_NB];
            const double* a = A + (size_t)(kb + nb + r) * ld + kb;
#pragma unroll
            for (int j = 0; j < CH_NB; ++j) x[j] = (j < nb) ? a[j] : 0.0;
#pragma unroll
            for (int j = 0; j < CH_NB; ++j) {
                if (j < nb) {
                    double s = x[j];
#pragma unroll
                    for (int k = 0; k < CH_NB; ++k)
                        if (k < j) s = fma(-x[k], D[j * LDP + k], s);
                    x[j] = s / D[j * LDP + j];
                }
            }
            double* o = A + (size_t)(kb + nb + r) * ld + kb;
#pragma unroll
            for (int j = 0; j < CH_NB; ++j) {
                if (j < nb) o[j] = x[j];
                P[r * LDP + j] = x[j];
            }
        }
        __syncthreads();
        // ---- trailing update (lower triangle): A[r][c] -= P[r] . P[c], 4x4 register tiles
        const int nt = (nr + 3) / 4;                       // tiles per side
        const int ntri = nt * (nt + 1) / 2;
        for (int t = tid; t < ntri; t += CH_T) {
            // (tr, tc) with tc <= tr from the linear index
            int tr = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
            while ((tr + 1) * (tr + 2) / 2 <= t) ++tr;
            while (tr * (tr + 1) / 2 > t) --tr;
            const int tc = t - tr * (tr + 1) / 2;
            double acc[4][4];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
            const int r0 = tr * 4, c0 = tc * 4;
            for (int k = 0; k < nb; ++k) {
                double pr[4], pc[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    pr[a] = P[min(r0 + a, nr - 1) * LDP + k];
                    pc[a] = P[min(c0 + a, nr - 1) * LDP + k];
                }
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[a][b] = fma(pr[a], pc[b], acc[a][b]);
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int r = r0 + a, c = c0 + b;
                    if (r < nr && c <= r) A[(size_t)(kb + nb + r) * ld + kb + nb + c] -= acc[a][b];
                }
        }
        __threadfence_block();
        __syncthreads();
    }
    if (tid == 0) {
        *info = s_fail;
        if (min_pivot) *min_pivot = s_minp;
    }
}

// Small SPD matrix (q <= 128) entirely in LDS: unblocked right-looking Cholesky with the whole
// workgroup; writes L (q x q lower triangular, row-major).
__global__ __launch_bounds__(1024) void k_chol_small(const double* __restrict__ G, int q, int ldg,
                                                         double* __restrict__ Rinv, int ldr,
                                                         int32_t* __restrict__ info,
                                                         double* __restrict__ min_pivot) {
    extern __shared__ double sm[];
    const int LDS_ = q + 1;
    double* L = sm;                      // q x (q+1)
    __shared__ int s_fail;
    __shared__ double s_minp;
    const int tid = threadIdx.x, nt = blockDim.x;
    if (tid == 0) { s_fail = 0; s_minp = __builtin_inf(); }
    for (int t = tid; t < q * q; t += nt) {
        const int i = t / q, j = t % q;
        L[i * LDS_ + j] = (j <= i) ? G[(size_t)i * ldg + j] : 0.0;
    }
    __syncthreads();
    for (int j = 0; j < q; ++j) {
        const double djj = L[j * LDS_ + j];
        if (!(djj > 0.0)) {
            if (tid == 0) s_fail = j + 1;
            break;                                            // uniform: every thread read the same value
        }
        const double l = sqrt(djj);
        __syncthreads();                                      // everyone has read d_jj before it changes
        if (tid == 0) { L[j * LDS_ + j] = l; s_minp = fmin(s_minp, djj); }
        for (int i = j + 1 + tid; i < q; i += nt) L[i * LDS_ + j] /= l;
        __syncthreads();
        // trailing update, lower triangle: element (i, c), j < c <= i; 32 x 32 thread grid, no index division
        {
            const int ty = tid >> 5, tx = tid & 31;
            for (int i = j + 1 + ty; i < q; i += 32) {
                const double lij = L[i * LDS_ + j];
                for (int c = j + 1 + tx; c <= i; c += 32)
                    L[i * LDS_ + c] = fma(-lij, L[c * LDS_ + j], L[i * LDS_ + c]);
            }
        }
        __syncthreads();
    }
    __syncthreads();
    if (s_fail == 0) {
        for (int t = tid; t < q * q; t += nt) {
            const int i = t / q, j = t % q;
            Rinv[(size_t)i * ldr + j] = (j <= i) ? L[i * LDS_ + j] : 0.0;      // output: L (lower triangular)
        }
    }
    if (tid == 0) {
        *info = s_fail;
        if (min_pivot) *min_pivot = s_minp;
    }
}

// Q[r, :] = Y[r, :] R^-1 with R = L^T (L lower, q x q <= 128): ONE WAVE PER ROW.  Forward substitution
// x_j = (y_j - sum_{k<j} x_k L[j][k]) / L[j][j]; lane l holds x_l and x_{l+64}, the dot product over k is
// split across the lanes and reduced with DPP; L is staged in LDS once per workgroup.
__device__ __forceinline__ double trsm_wsum(double v) {
    int lo, hi;
#define TS_DPP(CTRL)                                                                   \
    lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);     \
    hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);     \
    v += __hiloint2double(hi, lo);
    TS_DPP(0x128) TS_DPP(0x124) TS_DPP(0x122) TS_DPP(0x121)                            // row_ror 8,4,2,1
#undef TS_DPP
    const double a = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 0),
                                      __builtin_amdgcn_readlane(__double2loint(v), 0));
    const double b = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 16),
                                      __builtin_amdgcn_readlane(__double2loint(v), 16));
    const double c = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 32),
                                      __builtin_amdgcn_readlane(__double2loint(v), 32));
    const double d = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 48),
                                      __builtin_amdgcn_readlane(__double2loint(v), 48));
    return ((a + b) + c) + d;
}

template <bool L_IN_LDS>
__global__ __launch_bounds__(256) void k_trsm_rows(const double* __restrict__ Y, int64_t m, int q, int ldy,
                                                   const double* __restrict__ L, int ldl,
                                                   double* __restrict__ Q, int ldq) {
    extern __shared__ double sl[];                   // q x (q+1) when L_IN_LDS (q <= 128)
    const int LDL = L_IN_LDS ? q + 1 : ldl;
    if constexpr (L_IN_LDS) {
        for (int t = threadIdx.x; t < q * q; t += blockDim.x) {
            const int i = t / q, j = t % q;
            sl[i * LDL + j] = (j <= i) ? L[(size_t)i * ldl + j] : 0.0;
        }
        __syncthreads();
    }
    const double* Lp = L_IN_LDS ? sl : L;            // q <= 256: rows of L straight from L2
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= m) return;                              // whole wave
    const double* y = Y + r * ldy;
    double x[4] = {0.0, 0.0, 0.0, 0.0};              // x[lane + 64 t] (0 until solved)
    double yv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) yv[t] = (lane + 64 * t < q) ? y[lane + 64 * t] : 0.0;
    for (int j = 0; j < q; ++j) {
        const double* lj = Lp + (size_t)j * LDL;     // row j of L; entries right of the diagonal meet x = 0
        double part = 0.0;
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (64 * t < q) part = fma(x[t], (lane + 64 * t <= j) ? lj[lane + 64 * t] : 0.0, part);
        const double dot = trsm_wsum(part);
        const int jt = j >> 6, jl = j & 63;
        const double ysel = (jt == 0) ? yv[0] : (jt == 1) ? yv[1] : (jt == 2) ? yv[2] : yv[3];
        const double yj = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(ysel), jl),
                                           __builtin_amdgcn_readlane(__double2loint(ysel), jl));
        const double xj = (yj - dot) / lj[j];
#pragma unroll
        for (int t = 0; t < 4; ++t) x[t] = (lane + 64 * t == j) ? xj : x[t];
    }
    double* o = Q + r * ldq;
#pragma unroll
    for (int t = 0; t < 4; ++t)
        if (lane + 64 * t < q) o[lane + 64 * t] = x[t];
}

// cov -> sqrt(nan_to_num(cov) * nan_to_num(cov).T) (SOBER/_utils.py:143-144) and the exact-symmetry
// test of :127 on the input: flag[0] |= 1 if some cov[i][j] != cov[j][i].
__global__ __launch_bounds__(256) void k_abs_sym(const double* __restrict__ C, int n, int ld,
                                                 double* __restrict__ out, int ldo, int32_t* __restrict__ flag) {
    // 32 x 32 tile and its mirror image through LDS: both reads are row-contiguous
    __shared__ double tm[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    for (int r = ty; r < 32; r += 8) {
        const int i = j0 + r, j = i0 + tx;                            // mirror tile element C[j0 + r][i0 + tx]
        tm[r][tx] = (i < n && j < n) ? C[(size_t)i * ld + j] : 0.0;
    }
    __syncthreads();
    bool asym = false;
    const double big = 1.7976931348623157e308;
    for (int r = ty; r < 32; r += 8) {
        const int i = i0 + r, j = j0 + tx;
        if (i < n && j < n) {
            double a = C[(size_t)i * ld + j], b = tm[tx][r];          // C[j][i]
            asym |= !(a == b);
            a = (a != a) ? 0.0 : fmin(fmax(a, -big), big);            // torch.nan_to_num
            b = (b != b) ? 0.0 : fmin(fmax(b, -big), big);
            out[(size_t)i * ldo + j] = sqrt(a * b);
        }
    }
    if (__syncthreads_or(asym) && threadIdx.x == 0) atomicOr(flag, 1);
}

// the jitter ladder of make_cov_psd (SOBER/_utils.py:151-152), k rungs at once with the reference's own
// sequence of roundings: jitter = 1e-5; repeat k times { diag += jitter; jitter *= 2 }
__global__ void k_jitter_ladder(double* __restrict__ A, int n, int ld, int k) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double d = A[(size_t)i * ld + i], jit = 1e-5;
    for (int t = 0; t < k; ++t) { d = __dadd_rn(d, jit); jit = __dmul_rn(jit, 2.0); }
    A[(size_t)i * ld + i] = d;
}

}  // namespace sober

extern "C" int sober_chol_max_n(void) { return sober::CH_MAXN; }

extern "C" int sober_cholesky(double* A, int n, int ld, double shift, int32_t* info, double* min_pivot,
                              void* stream) {
    if (!A || !info || n <= 0 || ld < n) return SOBER_E_ARG;
    if (n > sober::CH_MAXN) return SOBER_E_DIM;
    const int nr = n > sober::CH_NB ? n - sober::CH_NB : 0;
    const size_t bytes = ((size_t)sober::CH_NB * (sober::CH_NB + 1) + (size_t)nr * (sober::CH_NB + 1)) * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void*)sober::k_chol, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    160 * 1024 - 64));
        attr_set = true;
    }
    hipLaunchKernelGGL(sober::k_chol, dim3(1), dim3(sober::CH_T), bytes, (hipStream_t)stream, A, n, ld, shift, info,
                       min_pivot, (const double*)nullptr, 0, (const double*)nullptr);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int sober_cholesky_probe(const double* src, int n, int ld_src, const double* shifts, int n_shifts,
                                    double* work, int32_t* info, void* stream) {
    if (!src || !shifts || !work || !info || n <= 0 || ld_src < n || n_shifts <= 0) return SOBER_E_ARG;
    if (n > sober::CH_MAXN) return SOBER_E_DIM;
    const int nr = n > sober::CH_NB ? n - sober::CH_NB : 0;
    const size_t bytes = ((size_t)sober::CH_NB * (sober::CH_NB + 1) + (size_t)nr * (sober::CH_NB + 1)) * sizeof(double);
    HIP_TRY(hipFuncSetAttribute((const void*)sober::k_chol, hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024 - 64));
    hipLaunchKernelGGL(sober::k_chol, dim3(n_shifts), dim3(sober::CH_T), bytes, (hipStream_t)stream, work, n, n, 0.0,
                       info, (double*)nullptr, src, ld_src, shifts);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int sober_chol_small(const double* G, int q, int ldg, double* Rinv, int ldr, int32_t* info,
                                    double* min_pivot, void* stream) {
    if (!G || !Rinv || !info || q <= 0 || q > 128 || ldg < q || ldr < q) return SOBER_E_ARG;
    const size_t bytes = (size_t)q * (q + 1) * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void*)sober::k_chol_small, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    160 * 1024 - 64));
        attr_set = true;
    }
    hipLaunchKernelGGL(sober::k_chol_small, dim3(1), dim3(1024), bytes, (hipStream_t)stream, G, q, ldg, Rinv, ldr,
                       info, min_pivot);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int sober_trsm_rows(const double* Y, int64_t m, int q, int ldy, const double* L, int ldl, double* Q,
                               int ldq, void* stream) {
    if (!Y || !L || !Q || m <= 0 || q <= 0 || q > 256 || ldy < q || ldl < q || ldq < q) return SOBER_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((m + 3) / 4)), block(256);
    if (q <= 128) {
        const size_t bytes = (size_t)q * (q + 1) * sizeof(double);
        static bool attr_set = false;
        if (!attr_set) {
            HIP_TRY(hipFuncSetAttribute((const void*)sober::k_trsm_rows<true>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
            attr_set = true;
        }
        hipLaunchKernelGGL(sober::k_trsm_rows<true>, grid, block, bytes, st, Y, m, q, ldy, L, ldl, Q, ldq);
    } else {
        hipLaunchKernelGGL(sober::k_trsm_rows<false>, grid, block, 0, st, Y, m, q, ldy, L, ldl, Q, ldq);
    }
    LAUNCH_CHECK();
    return 0;
}

extern "C" int sober_abs_sym(const double* C, int n, int ld, double* out, int ldo, int32_t* flag, void* stream) {
    if (!C || !out || !flag || n <= 0 || ld < n || ldo < n || n > 65535) return SOBER_E_ARG;
    hipLaunchKernelGGL(sober::k_abs_sym, dim3((n + 31) / 32, (n + 31) / 32), dim3(256), 0, (hipStream_t)stream, C, n,
                       ld, out, ldo, flag);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int sober_jitter_ladder(double* A, int n, int ld, int k, void* stream) {
    if (!A || n <= 0 || ld < n || k < 0) return SOBER_E_ARG;
    if (k == 0) return 0;
    hipLaunchKernelGGL(sober::k_jitter_ladder, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, A, n, ld, k);
    LAUNCH_CHECK();
    return 0;
}
