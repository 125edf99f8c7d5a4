// K5 + K6 on the device: Tchernychova_Lyons_CAR (SOBER/_rchq.py:224-270) as ONE persistent
// workgroup (1024 threads, everything on chip).
//
// The reference takes the null space of A = [1 | X]^T (m x N, m < N <= 2m) from LAPACK's full SVD:
// Phi = Vh[m:, :].T.  Which basis of the null space comes back decides which points survive
// (SURVEY.md App. C), so "any orthonormal basis" is not good enough.  MKL's gesdd (the LAPACK
// behind torch here) produces Vh[m:, :] = rows m..N-1 of P^T, where P = G(0) G(1) ... G(m-1) is
// the product of the RIGHT Householder reflectors of the Golub-Kahan bidiagonalisation of A
// (dgebrd, lower-bidiagonal case m < N, dlarfg sign convention) -- checked against
// torch.linalg.svd on the reference's own per-level inputs in tests/test_car_algorithm.py (CPU) and
// tests/test_hip_parity.py (GPU).  That product is a deterministic
// function of A, so it can be recomputed here:
//
//   phase 1  bidiagonalise A in LDS (m*N doubles = 160,000 B at batch 100: the whole 160 KiB LDS
//            of one CU, which is why this is a one-workgroup kernel), keeping the right reflectors
//            v_i in place like dgebd2;
//   phase 2  Phi = P [0; I] by backward accumulation, Phi distributed over the VGPRs of 13 waves
//            (8 columns x 8 row groups per wave, cross-group sums by DPP/shuffle: no barriers);
//   phase 3  the N-m pivots of :237-266 on the register-resident Phi: ratio test = wave argmin with
//            first-index tie break, rank-1 elimination, one barrier per pivot.
//
// Limits: N <= 200, m*N <= 20,000, N-m <= 128 (batch <= 100).  Larger batches use the host LAPACK
// path (sober_car_pivot_host).
#include "common.hpp"

namespace sober {

constexpr int CAR_T = 1024;
constexpr int CAR_RP = 25;          // Phi rows per thread: N <= 8 * 25

__device__ __forceinline__ double wsum(double v) {      // butterfly: every lane gets the total
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ double gsum(double v) {      // sum over the 8 row groups (lane bits 3..5)
    v += __shfl_xor(v, 8, 64);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// dlarfg: reflector H = I - tau [1; v][1; v]^T with H [alpha; x] = [beta; 0]
__device__ __forceinline__ void larfg(double alpha, double xnorm, double& beta, double& tau, double& scal) {
    if (xnorm == 0.0) {
        beta = alpha; tau = 0.0; scal = 0.0;
    } else {
        beta = -copysign(hypot(alpha, xnorm), alpha);
        tau = (beta - alpha) / beta;
        scal = 1.0 / (alpha - beta);
    }
}

__global__ __launch_bounds__(CAR_T) void k_car(const double* __restrict__ X, int ldx, int N, int m,
                                               const double* __restrict__ mu_in,
                                               int32_t* __restrict__ keep_rank,
                                               double* __restrict__ w_star,
                                               int32_t* __restrict__ n_keep_out,
                                               double* __restrict__ mu_out) {
    extern __shared__ double lds[];
    double* A = lds;                       // m x N, row-major
    double* taup = lds + (size_t)m * N;    // m
    double* ubuf = taup + m;               // m   (left reflector of the current step)
    double* scal = ubuf + m;               // [0] = tauq

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NC = N - m;

    // ---------------- load A = [1 | X]^T ----------------
    for (int j = tid; j < N; j += CAR_T) A[j] = 1.0;
    for (int t = tid; t < (m - 1) * N; t += CAR_T) {
        const int j = t / (m - 1), i = t % (m - 1);       // X[j][i], coalesced over i
        A[(size_t)(i + 1) * N + j] = X[(size_t)j * ldx + i];
    }
    __syncthreads();

    // ---------------- phase 1: Golub-Kahan bidiagonalisation (dgebd2, m < N) ----------------
    for (int i = 0; i < m; ++i) {
        // ---- right reflector G(i) from row i, columns i..N-1 (every wave computes it redundantly)
        double* rowi = A + (size_t)i * N;
        double vreg[4];
        double ss = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = i + 1 + lane + 64 * q;
            vreg[q] = (c < N) ? rowi[c] : 0.0;
            ss = fma(vreg[q], vreg[q], ss);
        }
        ss = wsum(ss);
        const double alpha = rowi[i];
        double beta, tau, sc;
        larfg(alpha, sqrt(ss), beta, tau, sc);
#pragma unroll
        for (int q = 0; q < 4; ++q) vreg[q] *= sc;
        // apply to rows r > i: each wave owns rows i+1+wave, +16, ... (row-local: no barrier)
        for (int r = i + 1 + wave; r < m; r += 16) {
            double* row = A + (size_t)r * N;
            double a[4];
            double w = (lane == 0) ? row[i] : 0.0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = i + 1 + lane + 64 * q;
                a[q] = (c < N) ? row[c] : 0.0;
                w = fma(a[q], vreg[q], w);
            }
            w = wsum(w);
            const double t = tau * w;
            if (lane == 0) row[i] -= t;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = i + 1 + lane + 64 * q;
                if (c < N) row[c] = fma(-t, vreg[q], a[q]);
            }
        }
        __syncthreads();                                               // (1) rows updated, row i read
        // ---- wave 0: store v_i / taup, build the left reflector H(i) from column i
        if (wave == 0) {
            if (lane == 0) { rowi[i] = beta; taup[i] = tau; }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = i + 1 + lane + 64 * q;
                if (c < N) rowi[c] = vreg[q];
            }
            if (i < m - 1) {
                double ureg[2];
                double s2 = 0.0;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int r = i + 2 + lane + 64 * q;
                    ureg[q] = (r < m) ? A[(size_t)r * N + i] : 0.0;
                    s2 = fma(ureg[q], ureg[q], s2);
                }
                s2 = wsum(s2);
                double beta2, tauq, sc2;
                larfg(A[(size_t)(i + 1) * N + i], sqrt(s2), beta2, tauq, sc2);
                if (lane == 0) { ubuf[0] = 1.0; scal[0] = tauq; }
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int r = i + 2 + lane + 64 * q;
                    if (r < m) ubuf[r - i - 1] = ureg[q] * sc2;
                }
            }
        }
        if (i == m - 1) break;
        __syncthreads();                                               // (2) ubuf / tauq visible
        // ---- apply H(i) to A(i+1:m-1, i+1:N-1): lane = (row group g, column cl), 8 columns a wave
        {
            const double tauq = scal[0];
            const int g = lane >> 3, cl = lane & 7;
            const int L = m - i - 1;                                   // rows i+1 .. m-1
            for (int cb = wave; cb * 8 < N - i - 1; cb += 16) {
                const int c = i + 1 + cb * 8 + cl;
                const bool okc = c < N;
                double part = 0.0;
                for (int k = g; k < L; k += 8)
                    if (okc) part = fma(ubuf[k], A[(size_t)(i + 1 + k) * N + c], part);
                const double t = tauq * gsum(part);
                for (int k = g; k < L; k += 8)
                    if (okc) {
                        double* e = A + (size_t)(i + 1 + k) * N + c;
                        *e = fma(-t, ubuf[k], *e);
                    }
            }
        }
        __syncthreads();                                               // (3) before the next row step
    }
    __syncthreads();

    // ---------------- phase 2: Phi = G(0) ... G(m-1) [0; I]  (N x NC, in registers) ----------------
    const int g = lane >> 3, cl = lane & 7;
    const int col = wave * 8 + cl;                 // my column of Phi
    const bool okcol = col < NC;
    double phi[CAR_RP];
#pragma unroll
    for (int k = 0; k < CAR_RP; ++k) phi[k] = (okcol && (g + 8 * k) == m + col) ? 1.0 : 0.0;
    if (wave * 8 < NC) {
        for (int i = m - 1; i >= 0; --i) {
            const double* vi = A + (size_t)i * N;          // v_i in columns i+1.., implicit 1 at column i
            const double tau = taup[i];
            double vv[CAR_RP];
            double part = 0.0;
#pragma unroll
            for (int k = 0; k < CAR_RP; ++k) {
                const int r = g + 8 * k;
                vv[k] = (r < N && r > i) ? vi[r] : ((r == i) ? 1.0 : 0.0);
                part = fma(vv[k], phi[k], part);
            }
            const double t = tau * gsum(part);
#pragma unroll
            for (int k = 0; k < CAR_RP; ++k) phi[k] = fma(-t, vv[k], phi[k]);
        }
    }
    __syncthreads();                                                   // A is dead from here on

    // ---------------- phase 3: the pivots of SOBER/_rchq.py:237-266 ----------------
    double* colbuf = lds;                  // [2][N]
    double* mubuf = lds + 2 * N;           // [2][N]
    for (int r = tid; r < N; r += CAR_T) mubuf[r] = mu_in[r];
    if (wave == 0 && cl == 0) {
#pragma unroll
        for (int k = 0; k < CAR_RP; ++k)
            if (g + 8 * k < N) colbuf[g + 8 * k] = phi[k];
    }
    __syncthreads();

    int cur = 0;
    for (int s = 0; s < NC; ++s, cur ^= 1) {
        const double* cb = colbuf + cur * N;
        const double* mb = mubuf + cur * N;
        // ratio test: first argmin of mu/Phi[:,0] over Phi[:,0] > 0; a NaN ratio wins (torch.argmin)
        double best = 0.0;
        int piv = -1;
        double ph4[4], mu4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = lane + 64 * q;
            ph4[q] = (r < N) ? cb[r] : 0.0;
            mu4[q] = (r < N) ? mb[r] : 0.0;
            if (ph4[q] > 0.0) {
                const double a = mu4[q] / ph4[q];
                if (piv < 0 || (best == best && (a < best || a != a))) { piv = r; best = a; }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double ob = __shfl_xor(best, o, 64);
            const int op = __shfl_xor(piv, o, 64);
            bool take;
            if (op < 0) take = false;
            else if (piv < 0) take = true;
            else {
                const bool bn = best != best, on = ob != ob;
                if (bn || on) take = on && (!bn || op < piv);           // earliest NaN wins
                else take = (ob < best) || (ob == best && op < piv);
            }
            if (take) { best = ob; piv = op; }
        }
        if (piv < 0) break;                                             // Q6 (:241-242), uniform
        const double alpha = best;
        const double pp = cb[piv];
        // mu[:] = mu - alpha * Phi[:,0]; mu[idx] = 0   (two roundings like the tensor expression)
        if (wave == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = lane + 64 * q;
                if (r < N) mubuf[(cur ^ 1) * N + r] = (r == piv) ? 0.0 : __dsub_rn(mu4[q], __dmul_rn(alpha, ph4[q]));
            }
        }
        // rank-1 elimination of my column (if still alive): Phi[:,c] -= Phi[:,0] * Phi[idx,c]/Phi[idx,0]
        if (okcol && col > s) {
            const int kp = piv >> 3, gp = piv & 7;
            double mine = 0.0;
#pragma unroll
            for (int k = 0; k < CAR_RP; ++k) mine = (k == kp) ? phi[k] : mine;
            const double prow = __shfl(mine, gp * 8 + cl, 64);
            const double qv = prow / pp;
#pragma unroll
            for (int k = 0; k < CAR_RP; ++k) {
                const int r = g + 8 * k;
                if (r < N) phi[k] = (r == piv) ? 0.0 : fma(-qv, cb[r], phi[k]);
            }
            if (col == s + 1) {                                        // next pivot column
#pragma unroll
                for (int k = 0; k < CAR_RP; ++k)
                    if (g + 8 * k < N) colbuf[(cur ^ 1) * N + g + 8 * k] = phi[k];
            }
        }
        __syncthreads();
    }

    // ---------------- output: w_star = mu[mu > 0], idx_star as ranks ----------------
    if (wave == 0) {
        const double* mb = mubuf + cur * N;
        int base = 0;
        for (int q = 0; q < 4; ++q) {
            const int r = lane + 64 * q;
            const double v = (r < N) ? mb[r] : 0.0;
            const bool keep = (r < N) && (v > 0.0);
            const unsigned long long bal = __ballot(keep);
            const int rank = base + __popcll(bal & ((1ull << lane) - 1ull));
            if (r < N) {
                keep_rank[r] = keep ? rank : -1;
                mu_out[r] = v;
                if (keep) w_star[rank] = v;
            }
            base += __popcll(bal);
        }
        if (lane == 0) *n_keep_out = base;
    }
}

}  // namespace sober

extern "C" int sober_car_supported(int N, int m) {
    return (m >= 2 && N > m && N <= 8 * sober::CAR_RP && (int64_t)m * N <= 20000 && N - m <= 128) ? 1 : 0;
}

extern "C" int sober_car_device(const double* X, int ldx, int N, int m, const double* mu_in,
                                int32_t* keep_rank, double* w_star, int32_t* n_keep, double* mu_out,
                                void* stream) {
    if (!X || !mu_in || !keep_rank || !w_star || !n_keep || !mu_out || ldx < m - 1) return SOBER_E_ARG;
    if (!sober_car_supported(N, m)) return SOBER_E_DIM;
    size_t doubles = (size_t)m * N + 2 * (size_t)m + 8;
    if (doubles < 4 * (size_t)N) doubles = 4 * (size_t)N;
    const size_t bytes = doubles * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void*)sober::k_car, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(sober::k_car, dim3(1), dim3(sober::CAR_T), bytes, (hipStream_t)stream, X, ldx, N, m,
                       mu_in, keep_rank, w_star, n_keep, mu_out);
    LAUNCH_CHECK();
    return 0;
}
