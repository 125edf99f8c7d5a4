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
constexpr int CAR_CQ = 13;          // columns per lane in the row sweeps: N <= 16 * 13

// ---- DPP cross-lane helpers (row = 16 lanes).  ds_bpermute-based __shfl costs an LDS round trip
// per step; these are plain VALU moves.
#ifdef CAR_NO_DPP
template <int CTRL>
__device__ __forceinline__ double dpp(double v) {
    const int l = threadIdx.x & 63;
    return __shfl(v, (l & 0x30) | ((l - (CTRL & 15)) & 15), 64);
}
#else
template <int CTRL>
__device__ __forceinline__ double dpp(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
#endif
constexpr int ROR1 = 0x121, ROR2 = 0x122, ROR4 = 0x124, ROR8 = 0x128;

__device__ __forceinline__ double row16_sum(double v) {   // every lane of a 16-lane row gets the row total
    v += dpp<ROR8>(v);
    v += dpp<ROR4>(v);
    v += dpp<ROR2>(v);
    v += dpp<ROR1>(v);
    return v;
}
__device__ __forceinline__ double grp8_sum(double v) {    // lanes {c2 + 2g}: sum over g, same c2
    v += dpp<ROR8>(v);
    v += dpp<ROR4>(v);
    v += dpp<ROR2>(v);
    return v;
}
__device__ __forceinline__ double rdlane(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {    // uniform total over the 64 lanes
    v = row16_sum(v);
    return ((rdlane(v, 0) + rdlane(v, 16)) + rdlane(v, 32)) + rdlane(v, 48);
}

// dlarfg: reflector H = I - tau [1; v][1; v]^T with H [alpha; x] = [beta; 0], ss = |x|^2
__device__ __forceinline__ void larfg(double alpha, double ss, double& beta, double& tau, double& scal) {
    if (ss == 0.0) {
        beta = alpha; tau = 0.0; scal = 0.0;
    } else {
        beta = -copysign(sqrt(fma(alpha, alpha, ss)), alpha);
        tau = (beta - alpha) / beta;
        scal = 1.0 / (alpha - beta);
    }
}

// ratio-test combine: first argmin, a NaN ratio wins (torch.argmin); piv < 0 = no candidate
__device__ __forceinline__ void amin_take(double& best, int& piv, double ob, int op) {
    bool take;
    if (op < 0) take = false;
    else if (piv < 0) take = true;
    else {
        const bool bn = best != best, on = ob != ob;
        if (bn || on) take = on && (!bn || op < piv);
        else take = (ob < best) || (ob == best && op < piv);
    }
    if (take) { best = ob; piv = op; }
}

__global__ __launch_bounds__(CAR_T) void k_car(const double* __restrict__ X, int ldx, int N, int m,
                                               const double* __restrict__ mu_in,
                                               int32_t* __restrict__ keep_rank,
                                               double* __restrict__ w_star,
                                               int32_t* __restrict__ n_keep_out,
                                               double* __restrict__ mu_out,
                                               double* __restrict__ phi_out) {
    extern __shared__ double lds[];
    double* A = lds;                       // m x N, row-major
    double* taup = lds + (size_t)m * N;    // m
    double* ubuf = taup + m;               // m   (left reflector of the current step)
    double* scal = ubuf + m;               // [0] = tauq

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef CAR_STAMPS
    unsigned long long st_[8]; int sti_ = 0;
#define CAR_STAMP() do { st_[sti_++] = __builtin_amdgcn_s_memtime(); st_[sti_++] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define CAR_STAMP() do { } while (0)
#endif
    CAR_STAMP();
    const int l16 = lane & 15, rid = lane >> 4;          // lane in DPP row, DPP row in wave
    const int g = l16 >> 1, c2 = l16 & 1;                // row group / column parity inside a DPP row
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
        // ---- right reflector G(i) from row i, columns i..N-1.  Every DPP row (16 lanes x CAR_CQ
        // columns) holds the whole row: the norm is a row16 all-reduce, computed redundantly.
        double* rowi = A + (size_t)i * N;
        double vreg[CAR_CQ];
        double ss = 0.0;
#pragma unroll
        for (int q = 0; q < CAR_CQ; ++q) {
            const int c = i + 1 + l16 + 16 * q;
            vreg[q] = (c < N) ? rowi[c] : 0.0;
            ss = fma(vreg[q], vreg[q], ss);
        }
        ss = row16_sum(ss);
        double beta, tau, sc;
        larfg(rowi[i], ss, beta, tau, sc);
#pragma unroll
        for (int q = 0; q < CAR_CQ; ++q) vreg[q] *= sc;
        // apply to rows r > i: one matrix row per DPP row, 64 rows per sweep of the workgroup
        for (int r = i + 1 + wave * 4 + rid; r < m; r += 64) {
            double* row = A + (size_t)r * N;
            double a[CAR_CQ];
            double w = (l16 == 0) ? row[i] : 0.0;
#pragma unroll
            for (int q = 0; q < CAR_CQ; ++q) {
                const int c = i + 1 + l16 + 16 * q;
                a[q] = (c < N) ? row[c] : 0.0;
                w = fma(a[q], vreg[q], w);
            }
            const double t = tau * row16_sum(w);
            if (l16 == 0) row[i] -= t;
#pragma unroll
            for (int q = 0; q < CAR_CQ; ++q) {
                const int c = i + 1 + l16 + 16 * q;
                if (c < N) row[c] = fma(-t, vreg[q], a[q]);
            }
        }
        __syncthreads();                                               // (1) rows updated, row i read
        // ---- wave 0: store v_i / taup, build the left reflector H(i) from column i
        if (wave == 0) {
            if (lane == 0) { rowi[i] = beta; taup[i] = tau; }
            if (rid == 0) {
#pragma unroll
                for (int q = 0; q < CAR_CQ; ++q) {
                    const int c = i + 1 + l16 + 16 * q;
                    if (c < N) rowi[c] = vreg[q];
                }
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
                s2 = wave_sum(s2);
                double beta2, tauq, sc2;
                larfg(A[(size_t)(i + 1) * N + i], s2, beta2, tauq, sc2);
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
        // ---- apply H(i) to A(i+1:m-1, i+1:N-1): a wave owns 8 columns (2 per DPP row), the 8 row
        // groups of a column sit in one DPP row -> the column dot product is a DPP reduction
        {
            const double tauq = scal[0];
            const int L = m - i - 1;                                   // rows i+1 .. m-1
            for (int cb = wave; cb * 8 < N - i - 1; cb += 16) {
                const int c = i + 1 + cb * 8 + rid * 2 + c2;
                const bool okc = c < N;
                double part = 0.0;
                for (int k = g; k < L; k += 8)
                    if (okc) part = fma(ubuf[k], A[(size_t)(i + 1 + k) * N + c], part);
                const double t = tauq * grp8_sum(part);
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

    CAR_STAMP();
    // ---------------- phase 2: Phi = G(0) ... G(m-1) [0; I]  (N x NC, in registers) ----------------
    // thread (wave, rid, c2, g) owns rows g, g+8, ... of column wave*8 + rid*2 + c2
    const int col = wave * 8 + rid * 2 + c2;
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
            const double t = tau * grp8_sum(part);
#pragma unroll
            for (int k = 0; k < CAR_RP; ++k) phi[k] = fma(-t, vv[k], phi[k]);
        }
    }
    if (phi_out != nullptr && okcol) {                                 // stage-level test hook
#pragma unroll
        for (int k = 0; k < CAR_RP; ++k)
            if (g + 8 * k < N) phi_out[(size_t)(g + 8 * k) * NC + col] = phi[k];
    }
    __syncthreads();                                                   // A is dead from here on

    CAR_STAMP();
    // ---------------- phase 3: the pivots of SOBER/_rchq.py:237-266 ----------------
    double* colbuf = lds;                  // [2][N]
    double* mubuf = lds + 2 * N;           // [2][N]
    for (int r = tid; r < N; r += CAR_T) mubuf[r] = mu_in[r];
    if (okcol && col == 0) {
#pragma unroll
        for (int k = 0; k < CAR_RP; ++k)
            if (g + 8 * k < N) colbuf[g + 8 * k] = phi[k];
    }
    __syncthreads();

    int cur = 0;
    for (int s = 0; s < NC; ++s, cur ^= 1) {
        const double* cb = colbuf + cur * N;
        const double* mb = mubuf + cur * N;
        // ratio test: first argmin of mu/Phi[:,0] over Phi[:,0] > 0 (every wave, redundantly)
        double best = 0.0;
        int piv = -1;
        double ph4[4], mu4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = lane + 64 * q;
            ph4[q] = (r < N) ? cb[r] : 0.0;
            mu4[q] = (r < N) ? mb[r] : 0.0;
            if (ph4[q] > 0.0) amin_take(best, piv, mu4[q] / ph4[q], r);
        }
#ifndef CAR_ARGMIN_DPP   // the DPP variant mis-combines (value, index) under divergent control flow; keep shuffles
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double ob = __shfl_xor(best, o, 64);
            const int op = __shfl_xor(piv, o, 64);
            amin_take(best, piv, ob, op);
        }
#else
#pragma unroll
        for (int o = 0; o < 4; ++o) {                                   // all-reduce inside each DPP row
            double ob; int op;
            if (o == 0) { ob = dpp<ROR8>(best); op = __builtin_amdgcn_update_dpp(0, piv, ROR8, 0xf, 0xf, false); }
            else if (o == 1) { ob = dpp<ROR4>(best); op = __builtin_amdgcn_update_dpp(0, piv, ROR4, 0xf, 0xf, false); }
            else if (o == 2) { ob = dpp<ROR2>(best); op = __builtin_amdgcn_update_dpp(0, piv, ROR2, 0xf, 0xf, false); }
            else { ob = dpp<ROR1>(best); op = __builtin_amdgcn_update_dpp(0, piv, ROR1, 0xf, 0xf, false); }
            amin_take(best, piv, ob, op);
        }
        {                                                               // then across the four rows
            double b0 = rdlane(best, 0);
            int p0 = __builtin_amdgcn_readlane(piv, 0);
            amin_take(b0, p0, rdlane(best, 16), __builtin_amdgcn_readlane(piv, 16));
            amin_take(b0, p0, rdlane(best, 32), __builtin_amdgcn_readlane(piv, 32));
            amin_take(b0, p0, rdlane(best, 48), __builtin_amdgcn_readlane(piv, 48));
            best = b0; piv = p0;
        }
#endif
        if (piv < 0) break;                                             // Q6 (:241-242), uniform
        const double alpha = best;
        const double pp = cb[piv];
        // mu[:] = mu - alpha * Phi[:,0]; mu[idx] = 0   (two roundings like the tensor expression)
        if (wave == 15) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = lane + 64 * q;
                if (r < N) mubuf[(cur ^ 1) * N + r] = (r == piv) ? 0.0 : __dsub_rn(mu4[q], __dmul_rn(alpha, ph4[q]));
            }
        }
        // rank-1 elimination of my column: Phi[:,c] -= Phi[:,0] * (Phi[idx,c] / Phi[idx,0]).  The pivot-row
        // entry of my column sits in the lane with g == piv % 8 of my own DPP row.
        if (okcol && col > s) {
            const int kp = piv >> 3, gp = piv & 7;
            double mine = 0.0;
#pragma unroll
            for (int k = 0; k < CAR_RP; ++k) mine = (k == kp) ? phi[k] : mine;
            const double prow = __shfl(mine, (lane & 0x30) | (gp << 1) | c2, 64);
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

    CAR_STAMP();
#ifdef CAR_STAMPS
    if (tid == 0 && phi_out != nullptr) for (int q = 0; q < 8; ++q) ((unsigned long long*)phi_out)[q] = st_[q];
#endif
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
    return (m >= 2 && N > m && N <= 8 * sober::CAR_RP && N <= 16 * sober::CAR_CQ && (int64_t)m * N <= 20000 && N - m <= 128) ? 1 : 0;
}

extern "C" int sober_car_device(const double* X, int ldx, int N, int m, const double* mu_in,
                                int32_t* keep_rank, double* w_star, int32_t* n_keep, double* mu_out,
                                double* phi_out, void* stream) {
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
                       mu_in, keep_rank, w_star, n_keep, mu_out, phi_out);
    LAUNCH_CHECK();
    return 0;
}
