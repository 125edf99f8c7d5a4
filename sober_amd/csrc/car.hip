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
constexpr int CAR_PAD = 208;        // >= 16 * CAR_CQ and >= 8 * CAR_RP + 8

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

// ratio-test combine, branch-free (selects only): first argmin, a NaN ratio wins (torch.argmin);
// piv < 0 = no candidate yet
__device__ __forceinline__ void amin_take(double& best, int& piv, double ob, int op) {
    const bool bn = best != best, on = ob != ob;
    const bool lt = (ob < best) | ((ob == best) & (op < piv));
    const bool nn = on & ((!bn) | (op < piv));
    const bool take = (op >= 0) & ((piv < 0) | ((bn | on) ? nn : lt));
    best = take ? ob : best;
    piv = take ? op : piv;
}

__global__ __launch_bounds__(CAR_T) void k_car(const double* __restrict__ X, int ldx, int N, int m,
                                               const double* __restrict__ mu_in,
                                               int32_t* __restrict__ keep_rank,
                                               double* __restrict__ w_star,
                                               int32_t* __restrict__ n_keep_out,
                                               double* __restrict__ mu_out,
                                               double* __restrict__ phi_out) {
    extern __shared__ double lds[];
    // m x N row-major, then CAR_PAD doubles of slack: the register-tiled sweeps read fixed strides
    // (immediate offsets, no per-element clamping -> no address VGPRs) and mask what lies beyond N
    double* A = lds;
    double* taup = lds + (size_t)m * (8 * CAR_RP) + CAR_PAD;    // m
    double* ubuf = taup + m;               // 104: left reflector of the current step over absolute rows
    double* scal = ubuf + 104;             // [0] tauq  [1] tau

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
    constexpr int NS = 8 * CAR_RP;                       // LDS row stride (200): columns N..NS-1 stay zero, so the
                                                         // register-tiled sweeps need no per-element bounds

    // ---------------- load A = [1 | X]^T ----------------
    for (int j = tid; j < m * NS + CAR_PAD; j += CAR_T) A[j] = 0.0;   // padding / over-reads hit zeros, never stale NaNs
    __syncthreads();
    for (int j = tid; j < N; j += CAR_T) A[j] = 1.0;
    for (int t = tid; t < (m - 1) * N; t += CAR_T) {
        const int j = t / (m - 1), i = t % (m - 1);       // X[j][i], coalesced over i
        A[(size_t)(i + 1) * NS + j] = X[(size_t)j * ldx + i];
    }
    __syncthreads();

    // ---------------- phase 1: Golub-Kahan bidiagonalisation (dgebd2, m < N) ----------------
    // per step: (A) wave 0 builds G(i) from row i and stores the FULL reflector vector
    //               v~ = [0 .. 0, 1, v_i, 0 pad] in row i                          | barrier
    //           (B) all waves apply it to rows > i: row -= tau (row . v~) v~        | barrier
    //           (C) wave 0 builds H(i) from column i, full vector u~ in ubuf        | barrier
    //           (D) all waves apply it to columns > i                               | barrier
    // Because v~ / u~ carry their own zeros, (B) and (D) sweep fixed 16- / 8-strided patterns with
    // unconditional LDS traffic; only whole blocks left of / above the diagonal are skipped (uniform).
    for (int i = 0; i < m; ++i) {
        double* rowi = A + (size_t)i * NS;
        if (wave == 0) {                                               // (A)
            double vr[4];
            double ss = 0.0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = i + 1 + lane + 64 * q;
                const double x = rowi[min(c, NS - 1)];
                vr[q] = (c < N) ? x : 0.0;
                ss = fma(vr[q], vr[q], ss);
            }
            ss = wave_sum(ss);
            double beta, tau, sc;
            larfg(rowi[i], ss, beta, tau, sc);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = i + 1 + lane + 64 * q;
                if (c < N) rowi[c] = vr[q] * sc;
            }
            if (lane == 0) { rowi[i] = 1.0; taup[i] = tau; scal[1] = tau; }
            for (int c = lane; c < i; c += 64) rowi[c] = 0.0;
        }
        if (i == m - 1) break;
        __syncthreads();
        {                                                              // (B) one matrix row per DPP row
            const double tau = scal[1];
            const int q0 = i >> 4;                                     // 16-column blocks left of the diagonal: skip
            double vreg[CAR_CQ];
#pragma unroll
            for (int q = 0; q < CAR_CQ; ++q) {
                double x = 0.0;
                if (q >= q0) x = rowi[min(l16 + 16 * q, NS - 1)];      // uniform branch
                vreg[q] = (l16 + 16 * q < NS) ? x : 0.0;
            }
            for (int r = i + 1 + wave * 4 + rid; r < m; r += 64) {
                double* row = A + (size_t)r * NS + l16;
                double a[CAR_CQ];
                double w0 = 0.0, w1 = 0.0;
#pragma unroll
                for (int q = 0; q < CAR_CQ; ++q) {
                    if (q >= q0) {
                        a[q] = row[16 * q];                            // q = 12 over-reads the next row: v~ is 0 there
                        if (q & 1) w1 = fma(a[q], vreg[q], w1); else w0 = fma(a[q], vreg[q], w0);
                    }
                }
                const double t = tau * row16_sum(w0 + w1);
#pragma unroll
                for (int q = 0; q < CAR_CQ - 1; ++q)
                    if (q >= q0) row[16 * q] = fma(-t, vreg[q], a[q]);
                if (l16 < NS - 16 * (CAR_CQ - 1)) row[16 * (CAR_CQ - 1)] = fma(-t, vreg[CAR_CQ - 1], a[CAR_CQ - 1]);
            }
        }
        __syncthreads();
        if (wave == 0) {                                               // (C)
            double ureg[2];
            double s2 = 0.0;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int r = i + 2 + lane + 64 * q;
                const double x = A[(size_t)min(r, m - 1) * NS + i];
                ureg[q] = (r < m) ? x : 0.0;
                s2 = fma(ureg[q], ureg[q], s2);
            }
            s2 = wave_sum(s2);
            double beta2, tauq, sc2;
            larfg(A[(size_t)(i + 1) * NS + i], s2, beta2, tauq, sc2);
            if (lane == 0) scal[0] = tauq;
            // u~ over absolute row indices 0 .. 103: zeros up to row i, 1 at row i+1, u below, zeros from m on
            for (int r = lane; r < 104; r += 64)
                if (r <= i + 1 || r >= m) ubuf[r] = (r == i + 1) ? 1.0 : 0.0;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int r = i + 2 + lane + 64 * q;
                if (r < m) ubuf[r] = ureg[q] * sc2;
            }
        }
        __syncthreads();
        {                                                              // (D) 8 columns per wave sweep
            const double tauq = scal[0];
            const int kk0 = (i + 1) >> 3;                              // 8-row blocks above row i+1: skip
            const int kkf = m >> 3;                                    // blocks kk < kkf are complete (rows < m)
            for (int cb = wave; cb * 8 < N - i - 1; cb += 16) {
                const int c = min(i + 1 + cb * 8 + rid * 2 + c2, N - 1);   // clamped: duplicates are benign
                double* colp = A + (size_t)g * NS + c;
                double av[13];
                double pa = 0.0, pb = 0.0;
#pragma unroll
                for (int kk = 0; kk < 13; ++kk) {
                    if (kk >= kk0 && kk < kkf) {                       // uniform
                        av[kk] = colp[(size_t)kk * 8 * NS];
                        if (kk & 1) pb = fma(ubuf[g + 8 * kk], av[kk], pb); else pa = fma(ubuf[g + 8 * kk], av[kk], pa);
                    }
                }
                double alast = 0.0;
                const bool lastok = (kkf < 13) && (g + 8 * kkf < m);   // the one partial block, per lane
                if (lastok) { alast = colp[(size_t)kkf * 8 * NS]; pa = fma(ubuf[g + 8 * kkf], alast, pa); }
                const double t = tauq * grp8_sum(pa + pb);
#pragma unroll
                for (int kk = 0; kk < 13; ++kk)
                    if (kk >= kk0 && kk < kkf) colp[(size_t)kk * 8 * NS] = fma(-t, ubuf[g + 8 * kk], av[kk]);
                if (lastok) colp[(size_t)kkf * 8 * NS] = fma(-t, ubuf[g + 8 * kkf], alast);
            }
        }
        __syncthreads();
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
            const double* vi = A + (size_t)i * NS + g;     // row i now holds [0.., 1, v_i, 0 pad]; stride-8 walk
            const double tau = taup[i];
            double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
#pragma unroll
            for (int k = 0; k < CAR_RP; ++k) {
                {
                    const double v = vi[8 * k];                        // zero beyond column N-1
                    if ((k & 3) == 0) p0 = fma(v, phi[k], p0);
                    else if ((k & 3) == 1) p1 = fma(v, phi[k], p1);
                    else if ((k & 3) == 2) p2 = fma(v, phi[k], p2);
                    else p3 = fma(v, phi[k], p3);
                }
                if ((k % 6) == 5) __builtin_amdgcn_sched_barrier(0);   // cap load hoisting: VGPR budget is 128
            }
            const double t = tau * grp8_sum((p0 + p1) + (p2 + p3));
#pragma unroll
            for (int k = 0; k < CAR_RP; ++k) {                         // second pass re-reads v_i from LDS:
                phi[k] = fma(-t, vi[8 * k], phi[k]);                   // cheaper than 50 more live VGPRs
                if ((k % 6) == 5) __builtin_amdgcn_sched_barrier(0);
            }
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
    // One barrier per pivot.  Step s: every column owner eliminates with (piv_s, alpha_s) read from LDS;
    // the wave that owns pivot column s+1 then, still inside the same step, publishes that column,
    // applies the mu update of step s (:253-254) and runs the ratio test of step s+1 (:239-247).
    const int NP = 264;                    // padded stride >= N + 64
    double* colbuf = lds;                  // [2][NP]   current / next pivot column (zero beyond N)
    double* mubuf = lds + 2 * NP;          // [2][NP]
    double* pscal = lds + 4 * NP;          // [2][2]    (alpha, piv) of the current / next step
    for (int r = tid; r < 2 * NP; r += CAR_T) colbuf[r] = 0.0;
    __syncthreads();
    for (int r = tid; r < N; r += CAR_T) mubuf[r] = mu_in[r];
    if (okcol && col == 0) {
#pragma unroll
        for (int k = 0; k < CAR_RP; ++k)
            if (g + 8 * k < N) colbuf[g + 8 * k] = phi[k];
    }
    __syncthreads();

    // ratio test on (column cp, weights held in mu4): first argmin of mu/Phi over Phi > 0 (NaN wins)
#define CAR_RATIO_TEST(cp, outp)                                                          \
    {                                                                                     \
        double best_ = 0.0; int piv_ = -1;                                                \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                   \
            const int r = lane + 64 * q;                                                  \
            const double ph = (cp)[r];                                                    \
            const bool ok = (r < N) & (ph > 0.0);                                         \
            amin_take(best_, piv_, mu4[q] / ph, ok ? r : -1);                             \
        }                                                                                 \
        _Pragma("unroll") for (int o = 32; o > 0; o >>= 1) {                              \
            const double ob = __shfl_xor(best_, o, 64);                                   \
            const int op = __shfl_xor(piv_, o, 64);                                       \
            amin_take(best_, piv_, ob, op);                                               \
        }                                                                                 \
        if (lane == 0) { (outp)[0] = best_; (outp)[1] = (double)piv_; }                   \
    }
    if (wave == 0) {
        double mu4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) mu4[q] = mubuf[lane + 64 * q];
        CAR_RATIO_TEST(colbuf, pscal);
    }
    __syncthreads();

    int cur = 0;
    for (int s = 0; s < NC; ++s, cur ^= 1) {
        const double* cb = colbuf + cur * NP;
        const double alpha = pscal[cur * 2];
        const int piv = (int)pscal[cur * 2 + 1];
        if (piv < 0) break;                                             // Q6 (:241-242), uniform
        const int kp = piv >> 3, gp = piv & 7;
        if (okcol && col > s) {
            // rank-1 elimination of my column: Phi[:,c] -= Phi[:,0] * (Phi[idx,c] / Phi[idx,0]).  The
            // pivot-row entry of my column sits in the lane with g == piv % 8 of my own DPP row.
            double mine = 0.0;
            switch (kp) {                                               // uniform: a scalar jump, no select chain
#define CAR_CASE(K) case K: mine = phi[K]; break;
                CAR_CASE(0) CAR_CASE(1) CAR_CASE(2) CAR_CASE(3) CAR_CASE(4) CAR_CASE(5) CAR_CASE(6) CAR_CASE(7)
                CAR_CASE(8) CAR_CASE(9) CAR_CASE(10) CAR_CASE(11) CAR_CASE(12) CAR_CASE(13) CAR_CASE(14)
                CAR_CASE(15) CAR_CASE(16) CAR_CASE(17) CAR_CASE(18) CAR_CASE(19) CAR_CASE(20) CAR_CASE(21)
                CAR_CASE(22) CAR_CASE(23) CAR_CASE(24)
#undef CAR_CASE
                default: break;
            }
            const double prow = __shfl(mine, (lane & 0x30) | (gp << 1) | c2, 64);
            const double qv = prow / cb[piv];
            const double* cbg = cb + g;
#pragma unroll
            for (int k = 0; k < CAR_RP; ++k) {
                phi[k] = fma(-qv, cbg[8 * k], phi[k]);                  // column is zero beyond N
                if ((k % 6) == 5) __builtin_amdgcn_sched_barrier(0);
            }
            if (g == gp) {                                              // Phi[idx, :] = 0 (:266), exactly
                switch (kp) {
#define CAR_CASE(K) case K: phi[K] = 0.0; break;
                    CAR_CASE(0) CAR_CASE(1) CAR_CASE(2) CAR_CASE(3) CAR_CASE(4) CAR_CASE(5) CAR_CASE(6) CAR_CASE(7)
                    CAR_CASE(8) CAR_CASE(9) CAR_CASE(10) CAR_CASE(11) CAR_CASE(12) CAR_CASE(13) CAR_CASE(14)
                    CAR_CASE(15) CAR_CASE(16) CAR_CASE(17) CAR_CASE(18) CAR_CASE(19) CAR_CASE(20) CAR_CASE(21)
                    CAR_CASE(22) CAR_CASE(23) CAR_CASE(24)
#undef CAR_CASE
                    default: break;
                }
            }
            if (col == s + 1) {                                        // next pivot column
#pragma unroll
                for (int k = 0; k < CAR_RP; ++k)
                    if (g + 8 * k < N) colbuf[(cur ^ 1) * NP + g + 8 * k] = phi[k];
            }
        }
        if (wave == min((s + 1) >> 3, 15)) {
            // mu[:] = mu - alpha * Phi[:,0]; mu[idx] = 0   (two roundings like the tensor expression)
            const double* mb = mubuf + cur * NP;
            double mu4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = lane + 64 * q;
                mu4[q] = (r == piv) ? 0.0 : __dsub_rn(mb[r], __dmul_rn(alpha, cb[r]));
                if (r < N) mubuf[(cur ^ 1) * NP + r] = mu4[q];
            }
            if (s + 1 < NC) {
                const double* cn = colbuf + (cur ^ 1) * NP;            // written above by lanes of THIS wave
                CAR_RATIO_TEST(cn, pscal + (cur ^ 1) * 2);
            }
        }
        __syncthreads();
    }
#undef CAR_RATIO_TEST

    CAR_STAMP();
#ifdef CAR_STAMPS
    if (tid == 0 && phi_out != nullptr) for (int q = 0; q < 8; ++q) ((unsigned long long*)phi_out)[q] = st_[q];
#endif
    // ---------------- output: w_star = mu[mu > 0], idx_star as ranks ----------------
    if (wave == 0) {
        const double* mb = mubuf + cur * NP;
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
    return (m >= 2 && N > m && N <= 8 * sober::CAR_RP && N <= 16 * sober::CAR_CQ && m <= 100 && N - m <= 120) ? 1 : 0;
}

extern "C" int sober_car_device(const double* X, int ldx, int N, int m, const double* mu_in,
                                int32_t* keep_rank, double* w_star, int32_t* n_keep, double* mu_out,
                                double* phi_out, void* stream) {
    if (!X || !mu_in || !keep_rank || !w_star || !n_keep || !mu_out || ldx < m - 1) return SOBER_E_ARG;
    if (!sober_car_supported(N, m)) return SOBER_E_DIM;
    size_t doubles = (size_t)m * (8 * sober::CAR_RP) + sober::CAR_PAD + (size_t)m + 104 + 8;
    if (doubles < 4 * 264 + 8) doubles = 4 * 264 + 8;
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
