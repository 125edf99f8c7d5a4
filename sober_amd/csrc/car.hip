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
constexpr int CAR_P1 = 672 + 64 * 208;  // phase-1 exchange buffers (doubles)

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

// workgroup barrier that waits for this wave's LDS traffic only: global stores (the reflector vectors
// going to scratch) keep draining in the background instead of stalling every step
#define CAR_LDS_BARRIER() do { __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_s_barrier(); } while (0)

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
                                               double* __restrict__ phi_out,
                                               double* __restrict__ vws) {
    extern __shared__ double lds[];
    // LDS map.  Phases 2-3: [0, m*NS + CAR_PAD) = the reflector vectors v~_i as rows (NS = 200 doubles,
    // zero padded; fixed-stride sweeps over-read into zeros), then taup[m].  Phase 1 keeps the matrix
    // itself in REGISTERS and uses the same base region for its small exchange buffers.
    constexpr int NS = 8 * CAR_RP;
    const int REG = max(m * NS + CAR_PAD, CAR_P1);
    double* A = lds;                         // phases 2-3: rows of v~
    double* taup = lds + REG;                // m
    double* scal = taup + m;                 // [0] tauq  [1] tau
    double* vbuf = lds;                      // phase 1: current v~ (208)
    double* ubuf = lds + 208;                //          current u~ (128)
    double* colb = lds + 336;                //          column i of the matrix (128)
    double* zsum = lds + 464;                //          u~^T A per column (208)
    double* zpart = lds + 672;               //          64 x 208 partial column sums

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

    // ---------------- phase 1: Golub-Kahan bidiagonalisation (dgebd2, m < N), matrix in VGPRs ----------------
    // 2-D block-cyclic ownership: thread (R = tid >> 4, C = tid & 15) holds A[r][c] for r in {R, R + 64},
    // c = C + 16 q (q < 13): 26 doubles.  A matrix row lives in ONE DPP row (16 lanes), so row dot products
    // are DPP reductions; column dot products go through a 64 x 208 partial buffer in LDS.  LDS traffic per
    // step is ~4x lower than with the matrix itself in LDS (which is what bounded the previous version).
    // per step: (A) the owner DPP row builds G(i), publishes v~ (LDS + global scratch for phase 2)  | barrier
    //           (B) everyone applies it to its rows > i; column i is published                      | barrier
    //           (C) wave 0 builds H(i) from column i                                                 | barrier
    //           (D) partial column sums -> LDS | barrier | 208 threads finish the sums | barrier | update
    const int R = tid >> 4, C = l16;
    double a0[CAR_CQ], a1[CAR_CQ];
#pragma unroll
    for (int q = 0; q < CAR_CQ; ++q) {
        const int c = C + 16 * q;
        const bool okc = c < N;
        a0[q] = (okc && R < m) ? ((R == 0) ? 1.0 : X[(size_t)c * ldx + (R - 1)]) : 0.0;
        a1[q] = (okc && R + 64 < m) ? X[(size_t)c * ldx + (R + 63)] : 0.0;
    }
#ifdef CAR_STAMPS2
    unsigned long long acc_[6] = {0, 0, 0, 0, 0, 0}, tl_ = __builtin_amdgcn_s_memtime();
#define CAR_SUB(K) do { unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc_[K] += t_ - tl_; tl_ = t_; } while (0)
#else
#define CAR_SUB(K) do { } while (0)
#endif
    for (int i = 0; i < m; ++i) {
        const bool hi = i >= 64;                         // row i is the a1 (hi) or a0 row of DPP row (i & 63)
        const int qi = i >> 4, ci = i & 15;              // column i = lane ci, slot qi
        if (R == (i & 63)) {                                           // (A)
#define CAR_STEP_A(ARR)                                                                       \
            {                                                                                 \
                double ss = 0.0, al = 0.0;                                                    \
                _Pragma("unroll") for (int q = 0; q < CAR_CQ; ++q) {                          \
                    const int c = C + 16 * q;                                                 \
                    ss = (c > i) ? fma(ARR[q], ARR[q], ss) : ss;                              \
                    al = (c == i) ? ARR[q] : al;                                              \
                }                                                                             \
                ss = row16_sum(ss);                                                           \
                al = row16_sum(al);                                                           \
                double beta, tau, sc;                                                         \
                larfg(al, ss, beta, tau, sc);                                                 \
                _Pragma("unroll") for (int q = 0; q < CAR_CQ; ++q) {                          \
                    const int c = C + 16 * q;                                                 \
                    const double v = (c < i) ? 0.0 : ((c == i) ? 1.0 : ARR[q] * sc);          \
                    vbuf[c] = v;                                                              \
                    if (c < NS) vws[(size_t)i * NS + c] = v;                                  \
                }                                                                             \
                if (C == 0) { taup[i] = tau; scal[1] = tau; }                                 \
            }
            if (hi) CAR_STEP_A(a1) else CAR_STEP_A(a0)
#undef CAR_STEP_A
        }
        if (i == m - 1) break;
        CAR_LDS_BARRIER();
        CAR_SUB(0);
        {                                                              // (B)
            const double tau = scal[1];
            double vreg[CAR_CQ];
#pragma unroll
            for (int q = 0; q < CAR_CQ; ++q) vreg[q] = vbuf[C + 16 * q];
            if (R > i && R < m) {
                double w0 = 0.0, w1 = 0.0;
#pragma unroll
                for (int q = 0; q < CAR_CQ; ++q) { if (q & 1) w1 = fma(a0[q], vreg[q], w1); else w0 = fma(a0[q], vreg[q], w0); }
                const double t = tau * row16_sum(w0 + w1);
#pragma unroll
                for (int q = 0; q < CAR_CQ; ++q) a0[q] = fma(-t, vreg[q], a0[q]);
            }
            if (R + 64 > i && R + 64 < m) {
                double w0 = 0.0, w1 = 0.0;
#pragma unroll
                for (int q = 0; q < CAR_CQ; ++q) { if (q & 1) w1 = fma(a1[q], vreg[q], w1); else w0 = fma(a1[q], vreg[q], w0); }
                const double t = tau * row16_sum(w0 + w1);
#pragma unroll
                for (int q = 0; q < CAR_CQ; ++q) a1[q] = fma(-t, vreg[q], a1[q]);
            }
            if (C == ci) {                                             // publish column i
                double x0 = 0.0, x1 = 0.0;
                switch (qi) {                                          // uniform
#define CAR_CASE(K) case K: x0 = a0[K]; x1 = a1[K]; break;
                    CAR_CASE(0) CAR_CASE(1) CAR_CASE(2) CAR_CASE(3) CAR_CASE(4) CAR_CASE(5) CAR_CASE(6)
                    CAR_CASE(7) CAR_CASE(8) CAR_CASE(9) CAR_CASE(10) CAR_CASE(11) CAR_CASE(12)
#undef CAR_CASE
                    default: break;
                }
                colb[R] = x0;
                colb[R + 64] = x1;
            }
        }
        CAR_LDS_BARRIER();
        CAR_SUB(1);
        if (wave == 0) {                                               // (C)
            double ureg[2];
            double s2 = 0.0;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int r = i + 2 + lane + 64 * q;
                const double x = colb[min(r, 127)];
                ureg[q] = (r < m) ? x : 0.0;
                s2 = fma(ureg[q], ureg[q], s2);
            }
            s2 = wave_sum(s2);
            double beta2, tauq, sc2;
            larfg(colb[i + 1], s2, beta2, tauq, sc2);
            if (lane == 0) scal[0] = tauq;
            // u~ over absolute rows 0 .. 127: zeros up to row i, 1 at row i+1, u below, zeros from m on
            for (int r = lane; r < 128; r += 64)
                if (r <= i + 1 || r >= m) ubuf[r] = (r == i + 1) ? 1.0 : 0.0;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int r = i + 2 + lane + 64 * q;
                if (r < m) ubuf[r] = ureg[q] * sc2;
            }
        }
        CAR_LDS_BARRIER();
        CAR_SUB(2);
        const double u0 = ubuf[R], u1 = ubuf[R + 64];                  // (D)
#pragma unroll
        for (int q = 0; q < CAR_CQ; ++q) zpart[R * 208 + C + 16 * q] = fma(u1, a1[q], u0 * a0[q]);
        CAR_LDS_BARRIER();
        CAR_SUB(3);
        if (tid < 832) {                                               // 4 lanes per column, 16 partials each
            const int cc = tid >> 2, part = tid & 3;
            double z0 = 0.0, z1 = 0.0;
#pragma unroll
            for (int w = 0; w < 16; w += 2) {
                z0 += zpart[(part * 16 + w) * 208 + cc];
                z1 += zpart[(part * 16 + w + 1) * 208 + cc];
            }
            double zz = z0 + z1;
            zz += dpp<0xB1>(zz);                                       // quad_perm [1,0,3,2]
            zz += dpp<0x4E>(zz);                                       // quad_perm [2,3,0,1]
            if (part == 0) zsum[cc] = zz;
        }
        CAR_LDS_BARRIER();
        CAR_SUB(4);
        {
            const double tq = scal[0];
            const double f0 = tq * u0, f1 = tq * u1;
#pragma unroll
            for (int q = 0; q < CAR_CQ; ++q) {
                const int c = C + 16 * q;
                const double z = (c > i) ? zsum[c] : 0.0;              // H(i) acts on columns i+1 .. N-1 only
                a0[q] = fma(-f0, z, a0[q]);
                a1[q] = fma(-f1, z, a1[q]);
            }
        }
        // (the next (A) touches registers and vbuf only; vbuf was last read before barrier 2)
        CAR_SUB(5);
    }

    __threadfence_block();
    __syncthreads();
    // reflector vectors: global scratch -> LDS rows for phases 2 and 3
    for (int t = tid; t < m * NS + CAR_PAD; t += CAR_T) A[t] = (t < m * NS) ? vws[t] : 0.0;
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
    double* pscal = lds + 4 * NP;          // [2][4]    (alpha, piv, 1/Phi[piv,0]) of the current / next step
    for (int r = tid; r < 2 * NP; r += CAR_T) colbuf[r] = 0.0;
    __syncthreads();
    for (int r = tid; r < N; r += CAR_T) mubuf[r] = mu_in[r];
    if (okcol && col == 0) {
#pragma unroll
        for (int k = 0; k < CAR_RP; ++k)
            if (g + 8 * k < N) colbuf[g + 8 * k] = phi[k];
    }
    __syncthreads();

#define CAR_AMIN_ROW(CTRL)                                                                 \
        {                                                                                 \
            const double ob_ = dpp<CTRL>(best_);                                          \
            const int op_ = __builtin_amdgcn_update_dpp(0, piv_, CTRL, 0xf, 0xf, false);  \
            amin_take(best_, piv_, ob_, op_);                                             \
        }
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
        CAR_AMIN_ROW(ROR8) CAR_AMIN_ROW(ROR4) CAR_AMIN_ROW(ROR2) CAR_AMIN_ROW(ROR1)       \
        {                                                                                 \
            double b0_ = rdlane(best_, 0); int p0_ = __builtin_amdgcn_readlane(piv_, 0);  \
            amin_take(b0_, p0_, rdlane(best_, 16), __builtin_amdgcn_readlane(piv_, 16));  \
            amin_take(b0_, p0_, rdlane(best_, 32), __builtin_amdgcn_readlane(piv_, 32));  \
            amin_take(b0_, p0_, rdlane(best_, 48), __builtin_amdgcn_readlane(piv_, 48));  \
            best_ = b0_; piv_ = p0_;                                                      \
        }                                                                                 \
        if (lane == 0) { (outp)[0] = best_; (outp)[1] = (double)piv_;                     \
                         (outp)[2] = 1.0 / (cp)[max(piv_, 0)]; }                          \
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
        const double alpha = pscal[cur * 4];
        const int piv = (int)pscal[cur * 4 + 1];
        const double rpp = pscal[cur * 4 + 2];
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
            const double qv = prow * rpp;                               // Phi[idx,c] / Phi[idx,0]
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
                CAR_RATIO_TEST(cn, pscal + (cur ^ 1) * 4);
            }
        }
        __syncthreads();
    }
#undef CAR_RATIO_TEST

    CAR_STAMP();
#ifdef CAR_STAMPS
    if (tid == 0 && phi_out != nullptr) for (int q = 0; q < 8; ++q) ((unsigned long long*)phi_out)[q] = st_[q];
#endif
#ifdef CAR_STAMPS2
    if (tid == 0 && phi_out != nullptr) for (int q = 0; q < 6; ++q) ((unsigned long long*)phi_out)[16 + q] = acc_[q];
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

extern "C" int64_t sober_car_ws_bytes(int N, int m) {
    (void)N;
    return (int64_t)m * 8 * sober::CAR_RP * (int64_t)sizeof(double);
}

extern "C" int sober_car_device(const double* X, int ldx, int N, int m, const double* mu_in,
                                int32_t* keep_rank, double* w_star, int32_t* n_keep, double* mu_out,
                                double* phi_out, void* ws, int64_t ws_bytes, void* stream) {
    if (!X || !mu_in || !keep_rank || !w_star || !n_keep || !mu_out || !ws || ldx < m - 1) return SOBER_E_ARG;
    if (!sober_car_supported(N, m)) return SOBER_E_DIM;
    if (ws_bytes < sober_car_ws_bytes(N, m)) return SOBER_E_WS;
    size_t base = (size_t)m * (8 * sober::CAR_RP) + sober::CAR_PAD;
    if (base < (size_t)sober::CAR_P1) base = sober::CAR_P1;
    size_t doubles = base + (size_t)m + 8;
    if (doubles < 4 * 264 + 8) doubles = 4 * 264 + 8;
    const size_t bytes = doubles * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void*)sober::k_car, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(sober::k_car, dim3(1), dim3(sober::CAR_T), bytes, (hipStream_t)stream, X, ldx, N, m,
                       mu_in, keep_rank, w_star, n_keep, mu_out, phi_out, (double*)ws);
    LAUNCH_CHECK();
    return 0;
}
