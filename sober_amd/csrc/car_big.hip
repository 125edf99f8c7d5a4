// Tchernychova_Lyons_CAR (SOBER/_rchq.py:224-270) for ANY batch up to 1023 (N <= 2048 points, m < N test functions): the
// route behind csrc/car.hip (one compute unit, batch <= 100) and csrc/car_mc.hip (nine compute units with the matrix in
// registers, batch <= 224).  Before round 6 such a step went to the host: LAPACK's full SVD of [1 | X]^T (16 ms at batch 250,
// 90 ms at batch 512 on eight cores) and the C++ pivot loop, per level.
//
// Same mathematics as the two register-resident routes -- the null-space basis LAPACK's gesdd returns in Vh[m:, :] is
// Phi = G(0) ... G(m-1) [0; I], G(i) the right Householder reflectors of the Golub-Kahan bidiagonalisation of the m x N matrix
// A = [1 | X]^T (dgebd2 / dlarfg conventions), followed by the N - m pivots of :237-266 -- with the matrix in MEMORY (L2: 8 MB at
// batch 512) and one launch per dependency instead of waits between workgroups (nothing here can give up, so this is
// also the SOBER_CAR_SAFE route of the multi-CU sizes):
//
//   k_big_init    A = [1 | X]^T (m x ld, row-major), flags.
//   k_big_right   step i, a workgroup per FOUR ROWS below row i: G(i) from row i (every workgroup, redundantly: a reduction over
//                 <= 2048 entries), w = A v and A -= tau w v^T for its own rows -- the right reflector never needs another
//                 workgroup's rows.  Workgroup 0 files v (zero-padded) and tau.
//   k_big_left    step i, a workgroup per SIXTEEN COLUMNS right of column i: H(i) from column i (redundantly), z = u^T A and
//                 A -= tau u z^T for its own columns.  The two launches of a step are its two global dependencies
//                 (row i must be complete before G(i), column i before H(i)): 2 m - 2 launches in all.
//   k_big_phi     Phi = P [0; I]: a wave per null vector, the vector in registers, the m reflectors applied backwards.
//   k_big_pivot   the pivots in PANELS of eight null vectors, a workgroup (256 threads, a thread <-> up to eight points) per eight columns of
//                 Phi: every workgroup applies the previous panel's eight eliminations to its columns (the pivot rows'
//                 entries go round the workgroup through LDS, one barrier per pivot); workgroup 0, whose columns ARE the next
//                 panel, then runs that panel's eight ratio tests and in-panel eliminations and files columns, pivot indices
//                 and reciprocals for the next launch: (N - m) / 8 launches.  The early exit of :241-242 (no positive entry)
//                 sets a stop word every later launch honours.
//   k_big_finish  keep_rank / w_star / n_keep / mu_out (:268-269).
//
// Reductions are fixed trees: run-to-run bit-equal.  Elimination multipliers are formed as Phi[idx, c] * (1 / Phi[idx, 0]) like
// the other device routes (the reference divides the product; last-bit differences only, tests/test_car_algorithm.py).
#include "common.hpp"

namespace sober {
namespace big {

constexpr int MAXN = 2048;
constexpr int PB = 8;                 // null vectors per pivot panel
constexpr int PT = 256;               // threads of a pivot workgroup: ONE wave per SIMD (sixteen waves spent 3.4 us per pivot
                                      // on instruction issue alone: every wave runs the reductions and the barriers' code)
constexpr int FT = 1024;              // threads of the finishing workgroup

struct PanelRec {                     // what a panel leaves for the next launch
    int piv[PB];
    int nvalid;                       // eliminations done in the panel (< its width: the loop of :237 ended there)
    int pad[7];
    double rpp[PB];                   // 1 / Phi[idx, 0]
};
struct Flags {
    int stop;                         // the pivot loop has ended (:241-242)
    int pad[15];
};

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// the same total in every lane without a trip through the LDS crossbar per stage (DPP row rotations, then the gfx950 lane
// swaps: csrc/car_mc.hip's reduction) -- for the one place where the reduction IS the step: k_big_phi
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wsum_dpp(double v) {
    v += dpp_mov<0x128>(v);                                   // row_ror 8, 4, 2, 1: every lane of a 16-lane row holds the row's sum
    v += dpp_mov<0x124>(v);
    v += dpp_mov<0x122>(v);
    v += dpp_mov<0x121>(v);
    int lo = __double2loint(v), hi = __double2hiint(v);
    auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    v = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
    lo = __double2loint(v);
    hi = __double2hiint(v);
    auto c = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto d = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double(d[0], c[0]) + __hiloint2double(d[1], c[1]);
}

// dlarfg on (alpha, |x|^2): beta, tau and the scale of x (tau = 0, scale = 0: H = I)
__device__ __forceinline__ void larfg(double alpha, double ss, double& tau, double& scale) {
    if (ss == 0.0) { tau = 0.0; scale = 0.0; return; }
    const double nrm = sqrt(fma(alpha, alpha, ss));
    const double beta = alpha >= 0.0 ? -nrm : nrm;
    tau = (beta - alpha) / beta;
    scale = 1.0 / (alpha - beta);
}

__global__ __launch_bounds__(256) void k_big_init(const double* __restrict__ X, int ldx, int N, int m, int ld, double* __restrict__ A,
                                                  Flags* __restrict__ flags, PanelRec* __restrict__ rec) {
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int k = ty; k < 32; k += 8) {                      // tile[point][function]: X rows are points
        const int c = c0 + k, r = r0 + tx;
        tile[k][tx] = (c < N && r < m) ? (r == 0 ? 1.0 : X[(size_t)c * ldx + (r - 1)]) : 0.0;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int r = r0 + k, c = c0 + tx;
        if (r < m && c < ld) A[(size_t)r * ld + c] = tile[tx][k];
    }
    if (blockIdx.x == 0 && blockIdx.y == 0) {
        if (threadIdx.x == 0) flags->stop = 0;
        if (threadIdx.x < 2) rec[threadIdx.x].nvalid = 0;
    }
}

// NQ: 64-column slots of a row right of column i the launch was sized for (ceil((N - i) / 64) <= NQ)
template <int NQ>
__global__ __launch_bounds__(256) void k_big_right(double* __restrict__ A, double* __restrict__ Vg, double* __restrict__ taug,
                                                   int m, int N, int ld, int i) {
    __shared__ double s_v[MAXN];
    __shared__ double s_red[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const double* row = A + (size_t)i * ld;
    // this wave's own row, asked for before anything depends on it (one memory round trip per launch instead of two)
    const int r = i + 1 + blockIdx.x * 4 + w;
    double* ar = A + (size_t)(r < m ? r : i) * ld;
    double a[NQ];
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
        const int c = i + lane + 64 * k;
        a[k] = c < N ? ar[c] : 0.0;
    }
    double ss = 0.0;
    for (int c = i + 1 + threadIdx.x; c < N; c += 256) {
        const double x = row[c];
        s_v[c] = x;
        ss = fma(x, x, ss);
    }
    ss = wsum(ss);
    if (lane == 0) s_red[w] = ss;
    __syncthreads();
    ss = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    double tau, scale;
    larfg(row[i], ss, tau, scale);
    for (int c = i + 1 + threadIdx.x; c < N; c += 256) s_v[c] *= scale;
    if (threadIdx.x == 0) s_v[i] = 1.0;
    __syncthreads();
    if (blockIdx.x == 0) {
        double* vr = Vg + (size_t)i * ld;
        for (int c = threadIdx.x; c < ld; c += 256) vr[c] = (c < i || c >= N) ? 0.0 : s_v[c];
        if (threadIdx.x == 0) taug[i] = tau;
    }
    if (r >= m || tau == 0.0) return;
    double v[NQ];
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
        const int c = i + lane + 64 * k;
        v[k] = c < N ? s_v[c] : 0.0;
        acc = fma(a[k], v[k], acc);
    }
    const double tw = tau * wsum(acc);
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
        const int c = i + lane + 64 * k;
        if (c < N) ar[c] = fma(-tw, v[k], a[k]);
    }
}

// a workgroup per EIGHT columns (thread = column x 32 row groups); RT: rows per thread the launch was sized for
// (ceil((m - i - 1) / 32) <= RT), in registers from the first instruction on
template <int RT>
__global__ __launch_bounds__(256) void k_big_left(double* __restrict__ A, int m, int N, int ld, int i) {
    __shared__ double s_u[MAXN];
    __shared__ double s_red[4];
    __shared__ double s_p[32][9];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int L = m - i - 1;                                 // rows i+1 .. m-1 (u_0 = 1)
    const int cx = threadIdx.x & 7, ry = threadIdx.x >> 3;
    const int c = i + 1 + blockIdx.x * 8 + cx;
    double* ac = A + (size_t)(i + 1) * ld + (c < N ? c : i + 1);
    double a[RT];
#pragma unroll
    for (int q = 0; q < RT; ++q) {
        const int k = ry + 32 * q;
        a[q] = (k < L && c < N) ? ac[(size_t)k * ld] : 0.0;
    }
    const double* col = A + (size_t)(i + 1) * ld + i;
    double ss = 0.0;
    for (int k = 1 + threadIdx.x; k < L; k += 256) {
        const double x = col[(size_t)k * ld];
        s_u[k] = x;
        ss = fma(x, x, ss);
    }
    ss = wsum(ss);
    if (lane == 0) s_red[w] = ss;
    __syncthreads();
    ss = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    double tau, scale;
    larfg(col[0], ss, tau, scale);
    if (tau == 0.0) return;
    for (int k = 1 + threadIdx.x; k < L; k += 256) s_u[k] *= scale;
    if (threadIdx.x == 0) s_u[0] = 1.0;
    __syncthreads();
    double u[RT];
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < RT; ++q) {
        const int k = ry + 32 * q;
        u[q] = k < L ? s_u[k] : 0.0;
        acc = fma(u[q], a[q], acc);
    }
    s_p[ry][cx] = acc;
    __syncthreads();
    double z = 0.0;
#pragma unroll
    for (int q = 0; q < 32; ++q) z += s_p[q][cx];
    const double tz = tau * z;
    if (c < N) {
#pragma unroll
        for (int q = 0; q < RT; ++q) {
            const int k = ry + 32 * q;
            if (k < L) ac[(size_t)k * ld] = fma(-tz, u[q], a[q]);
        }
    }
}

// Phi^T (K x ld): null vector j in a wave's registers (lane <-> row mod 64, NQ slots), reflectors m-1 .. 0.  The reflectors
// come through LDS in chunks of CH rows (32 KB), the next chunk on its way from memory while this one is applied: a wave
// that fetched each reflector itself spent 3.7 us per step waiting for it (1.9 ms at batch 512).
template <int NQ>
__global__ __launch_bounds__(256) void k_big_phi(const double* __restrict__ Vg, const double* __restrict__ taug, int m, int N, int ld,
                                                 int K, double* __restrict__ PhiT) {
    constexpr int CH = 64 / NQ;                              // reflector rows per chunk: CH * ld <= 4096 doubles
    constexpr int PER = 16;                                  // doubles a thread moves per chunk
    __shared__ double s_v[2][4096];
    __shared__ double s_tau[2][CH];
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    double p[NQ];
#pragma unroll
    for (int k = 0; k < NQ; ++k) p[k] = (lane + 64 * k == m + j) ? 1.0 : 0.0;
    const int n_chunks = (m + CH - 1) / CH;
    double pre[PER];
    double pre_tau = 0.0;
    // chunk c holds reflectors hi .. lo (hi = m - 1 - c CH), stored row lo first
    auto fetch = [&](int c) {
        const int hi = m - 1 - c * CH, lo = hi - CH + 1 > 0 ? hi - CH + 1 : 0;
        const int64_t base = (int64_t)lo * ld, n = (int64_t)(hi - lo + 1) * ld;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int64_t e = threadIdx.x + 256 * q;
            pre[q] = e < n ? Vg[base + e] : 0.0;
        }
        if (threadIdx.x < CH) pre_tau = lo + (int)threadIdx.x <= hi ? taug[lo + threadIdx.x] : 0.0;
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int q = 0; q < PER; ++q) s_v[buf][threadIdx.x + 256 * q] = pre[q];
        if (threadIdx.x < CH) s_tau[buf][threadIdx.x] = pre_tau;
    };
    fetch(0);
    stash(0);
    __syncthreads();
    for (int c = 0; c < n_chunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < n_chunks) fetch(c + 1);
        const int hi = m - 1 - c * CH, lo = hi - CH + 1 > 0 ? hi - CH + 1 : 0;
        if (j < K) {
            for (int i = hi; i >= lo; --i) {
                const double* v = &s_v[buf][(i - lo) * ld + lane];
                const int k0 = i >> 6;                       // slots below hold zeros of v
                double vv[NQ];
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < NQ; ++k) {
                    vv[k] = 0.0;
                    if (k >= k0 && 64 * k < ld) { vv[k] = v[64 * k]; acc = fma(vv[k], p[k], acc); }
                }
                const double ty = s_tau[buf][i - lo] * wsum_dpp(acc);
#pragma unroll
                for (int k = 0; k < NQ; ++k)
                    if (k >= k0) p[k] = fma(-ty, vv[k], p[k]);
            }
        }
        if (c + 1 < n_chunks) stash(buf ^ 1);
        __syncthreads();
    }
    if (j >= K) return;
    double* out = PhiT + (size_t)j * ld + lane;
#pragma unroll
    for (int k = 0; k < NQ; ++k)
        if (64 * k < ld) out[64 * k] = p[k];
}

__global__ __launch_bounds__(256) void k_big_phi_out(const double* __restrict__ PhiT, int N, int K, int ld, double* __restrict__ phi_out) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (int64_t)N * K) return;
    const int r = (int)(t / K), j = (int)(t - (int64_t)r * K);
    phi_out[t] = PhiT[(size_t)j * ld + r];
}

// (value, index) minimum with torch.argmin's rules: the first minimum; a NaN counts as smaller than everything
// (NONE: no candidate; a candidate always beats none)
constexpr int NONE = 0x7fffffff;
__device__ __forceinline__ bool better(double a, int r, double b, int q) {
    return r != NONE && (q == NONE || a < b || (a == b && r < q));
}

template <int RPT>
__global__ __launch_bounds__(PT) void k_big_pivot(double* __restrict__ PhiT, int ld, int N, int K, int j0, const PanelRec* __restrict__ prev,
                                                  PanelRec* __restrict__ cur, const double* __restrict__ mu_in, double* __restrict__ mu,
                                                  Flags* __restrict__ flags) {
    __shared__ double s_pr[PB][PB];
    __shared__ double s_ba[PT / 64];
    __shared__ int s_br[PT / 64];
    __shared__ double s_pp, s_mu;
    __shared__ int s_opiv[PB];
    __shared__ double s_orpp[PB];
    if (flags->stop) return;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int jb = j0 + blockIdx.x * PB;                     // this workgroup's columns jb .. jb + nc - 1
    const int nc = min(PB, K - jb);
    double phi[RPT][PB];
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        const int r = t + PT * q;
#pragma unroll
        for (int c = 0; c < PB; ++c) phi[q][c] = (r < N && c < nc) ? PhiT[(size_t)(jb + c) * ld + r] : 0.0;
    }
    // ---- the previous panel's eliminations on these columns (its record and its columns asked for at once: a memory round
    //      trip per pivot inside the loop cost 1.5 us each)
    if (j0 > 0) {
        __shared__ int s_piv[PB];
        __shared__ double s_rpp[PB];
        __shared__ int s_nv;
        if (t < PB) { s_piv[t] = prev->piv[t]; s_rpp[t] = prev->rpp[t]; }
        if (t == 0) s_nv = prev->nvalid;
        double cs[RPT][PB];
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int r = t + PT * q;
#pragma unroll
            for (int sx = 0; sx < PB; ++sx) cs[q][sx] = r < N ? PhiT[(size_t)(j0 - PB + sx) * ld + r] : 0.0;
        }
        __syncthreads();
        const int nv = s_nv;
#pragma unroll
        for (int sx = 0; sx < PB; ++sx) {
            if (sx < nv) {
                const int piv = s_piv[sx];
                const double rpp = s_rpp[sx];
#pragma unroll
                for (int q = 0; q < RPT; ++q)
                    if (t + PT * q == piv) {
#pragma unroll
                        for (int c = 0; c < PB; ++c) s_pr[sx][c] = phi[q][c];
                    }
                __syncthreads();
#pragma unroll
                for (int q = 0; q < RPT; ++q) {
                    const int r = t + PT * q;
#pragma unroll
                    for (int c = 0; c < PB; ++c) {
                        const double f = s_pr[sx][c] * rpp;
                        phi[q][c] = (r == piv) ? 0.0 : fma(-cs[q][sx], f, phi[q][c]);
                    }
                }
            }
        }
    }
    if (blockIdx.x != 0) {
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int r = t + PT * q;
#pragma unroll
            for (int c = 0; c < PB; ++c)
                if (r < N && c < nc) PhiT[(size_t)(jb + c) * ld + r] = phi[q][c];
        }
        return;
    }
    // ---- workgroup 0: the panel's own pivots (:237-266)
    __syncthreads();                                         // (s_pr is reused below)
    double mr[RPT];
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        const int r = t + PT * q;
        mr[q] = r < N ? (j0 == 0 ? mu_in[r] : mu[r]) : 0.0;
    }
    int nvalid = 0;
    bool ended = false;
#pragma unroll
    for (int s = 0; s < PB; ++s) {
        if (s < nc && !ended) {
            // plis = Phi[:, 0] > 0; idx = first argmin of mu / Phi[:, 0] over plis
            double ba = __builtin_huge_val();
            int br = NONE;
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                const int r = t + PT * q;
                const double ph = phi[q][s];
                if (r < N && ph > 0.0) {
                    double a = mr[q] / ph;
                    if (a != a) a = -__builtin_huge_val();
                    if (better(a, r, ba, br)) { ba = a; br = r; }
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const double oa = __shfl_xor(ba, o);
                const int orr = __shfl_xor(br, o);
                if (better(oa, orr, ba, br)) { ba = oa; br = orr; }
            }
            if (lane == 0) { s_ba[w] = ba; s_br[w] = br; }
            __syncthreads();
            ba = s_ba[0]; br = s_br[0];
#pragma unroll
            for (int k = 1; k < PT / 64; ++k)
                if (better(s_ba[k], s_br[k], ba, br)) { ba = s_ba[k]; br = s_br[k]; }
            if (br == NONE) {                                // :241-242
                if (t == 0) flags->stop = 1;
                ended = true;
            } else {
                const int piv = br;
#pragma unroll
                for (int q = 0; q < RPT; ++q)
                    if (t + PT * q == piv) {
                        s_pp = phi[q][s];
                        s_mu = mr[q];
#pragma unroll
                        for (int c = 0; c < PB; ++c) s_pr[s][c] = phi[q][c];
                    }
                __syncthreads();
                const double pp = s_pp;
                const double alpha = s_mu / pp;              // = alpha[idx] (:246), a NaN included
                const double rpp = 1.0 / pp;
#pragma unroll
                for (int q = 0; q < RPT; ++q) {
                    const int r = t + PT * q;
                    const double cs = phi[q][s];
                    mr[q] = (r == piv) ? 0.0 : __dsub_rn(mr[q], __dmul_rn(alpha, cs));     // :253-254 (product, then difference)
#pragma unroll
                    for (int c = 0; c < PB; ++c)
                        if (c > s) phi[q][c] = (r == piv) ? 0.0 : fma(-cs, s_pr[s][c] * rpp, phi[q][c]);      // :260-266
                }
                if (t == 0) { s_opiv[s] = piv; s_orpp[s] = rpp; }
                nvalid = s + 1;
            }
        }
    }
    // (memory is written once, here: a store in front of a barrier makes the barrier wait for its acknowledgement, 1.5 us per
    //  pivot when the pivot columns were filed as they were found)
    __syncthreads();
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
        const int r = t + PT * q;
        if (r < N) {
            mu[r] = mr[q];
            // the pivot columns as the other workgroups will apply them: column s has not changed since its own step
#pragma unroll
            for (int c = 0; c < PB; ++c)
                if (c < nvalid) PhiT[(size_t)(jb + c) * ld + r] = phi[q][c];
        }
    }
    if (t < PB && t < nvalid) { cur->piv[t] = s_opiv[t]; cur->rpp[t] = s_orpp[t]; }
    if (t == 0) cur->nvalid = nvalid;
}

// w_star = mu[mu > 0], idx_star = arange(N)[mu > 0] (:268-269) as ranks
__global__ __launch_bounds__(FT) void k_big_finish(const double* __restrict__ mu, int N, int32_t* __restrict__ keep_rank,
                                                   double* __restrict__ w_star, int32_t* __restrict__ n_keep, double* __restrict__ mu_out) {
    __shared__ int s_cnt[FT / 64];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    int base = 0;
    for (int r0 = 0; r0 < N; r0 += FT) {
        const int r = r0 + t;
        const double v = r < N ? mu[r] : 0.0;
        const bool keep = r < N && v > 0.0;
        const unsigned long long bal = __ballot(keep);
        __syncthreads();
        if (lane == 0) s_cnt[w] = __popcll(bal);
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int k = 0; k < FT / 64; ++k) {
            if (k < w) before += s_cnt[k];
            total += s_cnt[k];
        }
        const int rank = base + before + __popcll(bal & ((1ull << lane) - 1ull));
        if (r < N) {
            keep_rank[r] = keep ? rank : -1;
            mu_out[r] = v;
            if (keep) w_star[rank] = v;
        }
        base += total;
    }
    if (t == 0) *n_keep = base;
}

}  // namespace big
}  // namespace sober

extern "C" int sober_car_big_supported(int N, int m) { return (m >= 2 && N > m && N <= sober::big::MAXN) ? 1 : 0; }

// scratch: A (m x ld) | reflectors (m x ld) | tau (m, padded) | Phi^T ((N - m) x ld) | mu (ld) | flags | two panel records
// (a workspace sized for (N, m) also serves every (N' <= N, m): the final direct level)
extern "C" int64_t sober_car_big_ws_bytes(int N, int m) {
    if (!sober_car_big_supported(N, m)) return SOBER_E_DIM;
    const int64_t ld = (N + 63) / 64 * 64, mp = (m + 63) / 64 * 64;
    return (2 * (int64_t)m * ld + mp + (int64_t)(N - m) * ld + ld) * (int64_t)sizeof(double) + (int64_t)sizeof(sober::big::Flags)
           + 2 * (int64_t)sizeof(sober::big::PanelRec) + 256;
}

extern "C" int sober_car_big_device(const double* X, int ldx, int N, int m, const double* mu_in, int32_t* keep_rank, double* w_star,
                                    int32_t* n_keep, double* mu_out, double* phi_out, void* ws, int64_t ws_bytes, void* stream) {
    using namespace sober::big;
    if (!X || !mu_in || !keep_rank || !w_star || !n_keep || !mu_out || !ws || ldx < m - 1) return SOBER_E_ARG;
    if (!sober_car_big_supported(N, m)) return SOBER_E_DIM;
    if (ws_bytes < sober_car_big_ws_bytes(N, m)) return SOBER_E_WS;
    hipStream_t st = (hipStream_t)stream;
    const int ld = (N + 63) / 64 * 64, mp = (m + 63) / 64 * 64, K = N - m;
    double* A = (double*)ws;
    double* Vg = A + (size_t)m * ld;
    double* taug = Vg + (size_t)m * ld;
    double* PhiT = taug + mp;
    double* mu = PhiT + (size_t)K * ld;
    Flags* flags = (Flags*)(mu + ld);
    PanelRec* rec = (PanelRec*)(flags + 1);
    hipLaunchKernelGGL(k_big_init, dim3((ld + 31) / 32, (m + 31) / 32), dim3(256), 0, st, X, ldx, N, m, ld, A, flags, rec);
    LAUNCH_CHECK();
    for (int i = 0; i < m; ++i) {
        const int rows = m - i - 1;
        const dim3 gr(rows > 0 ? (rows + 3) / 4 : 1);
        const int nqr = (N - i + 63) / 64;
        if (nqr <= 4) hipLaunchKernelGGL(k_big_right<4>, gr, dim3(256), 0, st, A, Vg, taug, m, N, ld, i);
        else if (nqr <= 8) hipLaunchKernelGGL(k_big_right<8>, gr, dim3(256), 0, st, A, Vg, taug, m, N, ld, i);
        else if (nqr <= 16) hipLaunchKernelGGL(k_big_right<16>, gr, dim3(256), 0, st, A, Vg, taug, m, N, ld, i);
        else hipLaunchKernelGGL(k_big_right<32>, gr, dim3(256), 0, st, A, Vg, taug, m, N, ld, i);
        if (i <= m - 3) {
            const dim3 gl((N - i - 1 + 7) / 8);
            const int rt = (rows + 31) / 32;
            if (rt <= 4) hipLaunchKernelGGL(k_big_left<4>, gl, dim3(256), 0, st, A, m, N, ld, i);
            else if (rt <= 8) hipLaunchKernelGGL(k_big_left<8>, gl, dim3(256), 0, st, A, m, N, ld, i);
            else if (rt <= 16) hipLaunchKernelGGL(k_big_left<16>, gl, dim3(256), 0, st, A, m, N, ld, i);
            else if (rt <= 32) hipLaunchKernelGGL(k_big_left<32>, gl, dim3(256), 0, st, A, m, N, ld, i);
            else hipLaunchKernelGGL(k_big_left<64>, gl, dim3(256), 0, st, A, m, N, ld, i);
        }
    }
    LAUNCH_CHECK();
    const int nq = ld / 64;
    if (nq <= 8) hipLaunchKernelGGL(k_big_phi<8>, dim3((K + 3) / 4), dim3(256), 0, st, Vg, taug, m, N, ld, K, PhiT);
    else if (nq <= 16) hipLaunchKernelGGL(k_big_phi<16>, dim3((K + 3) / 4), dim3(256), 0, st, Vg, taug, m, N, ld, K, PhiT);
    else hipLaunchKernelGGL(k_big_phi<32>, dim3((K + 3) / 4), dim3(256), 0, st, Vg, taug, m, N, ld, K, PhiT);
    LAUNCH_CHECK();
    if (phi_out != nullptr) {
        hipLaunchKernelGGL(k_big_phi_out, dim3((unsigned)(((int64_t)N * K + 255) / 256)), dim3(256), 0, st, PhiT, N, K, ld, phi_out);
        LAUNCH_CHECK();
    }
    int p = 0;
    for (int j0 = 0; j0 < K; j0 += PB, ++p) {
        const int groups = (K - j0 + PB - 1) / PB;
        PanelRec *pr = rec + ((p + 1) & 1), *cr = rec + (p & 1);
        if (N <= 2 * PT) hipLaunchKernelGGL(k_big_pivot<2>, dim3(groups), dim3(PT), 0, st, PhiT, ld, N, K, j0, pr, cr, mu_in, mu, flags);
        else if (N <= 4 * PT) hipLaunchKernelGGL(k_big_pivot<4>, dim3(groups), dim3(PT), 0, st, PhiT, ld, N, K, j0, pr, cr, mu_in, mu, flags);
        else hipLaunchKernelGGL(k_big_pivot<8>, dim3(groups), dim3(PT), 0, st, PhiT, ld, N, K, j0, pr, cr, mu_in, mu, flags);
    }
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_big_finish, dim3(1), dim3(FT), 0, st, mu, N, keep_rank, w_star, n_keep, mu_out);
    LAUNCH_CHECK();
    return 0;
}
