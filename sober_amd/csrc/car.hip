// K5 + K6 on the device: Tchernychova_Lyons_CAR (SOBER/_rchq.py:224-270), everything on chip.
//
// The reference takes the null space of A = [1 | X]^T (m x N, m < N <= 2m) from LAPACK's full SVD:
// Phi = Vh[m:, :].T.  Which basis of the null space comes back decides which points survive
// (SURVEY.md App. C), so "any orthonormal basis" is not good enough.  MKL's gesdd (the LAPACK
// behind torch here) produces Vh[m:, :] = rows m..N-1 of P^T, where P = G(0) G(1) ... G(m-1) is
// the product of the RIGHT Householder reflectors of the Golub-Kahan bidiagonalisation of A
// (dgebrd, lower-bidiagonal case m < N, dlarfg sign convention) -- checked against
// torch.linalg.svd on the reference's own per-level inputs in tests/test_car_algorithm.py (CPU) and
// tests/test_hip_parity.py (GPU).  That product is a deterministic function of A, so it can be
// recomputed here.  Three launches on one stream, each shaped by what bounds its phase:
//
//   k_car_bidiag  1 workgroup x 256 threads.  The m bidiagonalisation steps are a chain of workgroup-wide
//                 reductions (two norms, two matrix-vector products per step): latency, not arithmetic.
//                 Four waves -- one per SIMD -- keep the barriers and the LDS exchanges cheap; the matrix
//                 lives in VGPRs (7 x 13 doubles per thread), a matrix row inside ONE 16-lane DPP row so
//                 that row dot products never leave the wave.  Reflectors v_i, tau_i -> global scratch.
//   k_car_phi     Phi = P [0; I] by backward accumulation.  The columns of Phi are independent, so this
//                 phase is spread over ceil((N-m)/8) workgroups of one wave (8 columns each, 16 lanes x 13
//                 rows per column): ~10 us instead of >100 us inside a single workgroup.
//   k_car_pivot   1 workgroup x 512 threads, Phi in VGPRs (4 columns x 13 rows per thread): the N-m
//                 pivots of :237-266, one barrier per pivot.  Every DPP row carries its own copy of the
//                 weights, so the DPP row that owns the next pivot column runs the ratio test on its
//                 registers (16-lane DPP argmin with first-index tie break) straight after its own
//                 elimination step.
//
// Limits of these one-CU kernels: N <= 208, m <= 112, N - m <= 112 (batch <= 100).  sober_car_device hands larger
// steps (N <= 448, m <= 256: batch <= 224) to the multi-CU kernels of car_mc.hip; beyond those the engine takes the
// host LAPACK route (sober_car_pivot_host).
#include "common.hpp"
#include <atomic>
#include <cstdlib>

namespace sober {

constexpr int CAR_MS = 7;           // matrix row slots per thread in the bidiagonalisation: m <= 16 * 7
constexpr int CAR_CQ = 13;          // 16-lane slots along N: N <= 16 * 13
constexpr int CAR_NS = 16 * CAR_CQ; // stride of a reflector / of the LDS columns
constexpr int CAR_PC = 128;         // stride of a Phi row in the global scratch: N - m <= 128
constexpr int CAR_BT = 256;         // threads of k_car_bidiag

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

// 64-bit integer DPP move; lanes the control does not write keep their own value
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long dpp_u64(unsigned long long v) {
    int lo = (int)(unsigned)v, hi = (int)(unsigned)(v >> 32);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
    return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
}
constexpr int BCAST15 = 0x142, BCAST31 = 0x143;   // lane 15 -> next row (rows 1, 3); lane 31 -> rows 2, 3

// order-preserving map double -> uint64 for the ratio test: NaN -> 0 (torch.argmin lets a NaN win),
// -inf < ... < -0 < +0 < ... < +inf in unsigned order
__device__ __forceinline__ unsigned long long ratio_key(double x) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    const unsigned long long k = (b >> 63) ? ~b : (b | 0x8000000000000000ull);
    return (x != x) ? 0ull : k;
}

// dlarfg: reflector H = I - tau [1; v][1; v]^T with H [alpha; x] = [beta; 0], ss = |x|^2, v = x * scal.
// v_rsq_f64 / v_rcp_f64 seeds + Newton steps (<= 1 ulp) instead of the IEEE sqrt and the two IEEE divisions: a third
// of the dependent chain that sits on the critical path of every step, and no branch (a taken branch costs a single
// resident wave 50-80 cycles of instruction fetch); operands far outside the normal range are rescaled with selects
__device__ __forceinline__ double car_rcp(double c) {
    double r = __builtin_amdgcn_rcp(c);
    double e = fma(-c, r, 1.0);
    r = fma(r, e, r);
    e = fma(-c, r, 1.0);
    return fma(r, e, r);
}
__device__ __forceinline__ void larfg(double alpha, double ss, double& beta, double& tau, double& scal) {
    const double n2r = fma(alpha, alpha, ss);
    const bool tiny = n2r < 1e-200, huge = n2r > 1e200;
    const double f = tiny ? 0x1p300 : (huge ? 0x1p-300 : 1.0), fi = tiny ? 0x1p-300 : (huge ? 0x1p300 : 1.0);
    const double al = alpha * f;
    const double n2 = fma(al, al, (ss * f) * f);
    double r = __builtin_amdgcn_rsq(n2);
    double h = 0.5 * r;
    double e = fma(-(n2 * r), h, 0.5);
    r = fma(r, e, r);
    h = 0.5 * r;
    e = fma(-(n2 * r), h, 0.5);
    r = fma(r, e, r);
    double nr = n2 * r;
    nr = fma(fma(-nr, nr, n2), 0.5 * r, nr);
    const double bs = -copysign(nr, al);
    const double ib = car_rcp(bs);
    double t = (bs - al) * ib;
    t = fma(fma(-t, bs, bs - al), ib, t);
    const bool none = ss == 0.0;
    beta = none ? alpha : bs * fi;
    tau = none ? 0.0 : t;
    scal = none ? 0.0 : car_rcp(al - bs) * f;
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


// ---- publishing the reflectors while the bidiagonalisation runs (fused launch, see k_car_bidiag_fused) ----------
// A double travels as one 16-byte granule {lo, tag, hi, tag} written by ONE store; readers poll the granule itself
// with L1-bypassing loads and accept it when both tags carry (epoch, reflector) -- no fence, no flag (the protocol of
// car_mc.hip, where it is measured).  Producer and consumers sit on the same XCD, so the stores are plain ones.
typedef unsigned int car_u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t car_rsrc_t;
constexpr unsigned CARF_ERR = 0, CARF_XCD = 16, CARF_TICKET = 32, CARF_PROGRESS = 48, CARF_DONE = 56;   // byte offsets in the comm block
__host__ __device__ constexpr int64_t carf_bytes(int) { return 64; }
constexpr unsigned CARF_SPIN_LIMIT = 1u << 22;
__device__ __forceinline__ void carf_put(car_rsrc_t rs, unsigned off, double v, unsigned tag) {
    car_u32x4 g;
    g.x = (unsigned)__double2loint(v); g.y = tag; g.z = (unsigned)__double2hiint(v); g.w = tag;
    __builtin_amdgcn_raw_buffer_store_b128(g, rs, off, 0, 0);
}
__device__ __forceinline__ car_u32x4 carf_load(car_rsrc_t rs, unsigned off) {
    return __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16);                    // sc1: past this CU's L1
}
__device__ __forceinline__ bool carf_ok(const car_u32x4& g, unsigned tag) { return (g.y == tag) & (g.w == tag); }
__device__ __forceinline__ double carf_val(const car_u32x4& g) { return __hiloint2double((int)g.z, (int)g.x); }
struct CarPub { car_rsrc_t rs; unsigned tag0; };
// "reflectors 0 .. k-1 are complete in memory": the wave that STORED reflector k-1 says so one step later, when its
// stores have long been acknowledged (the wait below is then free) -- nothing is added to the chain of the step itself
__device__ __forceinline__ void car_publish_progress(const CarPub& pub, int k) {
    const int tid = threadIdx.x;
    if (k > 0 && (tid >> 6) == (((k - 1) & 15) >> 2)) {                // (wave-uniform: the wave of DPP row (k-1) & 15)
        __builtin_amdgcn_s_waitcnt(0x0F70);                            // vmcnt(0) only
        if ((tid & 63) == 0) __builtin_amdgcn_raw_buffer_store_b32(pub.tag0 + (unsigned)k, pub.rs, CARF_PROGRESS, 0, 0);
    }
}

// ---------------- phase 1: Golub-Kahan bidiagonalisation (dgebd2, m < N), matrix in VGPRs ----------------
// Thread (R = tid >> 4, C = tid & 15) holds A[R + 16 k][C + 16 q], k < 7, q < 13.
// per step: (A) the owner DPP row builds G(i), publishes v~ (LDS + global)                         | barrier
//           (B) everyone applies it to its rows > i; column i is published                          | barrier
//           (C) every wave builds H(i) from column i for itself; partial column sums u~^T A -> LDS   | barrier
//           (D) 208 threads finish the sums                                                         | barrier
//           then the rank-1 update with H(i)
// Steps 16 S .. 16 S + 15 only touch row slots k >= S and column slots q >= S: the step body is instantiated
// once per S so that the finished part of the matrix costs no instructions (the kernel is bound by the
// instruction count of its one wave per SIMD, ~6 cycles per FP64 VALU op).
struct CarLds {
    double* vbuf; double* colb; double* zsum; double* zpart; double* scal;
};

#ifdef CAR_BSTAMPS      // diagnostic build (`make stamps`): per-segment cycle sums of every wave -> behind the tau block
#define CB_DECL unsigned long long cacc_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ctl_; \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ctl_) :: "memory");
#define CB_STAMP(K) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
    cacc_[K] += t_ - ctl_; ctl_ = t_; } while (0)
#define CB_FLUSH(S_) do { if ((threadIdx.x & 63) == 0) for (int k_ = 0; k_ < 10; ++k_) \
    ((unsigned long long*)(taup + 128 + CAR_NS * CAR_PC))[((threadIdx.x >> 6) * 8 + (S_)) * 10 + k_] = cacc_[k_]; } while (0)
#else
#define CB_DECL
#define CB_STAMP(K) do { } while (0)
#define CB_FLUSH(S_) do { } while (0)
#endif

template <int S, bool FUSED>
__device__ __forceinline__ void car_bidiag_block(double (&a)[CAR_MS][CAR_CQ], int m, const CarLds& L,
                                                 double* __restrict__ vws, double* __restrict__ taup, const CarPub& pub) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int R = tid >> 4, C = tid & 15;
    const int i_end = min(16 * S + 16, m);
    CB_DECL
    for (int i = 16 * S; i < i_end; ++i) {
        CB_STAMP(0);
        const int li = i & 15;                               // row i: DPP row li, slot S; column i: lane li, slot S
        if constexpr (FUSED) car_publish_progress(pub, i);   // reflector i - 1 is complete in memory by now
        if (R == li) {                                                 // (A)
            double ss0 = 0.0, ss1 = 0.0;
            {
                const double x = a[S][S];
                ss0 = (C > li) ? x * x : 0.0;
            }
#pragma unroll
            for (int q = S + 1; q < CAR_CQ; ++q) { if (q & 1) ss1 = fma(a[S][q], a[S][q], ss1); else ss0 = fma(a[S][q], a[S][q], ss0); }
            const double ss = row16_sum(ss0 + ss1);
            // the diagonal entry of the owner row sits in lane 17 li of its wave (R = li, C = li): one broadcast
            // instead of a masked 16-lane sum (+ 0.0: a negative zero reads as the sum read it)
            const double al = rdlane(a[S][S], (17 * li) & 63) + 0.0;
            double beta, tau, sc;
            larfg(al, ss, beta, tau, sc);
#pragma unroll
            for (int q = 0; q < CAR_CQ; ++q) {
                const int c = C + 16 * q;
                double v;
                if (q < S) v = 0.0;
                else if (q == S) v = (C < li) ? 0.0 : ((C == li) ? 1.0 : a[S][S] * sc);
                else v = a[S][q] * sc;
                L.vbuf[c] = v;
                vws[(size_t)i * CAR_NS + c] = v;
            }
            if (C == 0) { taup[i] = tau; L.scal[1] = tau; }
        }
        if (i == m - 1) { CB_FLUSH(S); return; }
        CB_STAMP(1);
        CAR_LDS_BARRIER();
        CB_STAMP(2);
        double tG[CAR_MS], vG[CAR_CQ];
        {                                                              // (B)
            const double tau = L.scal[1];
            double v[CAR_CQ];
#pragma unroll
            for (int q = S; q < CAR_CQ; ++q) v[q] = L.vbuf[C + 16 * q];
            // the row sums first, then column slot S alone -- it holds column i, which H(i) is built from -- and its
            // publication; the rest of the rank-1 update waits until after the barrier, where it shares a basic block
            // with H(i)'s dependent reflector chain (same operations, same results: only the order moved)
#pragma unroll
            for (int k = S; k < CAR_MS; ++k) {
                double w0 = 0.0, w1 = 0.0;
#pragma unroll
                for (int q = S; q < CAR_CQ; ++q) { if (q & 1) w1 = fma(a[k][q], v[q], w1); else w0 = fma(a[k][q], v[q], w0); }
                double t = tau * row16_sum(w0 + w1);
                tG[k] = (R + 16 * k > i) ? t : 0.0;                    // rows <= i stay (rows >= m are zero)
                a[k][S] = fma(-tG[k], v[S], a[k][S]);
            }
            if (C == li) {                                             // publish column i
#pragma unroll
                for (int k = S; k < CAR_MS; ++k) L.colb[R + 16 * k] = a[k][S];
            }
#pragma unroll
            for (int q = S + 1; q < CAR_CQ; ++q) vG[q] = v[q];
        }
        CB_STAMP(3);
        CAR_LDS_BARRIER();
        CB_STAMP(4);
        double u[CAR_MS], tauq;
        {                                                              // (C), redundantly in every wave
            const double x0 = L.colb[lane], x1 = L.colb[lane + 64];    // zero from row m on
            double s2 = ((lane >= i + 2) ? x0 * x0 : 0.0) + ((lane + 64 >= i + 2) ? x1 * x1 : 0.0);
            s2 = wave_sum(s2);
            double beta2, sc2;
            larfg(L.colb[i + 1], s2, beta2, tauq, sc2);
#pragma unroll
            for (int k = S; k < CAR_MS; ++k)                           // (the rest of G(i)'s update)
#pragma unroll
                for (int q = S + 1; q < CAR_CQ; ++q) a[k][q] = fma(-tG[k], vG[q], a[k][q]);
#pragma unroll
            for (int k = S; k < CAR_MS; ++k) {
                const int r = R + 16 * k;
                double cr = L.colb[r];                                 // (r < 128: read unconditionally, then select --
                asm volatile("" : "+v"(cr));                           //  a conditional read is a branch per row slot)
                u[k] = (r <= i) ? 0.0 : ((r == i + 1) ? 1.0 : cr * sc2);
            }
            double zp[CAR_CQ];
#pragma unroll
            for (int q = S; q < CAR_CQ; ++q) zp[q] = u[S] * a[S][q];
#pragma unroll
            for (int k = S + 1; k < CAR_MS; ++k)
#pragma unroll
                for (int q = S; q < CAR_CQ; ++q) zp[q] = fma(u[k], a[k][q], zp[q]);
#pragma unroll
            for (int q = S; q < CAR_CQ; ++q) L.zpart[R * CAR_NS + C + 16 * q] = zp[q];
        }
        CB_STAMP(5);
        CAR_LDS_BARRIER();
        CB_STAMP(6);
        if (tid >= 16 * S && tid < CAR_NS) {                           // (D): columns of the live slots
            double z0 = 0.0, z1 = 0.0;
#pragma unroll
            for (int w = 0; w < 16; w += 2) {
                z0 += L.zpart[w * CAR_NS + tid];
                z1 += L.zpart[(w + 1) * CAR_NS + tid];
            }
            L.zsum[tid] = z0 + z1;
        }
        CB_STAMP(7);
        CAR_LDS_BARRIER();
        CB_STAMP(8);
        {
            double zq[CAR_CQ];
#pragma unroll
            for (int q = S; q < CAR_CQ; ++q) {
                const int c = C + 16 * q;
                zq[q] = (c > i) ? L.zsum[c] : 0.0;                     // H(i) acts on columns i+1 .. N-1 only
            }
#pragma unroll
            for (int k = S; k < CAR_MS; ++k) {
                const double f = tauq * u[k];
#pragma unroll
                for (int q = S; q < CAR_CQ; ++q) a[k][q] = fma(-f, zq[q], a[k][q]);
            }
        }
        // (the next (A) touches registers, vbuf and scal[1] only; both were last read before the 2nd barrier)
        CB_STAMP(9);
    }
    CB_FLUSH(S);
}

template <bool FUSED>
__device__ __forceinline__ void car_bidiag_body(const double* __restrict__ X, int ldx, int N, int m,
                                                double* __restrict__ vws, double* __restrict__ taup, const CarPub& pub) {
    __shared__ double vbuf[CAR_NS];
    __shared__ double colb[128];
    __shared__ double zsum[CAR_NS];
    __shared__ double zpart[16 * CAR_NS];
    __shared__ double scal[4];
    const int tid = threadIdx.x;
    const int R = tid >> 4, C = tid & 15;
    double a[CAR_MS][CAR_CQ];
#pragma unroll
    for (int k = 0; k < CAR_MS; ++k)
#pragma unroll
        for (int q = 0; q < CAR_CQ; ++q) {
            const int r = R + 16 * k, c = C + 16 * q;
            a[k][q] = (c < N && r < m) ? ((r == 0) ? 1.0 : X[(size_t)c * ldx + (r - 1)]) : 0.0;
        }
    if (tid < 128) colb[tid] = 0.0;
    __syncthreads();
    const CarLds L{vbuf, colb, zsum, zpart, scal};
    car_bidiag_block<0, FUSED>(a, m, L, vws, taup, pub);
    if (m > 16) car_bidiag_block<1, FUSED>(a, m, L, vws, taup, pub);
    if (m > 32) car_bidiag_block<2, FUSED>(a, m, L, vws, taup, pub);
    if (m > 48) car_bidiag_block<3, FUSED>(a, m, L, vws, taup, pub);
    if (m > 64) car_bidiag_block<4, FUSED>(a, m, L, vws, taup, pub);
    if (m > 80) car_bidiag_block<5, FUSED>(a, m, L, vws, taup, pub);
    if (m > 96) car_bidiag_block<6, FUSED>(a, m, L, vws, taup, pub);
}

__global__ __launch_bounds__(CAR_BT) void k_car_bidiag(const double* __restrict__ X, int ldx, int N, int m,
                                                       double* __restrict__ vws, double* __restrict__ taup) {
    car_bidiag_body<false>(X, ldx, N, m, vws, taup, CarPub{});
}

// ---------------- phases 1 + 2 in one launch ----------------
// Phi = G(0) ... G(m-1) [0; I] is also the last N - m columns of P = G(0) G(1) ... G(m-1) accumulated FORWARD, and in
// that order every ROW of P is independent: p <- p - tau (p . v) v^T as each reflector appears.  So the 24 us that
// k_car_phi spent after the bidiagonalisation (100 dependent steps per column, on an otherwise idle chip) move BESIDE
// it: workgroup 0 is the bidiagonalisation and publishes every reflector as tagged granules; the other workgroups
// that find themselves on ITS XCD (hardware id; the rest exit at once) take tickets for groups of four rows, one wave
// per row, and follow the reflectors as they appear -- one step behind the producer, finished ~1 us after it.
// Tags carry an epoch (one per launch), so nothing has to be cleared between launches; spins are bounded (error word:
// k_car_pivot then reports n_keep = -1, the engine's signal for the host route).
__global__ __launch_bounds__(CAR_BT) void k_car_bidiag_fused(const double* __restrict__ X, int ldx, int N, int m,
                                                             double* __restrict__ vws, double* __restrict__ taup,
                                                             double* __restrict__ Phi, void* __restrict__ comm,
                                                             unsigned cbytes, unsigned epoch) {
    const car_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(comm, 0, (int)cbytes, 0x00020000);
    const unsigned tag0 = epoch << 7;                                  // (m <= 112 < 128)
    unsigned* words = (unsigned*)comm;
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u;       // HW_REG_XCC_ID
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0) {
            __hip_atomic_store(words + CARF_ERR / 4, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(words + CARF_TICKET / 4, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(words + CARF_DONE / 4, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(words + CARF_XCD / 4, (epoch << 4) | (xcc + 1u), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        const CarPub pub{rs, tag0};
        car_bidiag_body<true>(X, ldx, N, m, vws, taup, pub);
        __syncthreads();
        car_publish_progress(pub, m);                                  // the last reflector (this time the wait is real)
        return;
    }
    // ---- consumers ----
    __shared__ int s_group;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) {
        unsigned spins = 0, w;
        int grp = -1;
        while (((w = __hip_atomic_load(words + CARF_XCD / 4, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) >> 4) != epoch) {
            if (++spins > CARF_SPIN_LIMIT) { w = 0; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        if ((w & 15u) == xcc + 1u) grp = 0;                            // on the producer's XCD: a worker
        s_group = grp;
    }
    __syncthreads();
    if (s_group < 0) return;
    const int NC = N - m;
    // TWO rows per wave, eight per workgroup: the launch's register budget is the producer's (one workgroup per CU), and
    // the 26 row groups must all be resident on the ~31 free CUs of the producer's XCD while it runs
    constexpr int RW = 2;
    constexpr int n_groups = CAR_NS / (4 * RW);
    static_assert(n_groups == CAR_NS / 8, "the pivot kernels check CARF_DONE against CAR_NS / 8");
    for (;;) {
        __syncthreads();
        if (tid == 0)
            s_group = (int)__hip_atomic_fetch_add(words + CARF_TICKET / 4, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const int grp = s_group;
        if (grp >= n_groups) return;
        const int r0 = (4 * grp + wave) * RW;                          // my rows of P: r0, r0 + 1 (wave-uniform)
        double p[RW][4];
#pragma unroll
        for (int w = 0; w < RW; ++w)
#pragma unroll
            for (int q = 0; q < 4; ++q) p[w][q] = (lane + 64 * q == r0 + w) ? 1.0 : 0.0;
        const bool ok3 = lane + 192 < CAR_NS;
        bool failed = false;
        if (r0 < N) {
            const car_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(vws, 0, (int)((size_t)(m * CAR_NS + 128) * sizeof(double)), 0x00020000);
            int have = 0;                                              // reflectors known to be complete
            for (int i = 0; i < m; ++i) {
                unsigned spins = 0;
                while (have <= i) {                                    // (one broadcast load per poll)
                    const unsigned w = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs, CARF_PROGRESS, 0, 16);
                    if ((w >> 7) == epoch) have = (int)(w & 127u);
                    if (have > i) break;
                    if (++spins > CARF_SPIN_LIMIT ||
                        ((spins & 255u) == 0u && __builtin_amdgcn_raw_buffer_load_b32(rs, CARF_ERR, 0, 16) != 0)) {
                        __builtin_amdgcn_raw_buffer_store_b32(1, rs, CARF_ERR, 0, 16);
                        failed = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(4);
                }
                if (failed) break;
                // (sc1 loads: tau's cache line also holds taus that are not written yet -- never through this CU's L1)
                const unsigned base = (unsigned)(i * CAR_NS) * 8u;
                double vv[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const auto g2 = __builtin_amdgcn_raw_buffer_load_b64(rv, base + (unsigned)min(lane + 64 * q, CAR_NS - 1) * 8u, 0, 16);
                    vv[q] = __hiloint2double((int)g2[1], (int)g2[0]);
                }
                const auto gt2 = __builtin_amdgcn_raw_buffer_load_b64(rv, (unsigned)(m * CAR_NS + i) * 8u, 0, 16);
                const double tau = __hiloint2double((int)gt2[1], (int)gt2[0]);
                const double v0 = vv[0], v1 = vv[1], v2 = vv[2], v3 = ok3 ? vv[3] : 0.0;
                double d[RW];
#pragma unroll
                for (int w = 0; w < RW; ++w) d[w] = fma(v0, p[w][0], v1 * p[w][1]) + fma(v2, p[w][2], v3 * p[w][3]);
#pragma unroll
                for (int w = 0; w < RW; ++w) {
                    const double t = tau * wave_sum(d[w]);
                    p[w][0] = fma(-t, v0, p[w][0]); p[w][1] = fma(-t, v1, p[w][1]);
                    p[w][2] = fma(-t, v2, p[w][2]); p[w][3] = fma(-t, v3, p[w][3]);
                }
            }
        }
        // rows of Phi: the last N - m entries of the rows of P, zero padded to CAR_PC columns (rows >= N: zeros)
#pragma unroll
        for (int w = 0; w < RW; ++w) {
            const int r = r0 + w;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = lane + 64 * q;                           // column of P
                const int col = c - m;
                if (c < CAR_NS && col >= 0 && col < CAR_PC) Phi[(size_t)r * CAR_PC + col] = (r < N && col < NC) ? p[w][q] : 0.0;
            }
            // (columns beyond what the lanes above cover)
            for (int col = max(CAR_NS - m, 0) + lane; col < CAR_PC; col += 64) Phi[(size_t)r * CAR_PC + col] = 0.0;
        }
        // this group of rows is in memory: the pivot kernel refuses to run on a Phi with a group missing
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
        if (tid == 0 && !failed)
            __hip_atomic_fetch_add(words + CARF_DONE / 4, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---------------- phase 2: Phi = G(0) ... G(m-1) [0; I]  (N x NC) ----------------
// The columns of Phi are independent: ONE WAVE PER COLUMN (lane l holds rows l + 64 q, q < 4), four waves per
// workgroup, 32 workgroups.  A step is 4 + 4 FMAs and one 64-lane sum; three register buffers rotate through
// the reflectors so that an L2 round trip (longer than a step) is always two steps ahead.
__global__ __launch_bounds__(256) void k_car_phi(const double* __restrict__ vws, const double* __restrict__ taup,
                                                 int N, int m, double* __restrict__ Phi,
                                                 double* __restrict__ phi_out) {
    const int lane = threadIdx.x & 63;
    const int col = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int NC = N - m;
    double phi[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) phi[q] = ((lane + 64 * q) == m + col && col < NC) ? 1.0 : 0.0;
    if (col < NC) {                                                    // wave-uniform
        const int r3 = min(lane + 192, CAR_NS - 1);                    // slot 3 exists for lanes 0..15 only
        const bool ok3 = lane + 192 < CAR_NS;
        // loads are unconditional (index clamped): a conditional load would make the compiler wait for ALL
        // outstanding loads at the next use instead of counting them
#define CAR_LOAD(BUF, TAU, I)                                                              \
    {                                                                                      \
        const double* src_ = vws + (size_t)max((I), 0) * CAR_NS;                           \
        BUF[0] = src_[lane]; BUF[1] = src_[lane + 64]; BUF[2] = src_[lane + 128];          \
        BUF[3] = src_[r3];                                                                 \
        TAU = taup[max((I), 0)];                                                           \
    }
#define CAR_APPLY(V, TAU, I)                                                               \
    if ((I) >= 0) {                                                                        \
        const double v3_ = ok3 ? V[3] : 0.0;                                               \
        const double d_ = fma(V[0], phi[0], V[1] * phi[1]) + fma(V[2], phi[2], v3_ * phi[3]); \
        const double t_ = TAU * wave_sum(d_);                                              \
        phi[0] = fma(-t_, V[0], phi[0]); phi[1] = fma(-t_, V[1], phi[1]);                  \
        phi[2] = fma(-t_, V[2], phi[2]); phi[3] = fma(-t_, v3_, phi[3]);                   \
    }
        double vA[4], vB[4], vC[4];
        double tA = 0.0, tB = 0.0, tC = 0.0;
        CAR_LOAD(vA, tA, m - 1)
        CAR_LOAD(vB, tB, m - 2)
        for (int i = m - 1; i >= 0; i -= 3) {
            CAR_LOAD(vC, tC, i - 2)
            CAR_APPLY(vA, tA, i)
            CAR_LOAD(vA, tA, i - 3)
            CAR_APPLY(vB, tB, i - 1)
            CAR_LOAD(vB, tB, i - 4)
            CAR_APPLY(vC, tC, i - 2)
        }
#undef CAR_LOAD
#undef CAR_APPLY
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = lane + 64 * q;
        if (r < CAR_NS) {
            Phi[(size_t)r * CAR_PC + col] = (r < N) ? phi[q] : 0.0;
            if (phi_out != nullptr && col < NC && r < N) phi_out[(size_t)r * NC + col] = phi[q];
        }
    }
}

// ---------------- phase 3: the pivots of SOBER/_rchq.py:237-266 ----------------
// Wave w owns columns w, w + 16, ...; lane l holds rows l + 64 q (q < 4) of each.  One barrier per pivot.
// A dependent FP64 operation costs ~15 ns on one wave, so the kernel is shaped around the length of the
// dependency chain from "pivot s known" to "pivot s+1 known":
//   * the weights travel from owner to owner through LDS (only the wave that runs the ratio test needs them; a
//     dead row -- Phi[idx, :] = 0 of :266 -- is marked by -0.0); the wave that owns column s+1 has that column in a FIXED
//     register set (its columns are consumed in order; the array is rotated after each ownership), eliminates
//     it first and runs the ratio test of step s+1 (:239-247) on its registers, while its other columns are
//     updated by independent instructions the scheduler interleaves into the same block;
//   * the 4 quotients AND the 4 reciprocals of a lane are issued together (IEEE division, interleaved chains);
//   * quotients become order-preserving 64-bit keys; the 64-lane minimum is taken on the high and then the low
//     word with single-instruction DPP steps, the first index then comes from ballots on the scalar unit;
//   * the slot of the pivot row is picked by selects (no register-indexed branches) and the pivot row is not
//     zeroed in the registers (:266) but marked dead: every later reader of that row goes through the mask.
#ifndef CAR_PW_
#define CAR_PW_ 16
#endif
constexpr int CAR_PW = CAR_PW_;            // waves in k_car_pivot
#ifndef CAR_PJW_
#define CAR_PJW_ 7                        // 8 spills at 1024 threads (128 VGPRs); 7 x 16 = 112 columns cover batch 100
#endif
constexpr int CAR_PJW = CAR_PJW_;          // columns per wave: N - m <= CAR_PW * CAR_PJW

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_min_u32(unsigned v) {
    const unsigned o = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xf, false);
    return o < v ? o : v;
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {        // total in lane 63
    v = dpp_min_u32<ROR8, 0xf>(v);
    v = dpp_min_u32<ROR4, 0xf>(v);
    v = dpp_min_u32<ROR2, 0xf>(v);
    v = dpp_min_u32<ROR1, 0xf>(v);
    v = dpp_min_u32<BCAST15, 0xa>(v);
    v = dpp_min_u32<BCAST31, 0xc>(v);
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

__global__ __launch_bounds__(CAR_PW * 64) void k_car_pivot(const double* __restrict__ Phi, int N, int m,
                                                           const double* __restrict__ mu_in,
                                                           int32_t* __restrict__ keep_rank,
                                                           double* __restrict__ w_star,
                                                           int32_t* __restrict__ n_keep_out,
                                                           double* __restrict__ mu_out,
                                                           const unsigned* __restrict__ err) {
    if (err != nullptr && (err[CARF_ERR / 4] != 0u || err[CARF_DONE / 4] != (unsigned)(CAR_NS / 8))) {
        // the fused launch in front gave up on a reflector (bounded spins) or left a group of Phi's rows out: no result
        if (threadIdx.x == 0) *n_keep_out = -1;
        return;
    }
    __shared__ double colbuf[2 * 256];     // current / next pivot column (zero beyond N)
    __shared__ double pscal[2 * 4];        // (alpha, piv, 1/Phi[piv,0]) of the current / next step
    __shared__ double mubuf[256];          // the weights after the last finished update; -0.0 marks a dead row
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NC = N - m;
    double phi[CAR_PJW][4];                // phi[0] = my next column to become the pivot column
    bool inr[4];
#pragma unroll
    for (int j = 0; j < CAR_PJW; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = lane + 64 * q;
            phi[j][q] = (r < CAR_NS && wave + CAR_PW * j < CAR_PC) ? Phi[(size_t)r * CAR_PC + wave + CAR_PW * j] : 0.0;
        }
#pragma unroll
    for (int q = 0; q < 4; ++q) inr[q] = lane + 64 * q < N;
    if (tid < 512) colbuf[tid] = 0.0;
    __syncthreads();
    int nlive = (NC > wave) ? (NC - wave + CAR_PW - 1) / CAR_PW : 0;   // my columns not yet consumed as pivot columns

    // publish column phi[0] as the pivot column of buffer nb and run the ratio test on it: first argmin of
    // mu/Phi over Phi > 0, a NaN quotient wins (:239-247)
#define CAR_RATIO_TEST(nb)                                                                \
    {                                                                                     \
        double rt_[4], rc_[4];                                                            \
        unsigned kh_[4], kl_[4];                                                          \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                   \
            colbuf[(nb) * 256 + lane + 64 * q] = phi[0][q];       /* zero beyond N */     \
            rt_[q] = mu4[q] / phi[0][q];                                                  \
            rc_[q] = 1.0 / phi[0][q];                                                     \
        }                                                                                 \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                   \
            const unsigned long long k_ = ratio_key(rt_[q]);                              \
            const bool ok_ = inr[q] & (phi[0][q] > 0.0) & !dead[q];                       \
            kh_[q] = ok_ ? (unsigned)(k_ >> 32) : 0xffffffffu;                            \
            kl_[q] = ok_ ? (unsigned)k_ : 0xffffffffu;                                    \
        }                                                                                 \
        const unsigned h01_ = min(kh_[0], kh_[1]), h23_ = min(kh_[2], kh_[3]);            \
        const unsigned H_ = wave_min_u32(min(h01_, h23_));                                \
        const unsigned long long m0_ = __ballot(kh_[0] == H_), m1_ = __ballot(kh_[1] == H_); \
        const unsigned long long m2_ = __ballot(kh_[2] == H_), m3_ = __ballot(kh_[3] == H_); \
        unsigned L_ = 0;                                                                  \
        const bool single_ = (__popcll(m0_) + __popcll(m1_) + __popcll(m2_) + __popcll(m3_)) == 1; \
        if (!single_) {                      /* rare: several quotients share the high word */ \
            unsigned l_[4];                                                               \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) l_[q] = (kh_[q] == H_) ? kl_[q] : 0xffffffffu; \
            L_ = wave_min_u32(min(min(l_[0], l_[1]), min(l_[2], l_[3])));                 \
        }                                                                                 \
        int piv_ = -1;                                                                    \
        double al_ = 0.0, rp_ = 1.0;                                                      \
        if (H_ != 0xffffffffu) {         /* uniform; an all-ones high word = no candidate (or masked) */ \
            const unsigned long long b0_ = single_ ? m0_ : __ballot((kh_[0] == H_) & (kl_[0] == L_)); \
            const unsigned long long b1_ = single_ ? m1_ : __ballot((kh_[1] == H_) & (kl_[1] == L_)); \
            const unsigned long long b2_ = single_ ? m2_ : __ballot((kh_[2] == H_) & (kl_[2] == L_)); \
            const unsigned long long b3_ = single_ ? m3_ : __ballot((kh_[3] == H_) & (kl_[3] == L_)); \
            if (b0_)      { const int f_ = __ffsll((long long)b0_) - 1; piv_ = f_;       al_ = rdlane(rt_[0], f_); rp_ = rdlane(rc_[0], f_); } \
            else if (b1_) { const int f_ = __ffsll((long long)b1_) - 1; piv_ = f_ + 64;  al_ = rdlane(rt_[1], f_); rp_ = rdlane(rc_[1], f_); } \
            else if (b2_) { const int f_ = __ffsll((long long)b2_) - 1; piv_ = f_ + 128; al_ = rdlane(rt_[2], f_); rp_ = rdlane(rc_[2], f_); } \
            else          { const int f_ = __ffsll((long long)b3_) - 1; piv_ = f_ + 192; al_ = rdlane(rt_[3], f_); rp_ = rdlane(rc_[3], f_); } \
        }                                                                                 \
        if (lane == 0) { pscal[(nb) * 4] = al_; pscal[(nb) * 4 + 1] = (double)piv_;       \
                         pscal[(nb) * 4 + 2] = rp_; }                                     \
    }
    // rank-1 elimination of column J: Phi[:,c] -= Phi[:,0] * (Phi[idx,c] / Phi[idx,0])  (:260-266)
#define CAR_ELIM(J)                                                                       \
    {                                                                                     \
        const double lo_ = kp0 ? phi[J][0] : phi[J][1], hi_ = kp2 ? phi[J][2] : phi[J][3]; \
        const double qv_ = rdlane(kplo ? lo_ : hi_, lp) * rpp_;                           \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) phi[J][q] = fma(-qv_, pc[q], phi[J][q]); \
    }
    // eliminate live columns FROM .. nlive-1 (consumed columns are zero; uniform branches)
#define CAR_ELIM_IF(J, FROM) if ((J) >= (FROM) && (J) < CAR_PJW && (J) < nlive) CAR_ELIM((J) < CAR_PJW ? (J) : 0)
#define CAR_ELIM_LIVE(FROM)                                                               \
    CAR_ELIM_IF(0, FROM) CAR_ELIM_IF(1, FROM) CAR_ELIM_IF(2, FROM) CAR_ELIM_IF(3, FROM)   \
    CAR_ELIM_IF(4, FROM) CAR_ELIM_IF(5, FROM) CAR_ELIM_IF(6, FROM) CAR_ELIM_IF(7, FROM)   \
    CAR_ELIM_IF(8, FROM) CAR_ELIM_IF(9, FROM) CAR_ELIM_IF(10, FROM) CAR_ELIM_IF(11, FROM) \
    CAR_ELIM_IF(12, FROM) CAR_ELIM_IF(13, FROM) CAR_ELIM_IF(14, FROM) CAR_ELIM_IF(15, FROM)
    // weights after pivot `PIV` (alpha A, pivot column PCOL) from the weights in mubuf; -0.0 = dead row
#define CAR_MU_STEP(A, PIV, PCOL)                                                         \
    double mu4[4];                                                                        \
    bool dead[4];                                                                         \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                       \
        const double mp_ = mubuf[lane + 64 * q];                                          \
        dead[q] = ((__double2hiint(mp_) == (int)0x80000000) & (__double2loint(mp_) == 0)) | (lane + 64 * q == (PIV)); \
        mu4[q] = dead[q] ? -0.0 : __dsub_rn(mp_, __dmul_rn((A), (PCOL)[q]));              \
    }
    if (wave == 0) {                                                   // column 0 is the first pivot column
        double mu4[4];
        bool dead[4] = {false, false, false, false};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            mu4[q] = inr[q] ? mu_in[lane + 64 * q] + 0.0 : 0.0;         // (+ 0.0: an input -0.0 is not a dead row)
            mubuf[lane + 64 * q] = mu4[q];
        }
        CAR_RATIO_TEST(0)
#pragma unroll
        for (int j = 0; j + 1 < CAR_PJW; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) phi[j][q] = phi[j + 1][q];
#pragma unroll
        for (int q = 0; q < 4; ++q) phi[CAR_PJW - 1][q] = 0.0;
        --nlive;
    }
    __syncthreads();
#ifdef CAR_STAMPS
    unsigned long long acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl_ = __builtin_amdgcn_s_memtime(), arr_ = 0;
#define CAR_SUB(K) do { unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc_[K] += t_ - tl_; tl_ = t_; } while (0)
#else
#define CAR_SUB(K) do { } while (0)
#endif

    int cur = 0, s = 0;
    for (; s < NC; ++s, cur ^= 1) {
#ifdef CAR_STAMPS
        const unsigned long long it0_ = tl_;
#endif
        const double alpha = pscal[cur * 4];
        const int piv = (int)pscal[cur * 4 + 1];
        const double rpp = pscal[cur * 4 + 2];
        double rpp_ = rpp;
        if (piv < 0) break;                                             // Q6 (:241-242), uniform
        const int kp = piv >> 6, lp = piv & 63;
        const bool kp0 = kp == 0, kp2 = kp == 2, kplo = kp < 2;
        double pc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) pc[q] = colbuf[cur * 256 + lane + 64 * q];
        CAR_SUB(0);
        const int nxt = s + 1;
        const bool owner = nxt < NC && (nxt % CAR_PW) == wave;          // phi[0] is the next pivot column
        if (owner) {
            __builtin_amdgcn_s_setprio(3);                              // the critical chain of the step
            // mu[:] = mu - alpha * Phi[:,0]; mu[idx] = 0   (two roundings like the tensor expression, :253-254)
            CAR_MU_STEP(alpha, piv, pc)
#pragma unroll
            for (int q = 0; q < 4; ++q) mubuf[lane + 64 * q] = mu4[q];
            CAR_ELIM(0)
            CAR_RATIO_TEST(cur ^ 1)
            __builtin_amdgcn_s_setprio(0);
            CAR_SUB(1);
#ifdef CAR_STAMPS
            acc_[4] += tl_ - it0_; acc_[5] += 1;
#endif
        } else {
            CAR_ELIM_LIVE(0)
            CAR_SUB(2);
        }
#ifdef CAR_STAMPS
        arr_ += __builtin_amdgcn_s_memtime() - it0_;
        if (s == 50 && lane == 0) ((unsigned long long*)Phi)[(size_t)205 * CAR_PC + 64 + wave] = __builtin_amdgcn_s_memtime() - it0_;
#endif
        __syncthreads();
#ifdef CAR_STAMPS
        if (owner) { acc_[6] += __builtin_amdgcn_s_memtime() - tl_; }
#endif
        if (owner) {
            // off the critical path: the owner catches up with its other columns while the next owner (another
            // wave) is already working on step s+1; then its next column moves to slot 0.  (The empty asm pins
            // this register-only code behind the barrier: the compiler is otherwise free to hoist it.)
            asm volatile("" : "+v"(rpp_));
            CAR_ELIM_LIVE(1)
#pragma unroll
            for (int j = 0; j + 1 < CAR_PJW; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) phi[j][q] = phi[j + 1][q];
#pragma unroll
            for (int q = 0; q < 4; ++q) phi[CAR_PJW - 1][q] = 0.0;
            --nlive;
#ifdef CAR_STAMPS
            acc_[7] += __builtin_amdgcn_s_memtime() - tl_;
#endif
        }
        CAR_SUB(3);
    }
#ifdef CAR_STAMPS
    if (lane == 0) ((unsigned long long*)Phi)[(size_t)206 * CAR_PC + 64 + wave] = arr_;
    if (lane == 0 && wave < 2) for (int q_ = 0; q_ < 8; ++q_) ((unsigned long long*)Phi)[(size_t)207 * CAR_PC + 112 + 8 * wave + q_] = acc_[q_];
#endif
#undef CAR_RATIO_TEST
#undef CAR_ELIM
#undef CAR_ELIM_LIVE
#undef CAR_ELIM_IF

    // ---------------- output: w_star = mu[mu > 0], idx_star as ranks ----------------
    if (wave == 0) {
        double fin[4];
        if (s == NC && NC > 0) {                                        // the update of the last pivot is still due
            const int lc = (NC - 1) & 1;
            const double* pcl = colbuf + lc * 256;
            const double pcv[4] = {pcl[lane], pcl[lane + 64], pcl[lane + 128], pcl[lane + 192]};
            CAR_MU_STEP(pscal[lc * 4], (int)pscal[lc * 4 + 1], pcv)
#pragma unroll
            for (int q = 0; q < 4; ++q) fin[q] = mu4[q];
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) fin[q] = mubuf[lane + 64 * q];
        }
        int base = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = lane + 64 * q;
            const double v = (r < N) ? fin[q] + 0.0 : 0.0;              // -0.0 (dead) -> +0.0
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
#undef CAR_MU_STEP
}


// ---------------- phase 3, streaming form ----------------
// The same pivots without a workgroup barrier per pivot (the design of car_mc.hip's k_mc_pivot, through LDS instead of
// L2).  A wave owns SP_BC CONSECUTIVE columns and keeps its own copy of the weights.  While a column of its block is the
// pivot column it runs the ratio test on its registers, publishes (column, index, alpha, 1/pivot) in an LDS ring and goes
// straight on to its next column -- seven of eight pivots need no hand-over at all; every other wave applies the
// published pivots to its columns as they arrive, at low priority, in the issue slots the owner's dependent chain
// leaves empty.  k_car_pivot above rotates the ownership with every pivot (column c -> wave c mod 16): one barrier,
// one hand-over of the weights and one LDS round trip of the pivot column per pivot, 1.14 us each.
constexpr int SP_W = 16, SP_BC = 7, SP_RING = 32;
struct SpSlot { double col[256]; double alpha, rpp; int piv; int tag; };
struct SpState { double mu[4]; bool dead[4], inr[4]; };

__device__ __forceinline__ void sp_ratio_test(const double (&col)[4], const SpState& st, int& piv, double& al, double& rp) {
    double rt[4], rc[4];
    unsigned kh[4], kl[4];
#pragma unroll
#ifdef SP_X_NODIV      // timing-only builds (wrong results): what each part of the producer's loop costs
    for (int q = 0; q < 4; ++q) { rt[q] = st.mu[q] * col[q]; rc[q] = col[q]; }
#else
    for (int q = 0; q < 4; ++q) { rt[q] = st.mu[q] / col[q]; rc[q] = 1.0 / col[q]; }
#endif
    unsigned hmin = 0xffffffffu;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned long long k = ratio_key(rt[q]);
        const bool ok = st.inr[q] & (col[q] > 0.0) & !st.dead[q];
        kh[q] = ok ? (unsigned)(k >> 32) : 0xffffffffu;
        kl[q] = ok ? (unsigned)k : 0xffffffffu;
        hmin = min(hmin, kh[q]);
    }
    const unsigned H = wave_min_u32(hmin);
    piv = -1; al = 0.0; rp = 1.0;
    if (H == 0xffffffffu) return;                                     // uniform: no candidate (:241-242)
    unsigned long long mb[4];
    int cnt = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) { mb[q] = __ballot(kh[q] == H); cnt += __popcll(mb[q]); }
    if (cnt != 1) {                                                   // rare: several quotients share the high word
        unsigned lmin = 0xffffffffu;
#pragma unroll
        for (int q = 0; q < 4; ++q) lmin = min(lmin, (kh[q] == H) ? kl[q] : 0xffffffffu);
        const unsigned Lw = wave_min_u32(lmin);
#pragma unroll
        for (int q = 0; q < 4; ++q) mb[q] = __ballot((kh[q] == H) & (kl[q] == Lw));
    }
    bool found = false;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (!found && mb[q] != 0ull) {                                // uniform
            const int f = __ffsll((long long)mb[q]) - 1;
            piv = f + 64 * q;
            al = rdlane(rt[q], f);
            rp = rdlane(rc[q], f);
            found = true;
        }
    }
}
// mu[:] = mu - alpha * Phi[:, 0]; mu[idx] = 0  (two roundings like the tensor expression, :253-254)
__device__ __forceinline__ void sp_mu_step(SpState& st, const double (&col)[4], double alpha, int piv, int lane) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        st.dead[q] = st.dead[q] | (lane + 64 * q == piv);
        st.mu[q] = st.dead[q] ? 0.0 : __dsub_rn(st.mu[q], __dmul_rn(alpha, col[q]));
    }
}
//   Phi[:, c] -= Phi[:, 0] * (Phi[idx, c] / Phi[idx, 0])   (:260-266), my columns J0 .. SP_BC-1
template <int KP, int J0>
__device__ __forceinline__ void sp_elim_kp(double (&phi)[SP_BC][4], const double (&col)[4], int lp, double rpp) {
#pragma unroll
    for (int j = J0; j < SP_BC; ++j) {
        const double qv = rdlane(phi[j][KP], lp) * rpp;
#pragma unroll
        for (int q = 0; q < 4; ++q) phi[j][q] = fma(-qv, col[q], phi[j][q]);
    }
}
template <int J0>
__device__ __forceinline__ void sp_elim(double (&phi)[SP_BC][4], const double (&col)[4], int piv, double rpp) {
    const int kp = piv >> 6, lp = piv & 63;
    switch (kp) {                                                     // uniform
        case 0: sp_elim_kp<0, J0>(phi, col, lp, rpp); break;
        case 1: sp_elim_kp<1, J0>(phi, col, lp, rpp); break;
        case 2: sp_elim_kp<2, J0>(phi, col, lp, rpp); break;
        default: sp_elim_kp<3, J0>(phi, col, lp, rpp); break;
    }
}
// wait for pivot s in the ring; false = give up (bounded)
__device__ __forceinline__ bool sp_consume(SpSlot* ring, int s, int lane, double (&col)[4], double& alpha, double& rpp, int& piv) {
    SpSlot& e = ring[s % SP_RING];
    volatile int* tg = &e.tag;
    unsigned spins = 0;
    while (*tg != s + 1) {
        if (++spins > (1u << 24)) return false;
        __builtin_amdgcn_s_sleep(1);
    }
    alpha = e.alpha; rpp = e.rpp; piv = e.piv;
#pragma unroll
    for (int q = 0; q < 4; ++q) col[q] = e.col[lane + 64 * q];
    return true;
}

// pivot `sp` from column JJ of my block: ratio test, publish, update of my weights and of my columns behind JJ
template <int JJ>
__device__ __forceinline__ void sp_produce_step(double (&phi)[SP_BC][4], SpState& st, SpSlot* ring, int sp, int lane,
                                                double (&col)[4], bool& stop) {
#pragma unroll
    for (int q = 0; q < 4; ++q) col[q] = (st.dead[q] | !st.inr[q]) ? 0.0 : phi[JJ][q];
    int piv;
    double al, rp;
    sp_ratio_test(col, st, piv, al, rp);
    SpSlot& e = ring[sp % SP_RING];
#ifndef SP_X_NOPUB
    if (piv >= 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) e.col[lane + 64 * q] = col[q];
    }
#endif
    if (lane == 0) { e.alpha = al; e.rpp = rp; e.piv = piv; }
    asm volatile("" ::: "memory");                            // (program order; a wave's LDS operations execute in order:
    if (lane == 0) *(volatile int*)&e.tag = sp + 1;           //  the tag lands after the data it releases -- no wait)
    if (piv < 0) { stop = true; return; }                     // Q6: the loop ends here (:241-242)
    sp_mu_step(st, col, al, piv, lane);
#ifndef SP_X_NOELIM
    sp_elim<JJ + 1>(phi, col, piv, rp);
#endif
}

__global__ __launch_bounds__(SP_W * 64) void k_car_pivot_stream(const double* __restrict__ Phi, int N, int m,
                                                                const double* __restrict__ mu_in,
                                                                int32_t* __restrict__ keep_rank,
                                                                double* __restrict__ w_star,
                                                                int32_t* __restrict__ n_keep_out,
                                                                double* __restrict__ mu_out,
                                                                const unsigned* __restrict__ err) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sp_lds[];
    SpSlot* ring = (SpSlot*)sp_lds;
    volatile int* prog = (volatile int*)(sp_lds + sizeof(SpSlot) * SP_RING);    // [SP_W]: pivots consumed so far
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = N - m;
    const int c0 = w * SP_BC;
    if (tid < SP_RING) ring[tid].tag = 0;
    if (tid < SP_W) prog[tid] = 0;
    // the fused launch in front gave up on a reflector, or not every group of Phi's rows found a consumer
    const bool broken = err != nullptr && (err[CARF_ERR / 4] != 0u || err[CARF_DONE / 4] != (unsigned)(CAR_NS / 8));
    __syncthreads();                                                  // (the only workgroup barrier)
    if (broken) { if (tid == 0) *n_keep_out = -1; return; }
    if (c0 >= K && w != 0) return;
    double phi[SP_BC][4];
    SpState st;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = lane + 64 * q;
        st.inr[q] = row < N;
        st.dead[q] = false;
        st.mu[q] = st.inr[q] ? mu_in[row] + 0.0 : 0.0;
#pragma unroll
        for (int j = 0; j < SP_BC; ++j)
            phi[j][q] = (c0 + j < K && row < N) ? Phi[(size_t)row * CAR_PC + c0 + j] : 0.0;
    }
    bool fail = false, stop = false;
    double col[4] = {0.0, 0.0, 0.0, 0.0};
    // the pivots before my block: apply them to all my columns as they arrive
    const int s_mine = min(c0, K);
    for (int s = 0; s < s_mine && !fail && !stop; ++s) {
        double al, rp;
        int piv;
        if (s == s_mine - SP_BC) __builtin_amdgcn_s_setprio(2);      // on deck: the hand-over is on the critical chain
        if (!sp_consume(ring, s, lane, col, al, rp, piv)) { fail = true; break; }
        if (lane == 0) prog[w] = s + 1;
        if (piv < 0) { stop = true; break; }
        sp_mu_step(st, col, al, piv, lane);
        sp_elim<0>(phi, col, piv, rp);
    }
    // my block
    if (!fail && !stop && c0 < K) {
        __builtin_amdgcn_s_setprio(3);                                // the critical chain of the whole kernel
        const int s_end = min(c0 + SP_BC, K);
        // a ring slot is reused SP_RING pivots later: everybody who still follows must be past what I overwrite
        if (s_end > SP_RING) {
            const int need = s_end - SP_RING;
            unsigned spins = 0;
            for (;;) {
                bool ok = true;
                for (int o = 0; o < SP_W; ++o) {
                    const bool follows = (o > w && o * SP_BC < K) || (o == 0 && w != 0);
                    if (follows && prog[o] < need) ok = false;
                }
                if (ok) break;
                if (++spins > (1u << 22)) { fail = true; break; }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        // one instance of the step per column of my block: the columns keep their registers (rotating them down after
        // every pivot -- 24 doubles moved behind the elimination they depend on -- cost 15 % of the kernel)
        int sp = c0;
#define SP_STEP(JJ) if (sp < s_end && !stop) { sp_produce_step<JJ>(phi, st, ring, sp, lane, col, stop); ++sp; }
        SP_STEP(0) SP_STEP(1) SP_STEP(2) SP_STEP(3) SP_STEP(4) SP_STEP(5) SP_STEP(6)
#undef SP_STEP
        static_assert(SP_BC == 7, "one SP_STEP per column of a block");
        __builtin_amdgcn_s_setprio(0);
    }
    if (w != 0) return;
    // wave 0 follows the remaining pivots for the weights and writes the result
    for (int s = c0 + SP_BC; s < K && !fail && !stop; ++s) {
        double al, rp;
        int piv;
        if (!sp_consume(ring, s, lane, col, al, rp, piv)) { fail = true; break; }
        if (lane == 0) prog[0] = s + 1;
        if (piv < 0) { stop = true; break; }
        sp_mu_step(st, col, al, piv, lane);
    }
    int base = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = lane + 64 * q;
        const double v = (row < N) ? st.mu[q] + 0.0 : 0.0;           // -0.0 -> +0.0
        const bool keep = (row < N) && (v > 0.0);
        const unsigned long long bal = __ballot(keep);
        const int rank = base + __popcll(bal & ((1ull << lane) - 1ull));
        if (row < N) {
            keep_rank[row] = keep ? rank : -1;
            mu_out[row] = v;
            if (keep) w_star[rank] = v;
        }
        base += __popcll(bal);
    }
    if (lane == 0) *n_keep_out = fail ? -1 : base;
}

// ---------------- the extra elimination of the acquisition-guided branch (SOBER/_rchq.py:87-106, :177-196) ----------
// After the Caratheodory step with one more test function (the objective), n1 = b + 1 points are left; their
// weights move along the null vector w_null of [X_p; 1] (functions x points, a 1-dimensional null space) in the
// direction that does not decrease sum w * calc_obj until one more weight reaches zero.  w_null arrives as the
// one-column null-space basis of sober_car_device's phi_out (unit norm, sign arbitrary -- the reference fixes the
// sign by the objective, :92-93, so ANY null vector gives its result).  One workgroup; n1 <= 512.
//   in : phi[n1], objp[n1], w1[n1] (ranks of the first step), rank1[Nsets] (set -> rank or -1)
//   out: keep_rank[Nsets] (new ranks), w_star[n_keep], *n_keep_out
__global__ __launch_bounds__(512) void k_second_elim(const double* __restrict__ phi, const double* __restrict__ objp,
                                                     const double* __restrict__ w1, const int32_t* __restrict__ rank1,
                                                     int n1, int Nsets, int32_t* __restrict__ keep_rank,
                                                     double* __restrict__ w_star, int32_t* __restrict__ n_keep_out) {
    __shared__ double red[512];
    __shared__ unsigned long long kmin[512];
    __shared__ int kidx[512];
    __shared__ int cnt[512];
    const int t = threadIdx.x;
    const double ph = t < n1 ? phi[t] : 0.0, ob = t < n1 ? objp[t] : 0.0, w = t < n1 ? w1[t] : 0.0;
    red[t] = ob * ph;                                                   // torch.dot(obj_p, w_null): only its sign is used
    __syncthreads();
    for (int h = 256; h > 0; h >>= 1) {
        if (t < h) red[t] += red[t + h];
        __syncthreads();
    }
    const double wn = (red[0] < 0.0) ? -ph : ph;                        // :92-93
    // alpha[plis] = w_star[plis] / w_null[plis]; first argmin (a NaN quotient wins like torch.argmin)
    const bool ok = t < n1 && wn > 0.0;
    const double al = w / wn;
    kmin[t] = ok ? ratio_key(al) : ~0ull;
    kidx[t] = t;
    __syncthreads();
    for (int h = 256; h > 0; h >>= 1) {
        if (t < h) {
            const unsigned long long a = kmin[t], b = kmin[t + h];
            if (b < a || (b == a && kidx[t + h] < kidx[t])) { kmin[t] = b; kidx[t] = kidx[t + h]; }
        }
        __syncthreads();
    }
    const bool any = kmin[0] != ~0ull;
    const int piv = kidx[0];
    __shared__ double s_al;
    if (t == piv) s_al = al;
    __syncthreads();
    // w_star = w_star - alpha[idx_sp] * w_null; w_star[idx_sp] = 0   (two roundings, :99-100)
    double w2 = w;
    if (any && t < n1) w2 = (t == piv) ? 0.0 : __dsub_rn(w, __dmul_rn(s_al, wn));
    const bool keep = t < n1 && w2 > 0.0;
    cnt[t] = keep ? 1 : 0;
    __syncthreads();
    for (int h = 1; h < 512; h <<= 1) {                                  // inclusive scan
        const int v = (t >= h) ? cnt[t - h] : 0;
        __syncthreads();
        cnt[t] += v;
        __syncthreads();
    }
    if (keep) w_star[cnt[t] - 1] = w2;
    red[t] = keep ? (double)(cnt[t] - 1) : -1.0;                        // new rank by old rank
    __syncthreads();
    for (int sidx = t; sidx < Nsets; sidx += 512) {
        const int r1 = rank1[sidx];
        keep_rank[sidx] = (r1 >= 0 && r1 < n1) ? (int)red[r1] : -1;
    }
    if (t == 0) *n_keep_out = cnt[511];
}

}  // namespace sober

extern "C" int sober_car_mc_supported(int N, int m);
extern "C" int64_t sober_car_mc_ws_bytes(int N, int m);
extern "C" int sober_car_mc_device(const double* X, int ldx, int N, int m, const double* mu_in, int32_t* keep_rank,
                                   double* w_star, int32_t* n_keep, double* mu_out, double* phi_out, void* ws,
                                   int64_t ws_bytes, void* stream);

// one compute unit (batch <= 100)
static int car_one_cu(int N, int m) {
    return (m >= 2 && N > m && N <= sober::CAR_NS && m <= 16 * sober::CAR_MS && N - m <= sober::CAR_PW * sober::CAR_PJW) ? 1 : 0;
}

extern "C" int sober_car_supported(int N, int m) {
    return (car_one_cu(N, m) || sober_car_mc_supported(N, m)) ? 1 : 0;
}

// scratch: reflectors (m x 208), tau (m, padded to 128), Phi (208 x 128)
// (a workspace sized for (N, m) also serves every (N' <= N, m): the final direct level)
extern "C" int64_t sober_car_ws_bytes(int N, int m) {
    const int64_t one = ((int64_t)m * sober::CAR_NS + 128 + (int64_t)sober::CAR_NS * sober::CAR_PC + 512) * (int64_t)sizeof(double)   // (+512: stamp block)
                        + sober::carf_bytes(m);                                       // + the fused launch's granules
    if (car_one_cu(N, m)) return one;
    const int64_t mc = sober_car_mc_ws_bytes(N, m);
    return mc > one ? mc : one;
}

extern "C" int sober_car_device(const double* X, int ldx, int N, int m, const double* mu_in,
                                int32_t* keep_rank, double* w_star, int32_t* n_keep, double* mu_out,
                                double* phi_out, void* ws, int64_t ws_bytes, void* stream) {
    if (!X || !mu_in || !keep_rank || !w_star || !n_keep || !mu_out || !ws || ldx < m - 1) return SOBER_E_ARG;
    if (!car_one_cu(N, m))                                  // beyond one compute unit: car_mc.hip
        return sober_car_mc_device(X, ldx, N, m, mu_in, keep_rank, w_star, n_keep, mu_out, phi_out, ws, ws_bytes, stream);
    if (ws_bytes < sober_car_ws_bytes(N, m)) return SOBER_E_WS;
    hipStream_t st = (hipStream_t)stream;
    double* vws = (double*)ws;
    double* taup = vws + (size_t)m * sober::CAR_NS;
    double* Phi = taup + 128;
    static const bool unfused = getenv("SOBER_CAR_UNFUSED") != nullptr;                // (A/B switches)
    static const bool barrier_pivot = getenv("SOBER_CAR_PIVOT_BARRIER") != nullptr;
    const size_t sp_bytes = sizeof(sober::SpSlot) * sober::SP_RING + sober::SP_W * sizeof(int);
    static bool sp_attr = false;
    if (!sp_attr) {
        HIP_TRY(hipFuncSetAttribute((const void*)sober::k_car_pivot_stream, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)sp_bytes));
        sp_attr = true;
    }
#define CAR_LAUNCH_PIVOT(ERR)                                                                                          \
    if (barrier_pivot)                                                                                                 \
        hipLaunchKernelGGL(sober::k_car_pivot, dim3(1), dim3(sober::CAR_PW * 64), 0, st, Phi, N, m, mu_in, keep_rank,   \
                           w_star, n_keep, mu_out, (const unsigned*)(ERR));                                            \
    else                                                                                                               \
        hipLaunchKernelGGL(sober::k_car_pivot_stream, dim3(1), dim3(sober::SP_W * 64), sp_bytes, st, Phi, N, m, mu_in,  \
                           keep_rank, w_star, n_keep, mu_out, (const unsigned*)(ERR));
    if (phi_out != nullptr || unfused) {
        hipLaunchKernelGGL(sober::k_car_bidiag, dim3(1), dim3(sober::CAR_BT), 0, st, X, ldx, N, m, vws, taup);
        LAUNCH_CHECK();
        hipLaunchKernelGGL(sober::k_car_phi, dim3(sober::CAR_PC / 4), dim3(256), 0, st, vws, taup, N, m, Phi, phi_out);
        LAUNCH_CHECK();
        CAR_LAUNCH_PIVOT(nullptr)
        LAUNCH_CHECK();
        return 0;
    }
    // bidiagonalisation and Phi in ONE launch: workgroup 0 produces the reflectors, the workgroups that land on its XCD
    // (an eighth of the rest: 26 are needed, 40 are offered) accumulate the rows of Phi as the reflectors appear
    static std::atomic<unsigned> epoch_ctr{0};
    const unsigned epoch = (epoch_ctr.fetch_add(1) + 1u) & 0x1FFFFFFu;
    void* comm = (void*)(Phi + (size_t)sober::CAR_NS * sober::CAR_PC + 512);
    static const int per_xcd = getenv("SOBER_CARF_PER_XCD") ? atoi(getenv("SOBER_CARF_PER_XCD")) : 40;    // (tuning aid)
    hipLaunchKernelGGL(sober::k_car_bidiag_fused, dim3(1 + 8 * per_xcd), dim3(sober::CAR_BT), 0, st, X, ldx, N, m, vws, taup, Phi, comm,
                       (unsigned)sober::carf_bytes(m), epoch);
    LAUNCH_CHECK();
    CAR_LAUNCH_PIVOT(comm)
    LAUNCH_CHECK();
    return 0;
#undef CAR_LAUNCH_PIVOT
}

extern "C" int sober_second_elimination(const double* phi, const double* objp, const double* w1, const int32_t* rank1,
                                        int n1, int Nsets, int32_t* keep_rank, double* w_star, int32_t* n_keep,
                                        void* stream) {
    if (!phi || !objp || !w1 || !rank1 || !keep_rank || !w_star || !n_keep || n1 <= 0 || Nsets <= 0) return SOBER_E_ARG;
    if (n1 > 512) return SOBER_E_DIM;
    hipLaunchKernelGGL(sober::k_second_elim, dim3(1), dim3(512), 0, (hipStream_t)stream, phi, objp, w1, rank1, n1, Nsets,
                       keep_rank, w_star, n_keep);
    LAUNCH_CHECK();
    return 0;
}
