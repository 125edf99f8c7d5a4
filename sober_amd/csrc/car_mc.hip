// Tchernychova_Lyons_CAR (SOBER/_rchq.py:224-270) for batches beyond one compute unit: N <= 448, m <= 256
// (batch <= 224; BASELINE.json configs[2] has batch 200 -> A = [1 | X]^T is 200 x 400 = 640 KB, four times the
// LDS and 1.25 x the whole VGPR file of a CU).  Same mathematics and the same reflectors as car.hip (the null-space
// basis LAPACK's gesdd returns: Phi = G(0) ... G(m-1) [0; I] with G(i) the right Householder reflectors of the
// Golub-Kahan bidiagonalisation, dgebd2 / dlarfg conventions); what changes is where the matrix lives and how the
// workgroups talk.
//
//   k_mc_bidiag  G = 9 workgroups x 4 waves, one wave per SIMD.  The matrix lives in VGPRs, lane <-> row
//                (row = lane + 64 s), a wave owns whole columns (column c -> wave c mod 36).  One step i then needs
//                ONE exchange across the workgroups: every wave's partial of q = A[:, c > i] . row_i (a per-lane FMA
//                chain, no cross-lane traffic) is summed over the workgroup in LDS and over the G workgroups through
//                L2; with q, |row_i|^2 and column i on every CU, G(i), column i after G(i), H(i) and w = A v are
//                computed redundantly by every wave, z = u^T A is local to the owner of a column (a wave-level
//                transposed reduction through a private LDS tile), and the two rank-1 updates are fused.
//                The exchange uses data-tagged granules (cdna_hip_programming.md Guideline 16, form R2): a double
//                travels as two 8-byte {32 data bits, 32-bit tag = step + 1} words in one 16-byte store, the reader
//                polls the granule itself with L1-bypassing (sc1) loads until both tags match -- no flag, no fence.
//                The nine workgroups are ELECTED ONTO ONE XCD (elect(): XCC id from the hardware register + a ticket
//                per XCD), so the stores are plain stores into the L2 all workers share: an exchange costs 0.9 us
//                instead of 1.4-1.5 us through the fabric (scripts/xcd_exchange_probe.hip).  Two slots alternate (a
//                slot is rewritten at step i + 2, and nobody can publish step i + 1 before it has read step i).
//   k_mc_phi     Phi = P [0; I]: one wave per column, backward accumulation (as k_car_phi, 7 row slots).
//   k_mc_pivot   the N - m pivots of :237-266 as a streaming pipeline of 32 independent waves on one XCD (no barrier):
//                a wave owns 8 consecutive columns of Phi (lane <-> row, 7 slots); it applies the published pivots
//                (column, index, alpha, 1 / Phi[idx, 0]) to its columns in order as they arrive, and when the next
//                pivot column is its own it runs the ratio test and publishes.  Inside a block of 8 pivots there is
//                no exchange; a block hand-over costs one L2 hop.  Every pivot has its own slot (tag = pivot + 1).
//
// The communication area is zeroed by a hipMemsetAsync ahead of the launches of EVERY call; all spins are bounded
// (a give-up writes an error word that makes the step report n_keep = -1).
#include "common.hpp"
#include <cstdlib>

namespace sober {
namespace mc {

constexpr int RS_MAX = 4;             // row slots of the bidiagonalisation: m <= 256
constexpr int CPW = 13;               // columns per wave in the bidiagonalisation
constexpr int NQ = 7;                 // 64-lane slots along N
constexpr int NS = 64 * NQ;           // 448: stride of a reflector vector and of a Phi column
constexpr int GMAX = 9;               // workgroups of the bidiagonalisation, always all nine: 36 waves x 13 >= 448
constexpr int NW = 4 * GMAX;          // column c lives in wave c mod 36
constexpr int BC = 8;                 // pivot columns per wave
constexpr int PWAVES = 32;            // waves of the pivot kernel: N - m <= 256
constexpr int KMAX = BC * PWAVES;
constexpr unsigned SPIN_LIMIT = 1u << 22;
constexpr int ELECT_GRID = 128;       // workgroups launched per kernel: some XCD gets >= 16 of them whatever the placement
constexpr int FUSED_GRID = 8 * 32;    // bidiagonalisation + Phi in one launch: ONE workgroup per CU (forced by an LDS request that
                                      // two do not fit: a producer that shares its SIMDs slows all nine down) -- on the winning
                                      // XCD 9 producers and 23 consumers
constexpr int FUSED_LDS_PAD = 88 * 1024;
constexpr int PHI_RW = 5;             // rows of P per consumer wave (20 per workgroup: 23 groups for 448 rows)
constexpr int DBG_WORDS = 8192;       // stamp block at the end of the workspace (diagnostic build)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

// byte offsets inside the zeroed communication area
constexpr unsigned OFF_ERR = 0;                                   // error word (+ padding)
constexpr unsigned OFF_EL_B = 16;                                 // election of the bidiagonalisation: count[8], winner
constexpr unsigned OFF_EL_P = 64;                                 // election of the pivot kernel: count[8], winner
constexpr unsigned OFF_PROG = 128;                                // [GMAX] dwords: reflectors < k of this producer workgroup are in memory
constexpr unsigned OFF_TICKET = 176;                              // row-group tickets of the Phi consumers (fused launch)
constexpr unsigned OFF_DONE = 180;                                // row groups of Phi completed (fused launch)
constexpr unsigned OFF_Q = 192;                                   // [2][GMAX][256] granules: per-CU partial row dots
constexpr unsigned OFF_C = OFF_Q + 2u * GMAX * 256u * 16u;        // [2][256]: column i
constexpr unsigned OFF_S = OFF_C + 2u * 256u * 16u;               // [2][16]: per-CU partial |row_i|^2
constexpr unsigned OFF_H = OFF_S + 2u * 16u * 16u;                // [KMAX][4]: pivot headers (alpha, 1/pivot, index)
constexpr unsigned OFF_P = OFF_H + (unsigned)KMAX * 64u;          // [K][NS]: pivot columns
__host__ __device__ constexpr unsigned comm_bytes(int K) { return OFF_P + (unsigned)K * NS * 16u; }

// ---- cross-lane helpers --------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ double dpp(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);          // (no `old` operand: every lane is written)
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
constexpr int ROR1 = 0x121, ROR2 = 0x122, ROR4 = 0x124, ROR8 = 0x128;
constexpr int BCAST15 = 0x142, BCAST31 = 0x143;

__device__ __forceinline__ double row16_allsum(double v) {     // every lane of a 16-lane row: the row total, bitwise equal
    v += dpp<ROR8>(v);
    v += dpp<ROR4>(v);
    v += dpp<ROR2>(v);
    v += dpp<ROR1>(v);
    return v;
}
// sum over the four 16-lane rows (lanes l, l^16, l^32, l^48), bitwise equal in all four; gfx950 lane swaps:
// v_permlane16_swap exchanges the odd rows of its first operand with the even rows of its second,
// v_permlane32_swap the upper half of the first with the lower half of the second
__device__ __forceinline__ double xrow_allsum(double v) {
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
__device__ __forceinline__ double wave_allsum(double v) { return xrow_allsum(row16_allsum(v)); }

__device__ __forceinline__ double rdlane(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// v_rcp_f64 / v_rsq_f64 seeds + Newton steps: <= 1 ulp, a third of the dependent chain of the IEEE sequences
// (the reflector scalars sit on the critical path of every step); operands far outside the normal range take the
// IEEE route
__device__ __forceinline__ double fast_rcp(double c) {
    double r = __builtin_amdgcn_rcp(c);
    double e = fma(-c, r, 1.0);
    r = fma(r, e, r);
    e = fma(-c, r, 1.0);
    return fma(r, e, r);
}
// (reflector scalars: larfg_vt of common.hpp -- tau and the scale of v from 1 / norm alone, beta is never needed)
// workgroup barrier that waits for this wave's LDS traffic only (global stores and loads stay in flight)
#define MC_LDS_BARRIER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)

// ---- granules -------------------------------------------------------------------------------------------------
__device__ __forceinline__ void put_granule(rsrc_t rs, unsigned off, double v, unsigned tag) {
    u32x4 g;
    g.x = (unsigned)__double2loint(v); g.y = tag; g.z = (unsigned)__double2hiint(v); g.w = tag;
    __builtin_amdgcn_raw_buffer_store_b128(g, rs, off, 0, 0);                    // plain: the line stays in the XCD's L2
}
__device__ __forceinline__ u32x4 load_granule(rsrc_t rs, unsigned off) {
    return __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16);                // sc1: past this CU's L1, served by L2
}
__device__ __forceinline__ bool granule_ok(const u32x4& g, unsigned tag) { return (g.y == tag) & (g.w == tag); }
__device__ __forceinline__ double granule_val(const u32x4& g) { return __hiloint2double((int)g.z, (int)g.x); }

__device__ __forceinline__ unsigned load_err(rsrc_t rs) {
    return (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs, OFF_ERR, 0, 16);
}
__device__ __forceinline__ void store_err(rsrc_t rs, unsigned code) {
    __builtin_amdgcn_raw_buffer_store_b32(code, rs, OFF_ERR, 0, 16);
}
// one more unsuccessful poll: true = give up (limit reached, or somebody else already gave up)
__device__ __forceinline__ bool spin_fail(rsrc_t rs, unsigned& spins, unsigned code) {
    ++spins;
    if ((spins & 1023u) == 0u && load_err(rs) != 0u) return true;
    if (spins > SPIN_LIMIT) { store_err(rs, code); return true; }
    __builtin_amdgcn_s_sleep(1);
    return false;
}

// Workers of one launch all sit on ONE XCD, by construction: every workgroup reads its XCC id from the hardware
// register and takes a ticket on that XCD's counter; the XCD whose n-th ticket is drawn first wins, its tickets
// 0 .. n-1 are the workers (ticket = worker index), everybody else exits.  128 workgroups over 8 XCDs: some XCD
// always collects n <= 16 of them, whatever the dispatcher does.  What this buys: the workers share one L2, so a
// plain store by one is seen by an L1-bypassing (sc1) load of another after an L2 round trip (~0.1 us) -- the
// write-through + fabric path between XCDs costs 0.75 us per round trip and two to three of them per exchange.
// Called by thread 0; -1 = not a worker.
__device__ __forceinline__ int elect(void* comm, unsigned base, int n, rsrc_t rs) {
    unsigned* cnt = (unsigned*)((char*)comm + base);
    unsigned* win = cnt + 8;
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u;       // HW_REG_XCC_ID
    const unsigned t = __hip_atomic_fetch_add(cnt + xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t >= (unsigned)n) return -1;
    if (t == (unsigned)n - 1u) {
        unsigned expect = 0u;
        __hip_atomic_compare_exchange_strong(win, &expect, xcc + 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT);
    }
    unsigned spins = 0, w;
    while ((w = __hip_atomic_load(win, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u) {
        if (spin_fail(rs, spins, 0x10u)) return -1;
    }
    return (w == xcc + 1u) ? (int)t : -1;
}

// The same election for a launch in which the losers of the winning XCD stay on as CONSUMERS: -1 = exit (another
// XCD), 0 .. n-1 = worker, >= n = consumer number (ticket - n).  Called by thread 0.
__device__ __forceinline__ int elect_all(void* comm, unsigned base, int n, rsrc_t rs) {
    unsigned* cnt = (unsigned*)((char*)comm + base);
    unsigned* win = cnt + 8;
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u;       // HW_REG_XCC_ID
    const unsigned t = __hip_atomic_fetch_add(cnt + xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == (unsigned)n - 1u) {
        unsigned expect = 0u;
        __hip_atomic_compare_exchange_strong(win, &expect, xcc + 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT);
    }
    unsigned spins = 0, w;
    while ((w = __hip_atomic_load(win, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u) {
        if (spin_fail(rs, spins, 0x10u)) return -1;
    }
    return (w == xcc + 1u) ? (int)t : -1;
}

// in-kernel stamps (diagnostic build only, `make stamps`): per-segment cycle sums of every wave -> the debug block
// at the end of the workspace, which nothing else reads
#ifdef MC_STAMPS
#define MC_STAMP_DECL unsigned long long acc_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tl_; \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tl_) :: "memory");
#define MC_STAMP(K) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
    acc_[K] += t_ - tl_; tl_ = t_; } while (0)
#define MC_STAMP_FLUSH(DBG, SLOT) do { if ((threadIdx.x & 63) == 0) for (int k_ = 0; k_ < 12; ++k_) (DBG)[(SLOT) * 12 + k_] = acc_[k_]; } while (0)
#else
#define MC_STAMP_DECL
#define MC_STAMP(K) do { } while (0)
#define MC_STAMP_FLUSH(DBG, SLOT) do { } while (0)
#endif

// ================================================================================================================
// phase 1: bidiagonalisation
// ================================================================================================================
template <int RS>
struct BidiagLds {
    double lq[4][64 * RS];        // per-wave partial row dots
    double lqt[64 * RS];          // all-reduced row dots
    double lcol[64 * RS];         // column i
    double lcp[64 * RS];          // column i on its way out (owner wave -> publisher waves)
    double ltr[4][64 * 17];       // per-wave transposition tile of the column sums
    double lss[4];
    double lsc[2];                // |row_i|^2, A[i][i] after the exchange
    int ldead;
    int lgrp;                     // row group of a Phi consumer
};

// BODY(j) for the live register slots j = 0 .. nlive-1, written out (an early exit inside `#pragma unroll` keeps the
// loop rolled and the arrays in scratch): not-taken compares while live, ONE taken branch at the first dead slot
// (slots 0 .. 4 are ALWAYS live: nlive = CPW - ((i - gw) / NW + 1) >= 13 - (255 / 36 + 1) = 5 for every step i < m <= 64 RS_MAX
//  = 256 -- sober_car_mc_supported -- so their five compare + branch pairs per use, 25 pairs per step, are not emitted)
static_assert(CPW - ((64 * RS_MAX - 1) / NW + 1) >= 5, "MC_FOR_LIVE runs slots 0 .. 4 unconditionally");
#define MC_FOR_LIVE(NL, BODY)                                                                      \
    do {                                                                                           \
        BODY(0) BODY(1) BODY(2) BODY(3) BODY(4)  if ((NL) <= 5) break; BODY(5)                       \
        if ((NL) <= 6) break; BODY(6)  if ((NL) <= 7) break; BODY(7)  if ((NL) <= 8) break; BODY(8)   \
        if ((NL) <= 9) break; BODY(9)  if ((NL) <= 10) break; BODY(10) if ((NL) <= 11) break; BODY(11) \
        if ((NL) <= 12) break; BODY(12)                                                            \
    } while (0)

// steps i = 64 SL .. min(64 SL + 63, m - 1); false = abort
template <int RS, int SL>
__device__ __forceinline__ bool bidiag_steps(double (&a)[CPW][RS], double (&r)[CPW], int m, int gw, int cu, int wv,
                                             BidiagLds<RS>& L, rsrc_t rs, double* __restrict__ vws,
                                             double* __restrict__ taup, unsigned long long* dbg) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int i_end = min(64 * SL + 64, m);
    MC_STAMP_DECL
    for (int i = 64 * SL; i < i_end; ++i) {
        MC_STAMP(0);
        __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0): my entries of reflector i - 1 are in memory (stored a step ago: free)
        const int li = i & 63;
        const unsigned tag = (unsigned)i + 1u;
        const unsigned par = (unsigned)i & 1u;
        // my columns, kept in REVERSE order (register slot j holds column (CPW-1-j) NW + gw), so that the live ones
        // (c > i) are the slots j < nlive: the loops below leave at the first dead slot -- one taken branch
        const int nlive = CPW - ((i >= gw) ? (i - gw) / NW + 1 : 0);  // uniform
        // ---- 1. row i of my live columns (wave-uniform values, kept in VGPRs: the SGPR file is the scarce one),
        //         my share of |row_i|^2 and of q = A[:, c > i] row_i
        double ssp = 0.0;
        double qp[RS];
#pragma unroll
        for (int s = 0; s < RS; ++s) qp[s] = 0.0;
#define MC_ROWQ(J)                                                                         \
        {                                                                                  \
            double rj_ = rdlane(a[J][SL], li);                                             \
            asm volatile("" : "+v"(rj_));                                                  \
            r[J] = rj_;                                                                    \
            ssp = fma(rj_, rj_, ssp);                                                      \
            _Pragma("unroll") for (int s = SL; s < RS; ++s) qp[s] = fma(a[J][s], rj_, qp[s]); \
        }
        MC_FOR_LIVE(nlive, MC_ROWQ);
#undef MC_ROWQ
        MC_STAMP(1);
        // ---- 2. combine inside the workgroup (LDS); the owner of column i hands it over
#pragma unroll
        for (int s = SL; s < RS; ++s) L.lq[wv][s * 64 + lane] = qp[s];
        if (lane == 0) L.lss[wv] = ssp;
        const int gwi = i % NW;                                     // uniform: the wave that owns column i
        if (gw == gwi) {
#define MC_COLI(J) case J: { _Pragma("unroll") for (int s = SL; s < RS; ++s) L.lcp[s * 64 + lane] = a[CPW - 1 - J][s]; } break;
            switch (i / NW) {
                MC_COLI(0) MC_COLI(1) MC_COLI(2) MC_COLI(3) MC_COLI(4) MC_COLI(5) MC_COLI(6)
                MC_COLI(7) MC_COLI(8) MC_COLI(9) MC_COLI(10) MC_COLI(11) MC_COLI(12)
                default: break;
            }
#undef MC_COLI
        }
        MC_LDS_BARRIER();
        MC_STAMP(2);
        if (tid == 0) __builtin_amdgcn_raw_buffer_store_b32((unsigned)i, rs, OFF_PROG + 4u * (unsigned)cu, 0, 0);   // (all four waves are past their wait)
        // ---- 3. the exchange: thread t > i owns row t (publishes this CU's partial q_t, gathers the nine partials
        //         and column i's entry), thread i carries the partial norms in the slot of the (finished) row i and
        //         the diagonal entry A[i][i]; measured forms: scripts/xcd_exchange_probe.hip
        {
            const bool mine = tid >= i && tid < 64 * RS;
            bool ok = true;
            if (mine) {
                double pv;
                if (tid == i) pv = ((L.lss[0] + L.lss[1]) + L.lss[2]) + L.lss[3];
                else pv = ((L.lq[0][tid] + L.lq[1][tid]) + L.lq[2][tid]) + L.lq[3][tid];
                put_granule(rs, OFF_Q + ((par * GMAX + (unsigned)cu) * 256u + (unsigned)tid) * 16u, pv, tag);
                if (cu == (gwi >> 2)) put_granule(rs, OFF_C + (par * 256u + (unsigned)tid) * 16u, L.lcp[tid], tag);
                unsigned spins = 0;
                double qt = 0.0, ct = 0.0;
                for (;;) {
                    u32x4 gq[GMAX], gc;
#pragma unroll
                    for (int g = 0; g < GMAX; ++g)
                        gq[g] = load_granule(rs, OFF_Q + ((par * GMAX + (unsigned)g) * 256u + (unsigned)tid) * 16u);
                    gc = load_granule(rs, OFF_C + (par * 256u + (unsigned)tid) * 16u);
                    ok = granule_ok(gc, tag);
                    qt = 0.0;
#pragma unroll
                    for (int g = 0; g < GMAX; ++g) {
                        ok &= granule_ok(gq[g], tag);
                        qt += granule_val(gq[g]);
                    }
                    ct = granule_val(gc);
                    if (__all(ok)) break;
                    if (spin_fail(rs, spins, 0x100u + (unsigned)i)) break;
                    asm volatile("" ::: "memory");
                }
                L.lqt[tid] = qt;
                L.lcol[tid] = ct;
                if (tid == i) { L.lsc[0] = qt; L.lsc[1] = ct; }
                if (!ok) L.ldead = 1;
            }
        }
        MC_STAMP(3);
        MC_LDS_BARRIER();
        MC_STAMP(4);
        if (L.ldead) return false;
        const double sst = L.lsc[0], alpha = L.lsc[1];
        // ---- 4. G(i): every wave for itself; r <- v = row_i * scal
        double tau, scal;
        larfg_vt(alpha, sst, tau, scal);
#define MC_VSCALE(J) r[J] *= scal;
        MC_FOR_LIVE(nlive, MC_VSCALE);
#undef MC_VSCALE
        if (lane == 0) {
            double* vrow = vws + (size_t)i * NS + gw;
            // (slot 0 = column 432 + gw, which exists for gw < 16 only)
#define MC_VSTORE(J) if (J > 0 || gw < NS - (CPW - 1) * NW) vrow[(CPW - 1 - J) * NW] = r[J];
            MC_FOR_LIVE(nlive, MC_VSTORE);
#undef MC_VSTORE
            if (gw == 0) taup[i] = tau;
        }
        if (i == m - 1) { MC_STAMP_FLUSH(dbg, (cu * 4 + wv) * 4 + SL); return true; }
        MC_STAMP(5);
        double tw[RS], nc[RS], w[RS];
#pragma unroll
        for (int s = SL; s < RS; ++s) {
            const int row = s * 64 + lane;
            const double q = L.lqt[row], c0 = L.lcol[row];
            const bool live = row > i;
            const double ws = live ? fma(scal, q, c0) : 0.0;        // w = A[:, i:] [1; v]  (rows > i)
            w[s] = ws;
            nc[s] = live ? fma(-tau, ws, c0) : 0.0;                 // column i after G(i)
            tw[s] = tau * ws;
        }
        // ---- 5. H(i) from column i, rows i+1 .. m-1
        const int i1 = i + 1, l1 = i1 & 63;
        const bool nxt = (i1 >> 6) != SL;                           // uniform: row i+1 opens the next slot
        double p1 = 0.0, p2 = 0.0;
#pragma unroll
        for (int s = SL; s < RS; ++s) {
            const bool in = s * 64 + lane > i1;
            p1 = fma(in ? nc[s] : 0.0, nc[s], p1);
            p2 = fma(in ? nc[s] : 0.0, w[s], p2);
        }
        p1 = wave_allsum(p1);
        p2 = wave_allsum(p2);
        constexpr int SN = (SL + 1 < RS) ? SL + 1 : SL;
        const double alpha2 = rdlane(nxt ? nc[SN] : nc[SL], l1);
        const double w1 = rdlane(nxt ? w[SN] : w[SL], l1);
        double tauq, sc2;
        larfg_vt(alpha2, p1, tauq, sc2);
        const double uw = fma(sc2, p2, w1);                         // u^T w
        double u[RS], tu[RS];
#pragma unroll
        for (int s = SL; s < RS; ++s) {
            const int row = s * 64 + lane;
            u[s] = (row <= i) ? 0.0 : ((row == i1) ? 1.0 : nc[s] * sc2);
            tu[s] = tauq * u[s];
        }
        MC_STAMP(6);
        // ---- 6. z_c = u^T A[:, c] for my live columns: per-lane partials, transposed through the wave's LDS tile
        double* tr = L.ltr[wv];
#define MC_ZPART(J)                                                                        \
        {                                                                                  \
            double zp_ = 0.0;                                                              \
            _Pragma("unroll") for (int s = SL; s < RS; ++s) zp_ = fma(u[s], a[J][s], zp_); \
            tr[lane * 17 + J] = zp_;                                                       \
        }
        MC_FOR_LIVE(nlive, MC_ZPART);
#undef MC_ZPART
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        double acc = 0.0;
        {
            const int jj = lane & 15, pg = lane >> 4;
#pragma unroll
            for (int k = 0; k < 16; ++k) acc += tr[(16 * pg + k) * 17 + jj];
        }
        acc = xrow_allsum(acc);                                     // lane l: z of local column l & 15
        __builtin_amdgcn_wave_barrier();
        MC_STAMP(7);
        // ---- 7. both rank-1 updates fused: A <- A - tau w [1; v]^T - tauq u z'^T,  z' = z - tau (u^T w) [1; v]
        const double tuw = tau * uw, ntuw = -tuw;
#define MC_UPDATE(J)                                                                       \
        {                                                                                  \
            /* z_J straight from the scalar pair the read-out leaves (the compiler moves it into a vector register */ \
            /* first to use the two-operand fmac: two v_mov per column) */                 \
            double zj_;                                                                    \
            {                                                                              \
                const double aj_ = rdlane(acc, J);                                         \
                asm("v_fma_f64 %0, %1, %2, %3" : "=v"(zj_) : "v"(ntuw), "v"(r[J]), "s"(aj_)); \
            }                                                                              \
            _Pragma("unroll") for (int s = SL; s < RS; ++s) a[J][s] = fma(-tu[s], zj_, fma(-tw[s], r[J], a[J][s])); \
        }
        MC_FOR_LIVE(nlive, MC_UPDATE);
#undef MC_UPDATE
        MC_STAMP(8);
    }
    MC_STAMP_FLUSH(dbg, (cu * 4 + wv) * 4 + SL);
    return true;
}

// Phi beside the bidiagonalisation (fused launch): Phi = the last N - m columns of P = G(0) G(1) ... G(m-1) accumulated
// FORWARD, where every ROW of P is independent: p <- p - tau (p . v) v^T as each reflector appears.  A consumer wave
// holds PHI_RW rows, follows the nine producers' progress words (one line, one load per poll) and reads the reflectors
// past its CU's L1.  PhiT[col][row] like k_mc_phi.
__device__ __forceinline__ unsigned wave_min_u32(unsigned v);        // (phase 3 below)
template <class LDS>
__device__ __forceinline__ void phi_rows(const double* __restrict__ vws, int N, int m, double* __restrict__ PhiT,
                                         void* comm, rsrc_t rs, LDS& L) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = N - m;
    const rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)vws, 0, (int)(((size_t)64 * RS_MAX * NS + 256) * sizeof(double)), 0x00020000);
    const unsigned tau_off = (unsigned)((size_t)64 * RS_MAX * NS * sizeof(double));
    unsigned* ticket = (unsigned*)((char*)comm + OFF_TICKET);
    constexpr int n_groups = (NS + 4 * PHI_RW - 1) / (4 * PHI_RW);
    for (;;) {
        __syncthreads();
        if (tid == 0) L.lgrp = (int)__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const int grp = L.lgrp;
        if (grp >= n_groups) return;
        const int r0 = (4 * grp + wv) * PHI_RW;
        double p[PHI_RW][NQ];
#pragma unroll
        for (int w = 0; w < PHI_RW; ++w)
#pragma unroll
            for (int q = 0; q < NQ; ++q) p[w][q] = (lane + 64 * q == r0 + w) ? 1.0 : 0.0;
        if (r0 < N) {
            int have = 0;
            bool failed = false;
            for (int i = 0; i < m && !failed; ++i) {
                unsigned spins = 0;
                while (have <= i) {
                    const unsigned pw = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs, OFF_PROG + 4u * (unsigned)min(lane, GMAX - 1), 0, 16);
                    have = (int)wave_min_u32(pw);
                    if (have > i) break;
                    if (spin_fail(rs, spins, 0x400u + (unsigned)i)) { failed = true; break; }
                    // (poll gently: the producers' exchange runs through the same L2 and is latency-bound, while a
                    //  consumer has ~3 us per reflector to spare)
                    __builtin_amdgcn_s_sleep(24);
                }
                if (failed) break;
                double v[NQ];
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const auto g2 = __builtin_amdgcn_raw_buffer_load_b64(rv, (unsigned)(i * NS + lane + 64 * q) * 8u, 0, 16);
                    const double x = __hiloint2double((int)g2[1], (int)g2[0]);
                    const int c = lane + 64 * q;
                    v[q] = (c > i && c < N) ? x : ((c == i) ? 1.0 : 0.0);
                }
                const auto gt2 = __builtin_amdgcn_raw_buffer_load_b64(rv, tau_off + (unsigned)i * 8u, 0, 16);
                const double tau = __hiloint2double((int)gt2[1], (int)gt2[0]);
                double d[PHI_RW];
#pragma unroll
                for (int w = 0; w < PHI_RW; ++w) {
                    d[w] = 0.0;
#pragma unroll
                    for (int q = 0; q < NQ; ++q) d[w] = fma(v[q], p[w][q], d[w]);
                }
#pragma unroll
                for (int w = 0; w < PHI_RW; ++w) {
                    const double t = tau * wave_allsum(d[w]);
#pragma unroll
                    for (int q = 0; q < NQ; ++q) p[w][q] = fma(-t, v[q], p[w][q]);
                }
            }
        }
#pragma unroll
        for (int w = 0; w < PHI_RW; ++w) {
            const int r = r0 + w;
            if (r >= NS) continue;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int col = lane + 64 * q - m;
                if (col >= 0 && col < K) PhiT[(size_t)col * NS + r] = (r < N) ? p[w][q] : 0.0;
            }
        }
        // this group of rows is in memory (a spin that gave up has set the error word): the pivot kernel refuses to run
        // on a Phi with a group missing
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
        if (tid == 0)
            __hip_atomic_fetch_add((unsigned*)((char*)comm + OFF_DONE), 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int RS>
__global__ __launch_bounds__(256) void k_mc_bidiag(const double* __restrict__ X, int ldx, int N, int m,
                                                   double* __restrict__ vws, double* __restrict__ taup, void* comm,
                                                   unsigned cbytes, unsigned long long* dbg, double* __restrict__ PhiT,
                                                   int fused) {
    __shared__ BidiagLds<RS> L;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(comm, 0, (int)cbytes, 0x00020000);
    if (tid == 0) L.ldead = fused ? elect_all(comm, OFF_EL_B, GMAX, rs) : elect(comm, OFF_EL_B, GMAX, rs);
    __syncthreads();
    const int cu = __builtin_amdgcn_readfirstlane(L.ldead);
    if (cu < 0) return;
    if (cu >= GMAX) {                                               // (fused launch) a consumer: rows of Phi
        phi_rows(vws, N, m, PhiT, comm, rs, L);
        return;
    }
    __syncthreads();
    const int gw = cu * 4 + wv;
    double a[CPW][RS], r[CPW];
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
        r[j] = 0.0;
#pragma unroll
        for (int s = 0; s < RS; ++s) {
            const int row = s * 64 + lane, c = (CPW - 1 - j) * NW + gw;      // reverse order: bidiag_steps
            a[j][s] = (c < N && row < m) ? ((row == 0) ? 1.0 : X[(size_t)c * ldx + (row - 1)]) : 0.0;
        }
    }
    if (tid == 0) L.ldead = 0;
    __syncthreads();
    bool ok = bidiag_steps<RS, 0>(a, r, m, gw, cu, wv, L, rs, vws, taup, dbg);
    if constexpr (RS > 1) { if (ok && m > 64) ok = bidiag_steps<RS, 1>(a, r, m, gw, cu, wv, L, rs, vws, taup, dbg); }
    if constexpr (RS > 2) { if (ok && m > 128) ok = bidiag_steps<RS, 2>(a, r, m, gw, cu, wv, L, rs, vws, taup, dbg); }
    if constexpr (RS > 3) { if (ok && m > 192) ok = bidiag_steps<RS, 3>(a, r, m, gw, cu, wv, L, rs, vws, taup, dbg); }
    // the last reflector (this time the wait for the stores is a real one)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    if (ok && tid == 0) __builtin_amdgcn_raw_buffer_store_b32((unsigned)m, rs, OFF_PROG + 4u * (unsigned)cu, 0, 0);
}

// ================================================================================================================
// phase 2: Phi = G(0) ... G(m-1) [0; I], one wave per column; PhiT[col][row] (a column is contiguous)
// ================================================================================================================
__global__ __launch_bounds__(256) void k_mc_phi(const double* __restrict__ vws, const double* __restrict__ taup, int N,
                                                int m, double* __restrict__ PhiT, double* __restrict__ phi_out) {
    const int lane = threadIdx.x & 63;
    const int col = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int K = N - m;
    if (col >= K) return;                                           // wave-uniform
    double phi[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) phi[q] = (lane + 64 * q == m + col) ? 1.0 : 0.0;
    // reflector i: zero below i, one at i, the stored entries above; three register buffers rotate so that an L2
    // round trip is always two steps ahead
#define MC_LOAD(BUF, TAU, I)                                                               \
    {                                                                                      \
        const int ii_ = max((I), 0);                                                       \
        const double* src_ = vws + (size_t)ii_ * NS;                                       \
        _Pragma("unroll") for (int q = 0; q < NQ; ++q) BUF[q] = src_[lane + 64 * q];       \
        TAU = taup[ii_];                                                                   \
    }
#define MC_APPLY(V, TAU, I)                                                                \
    if ((I) >= 0) {                                                                        \
        double v_[NQ];                                                                     \
        double d_ = 0.0;                                                                   \
        _Pragma("unroll") for (int q = 0; q < NQ; ++q) {                                   \
            const int row_ = lane + 64 * q;                                                \
            v_[q] = (row_ > (I) && row_ < N) ? V[q] : ((row_ == (I)) ? 1.0 : 0.0);         \
            d_ = fma(v_[q], phi[q], d_);                                                   \
        }                                                                                  \
        const double t_ = TAU * wave_allsum(d_);                                           \
        _Pragma("unroll") for (int q = 0; q < NQ; ++q) phi[q] = fma(-t_, v_[q], phi[q]);   \
    }
    double vA[NQ], vB[NQ], vC[NQ];
    double tA = 0.0, tB = 0.0, tC = 0.0;
    MC_LOAD(vA, tA, m - 1)
    MC_LOAD(vB, tB, m - 2)
    for (int i = m - 1; i >= 0; i -= 3) {
        MC_LOAD(vC, tC, i - 2)
        MC_APPLY(vA, tA, i)
        MC_LOAD(vA, tA, i - 3)
        MC_APPLY(vB, tB, i - 1)
        MC_LOAD(vB, tB, i - 4)
        MC_APPLY(vC, tC, i - 2)
    }
#undef MC_LOAD
#undef MC_APPLY
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int row = lane + 64 * q;
        PhiT[(size_t)col * NS + row] = (row < N) ? phi[q] : 0.0;
        if (phi_out != nullptr && row < N) phi_out[(size_t)row * K + col] = phi[q];
    }
}

// ================================================================================================================
// phase 3: the pivots
// ================================================================================================================
// order-preserving map double -> uint64 for the ratio test: NaN -> 0 (torch.argmin lets a NaN win)
__device__ __forceinline__ unsigned long long ratio_key(double x) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    const unsigned long long k = (b >> 63) ? ~b : (b | 0x8000000000000000ull);
    return (x != x) ? 0ull : k;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_min_u32(unsigned v) {
    const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, CTRL, ROW_MASK, 0xf, false);   // (`old` = the identity: folds into v_min_u32_dpp)
    return o < v ? o : v;
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {        // uniform result
    v = dpp_min_u32<ROR8, 0xf>(v);
    v = dpp_min_u32<ROR4, 0xf>(v);
    v = dpp_min_u32<ROR2, 0xf>(v);
    v = dpp_min_u32<ROR1, 0xf>(v);
    v = dpp_min_u32<BCAST15, 0xa>(v);
    v = dpp_min_u32<BCAST31, 0xc>(v);
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// alive[q]: rows 64 q + lane inside the step and not cancelled (Phi[idx, :] = 0 of :266: readers go through the mask) --
// wave-uniform lane masks in scalar registers, used directly as the selects' conditions (csrc/car.hip, SpState)
struct PivState {
    double mu[NQ];
    unsigned long long alive[NQ];
};
__device__ __forceinline__ bool alive_lane(const PivState& st, int q) { return __builtin_amdgcn_inverse_ballot_w64(st.alive[q]); }
constexpr int PV_BAND = 4;          // screened ratio test: candidates = high words within PV_BAND of the minimum

// ratio test of :239-247 on column `col`: the row of the first argmin of mu / col over col > 0 (a NaN quotient wins);
// -1: none.  Screened form first (csrc/car.hip, sp_ratio_test: the same argument and the same test of the seed's
// accuracy): high words of mu * v_rcp_f64(col); a lone candidate within PV_BAND steps of the minimum IS the exact
// argmin; anything else -- two candidates, a negative / infinite / NaN quotient, none -- takes the exact test.
__device__ __forceinline__ int ratio_test(const double (&col)[NQ], const PivState& st, bool exact_only) {
    if (!exact_only) {
        unsigned sk[NQ];
        unsigned smin = 0xffffffffu;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const double r = st.mu[q] * __builtin_amdgcn_rcp(col[q]);
            const unsigned k = (unsigned)__double2hiint(r) + 0x80100000u;
            sk[q] = (col[q] > 0.0) ? k : 0xffffffffu;                 // (col is 0 on rows that are not alive)
            smin = min(smin, sk[q]);
        }
        const unsigned H = wave_min_u32(smin);
        if (__builtin_expect(H >= 0x80100000u && H < 0xffffffffu - (unsigned)PV_BAND, 1)) {
            unsigned long long any = 0ull;
            int cnt = 0, kp = 0;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const unsigned long long mb = __ballot(sk[q] <= H + (unsigned)PV_BAND);
                cnt += __popcll(mb);
                any |= mb;
                kp += (mb != 0ull) ? q : 0;
            }
            if (__builtin_expect(cnt == 1, 1)) return __ffsll((long long)any) - 1 + 64 * kp;   // one bit in all: no branch
        }
    }
    double rt[NQ];
    unsigned kh[NQ], kl[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) rt[q] = st.mu[q] / col[q];
    unsigned hmin = 0xffffffffu;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const unsigned long long k = ratio_key(rt[q]);
        const bool ok = col[q] > 0.0;
        kh[q] = ok ? (unsigned)(k >> 32) : 0xffffffffu;
        kl[q] = ok ? (unsigned)k : 0xffffffffu;
        hmin = min(hmin, kh[q]);
    }
    const unsigned H = wave_min_u32(hmin);
    if (H == 0xffffffffu) return -1;                                  // uniform: no candidate (:241-242)
    unsigned long long mb[NQ];
    int cnt = 0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) { mb[q] = __ballot(kh[q] == H); cnt += __popcll(mb[q]); }
    if (cnt != 1) {                                                   // rare: several quotients share the high word
        unsigned lmin = 0xffffffffu;
#pragma unroll
        for (int q = 0; q < NQ; ++q) lmin = min(lmin, (kh[q] == H) ? kl[q] : 0xffffffffu);
        const unsigned Lw = wave_min_u32(lmin);
#pragma unroll
        for (int q = 0; q < NQ; ++q) mb[q] = __ballot((kh[q] == H) & (kl[q] == Lw));
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q)
        if (mb[q] != 0ull) return __ffsll((long long)mb[q]) - 1 + 64 * q;     // uniform
    return -1;
}

// mu[:] = mu - alpha * Phi[:, 0]; mu[idx] = 0  (:253-254; one fma under -ffp-contract=fast, like csrc/car.hip)
__device__ __forceinline__ void mu_step(PivState& st, const double (&col)[NQ], double alpha, int piv, int lane) {
    const unsigned long long bit = 1ull << (piv & 63);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        st.alive[q] &= ~((piv >> 6) == q ? bit : 0ull);               // scalar
        st.mu[q] = __dsub_rn(st.mu[q], __dmul_rn(alpha, col[q]));   // (no select: csrc/car.hip sp_mu_step; the output applies the mask)
    }
}

// what a pivot needs from row `piv`: its entries of my columns J0 .. BC-1 (and, for the wave that found it, the row's
// weight and its entry of the pivot column) -- ONE uniform switch over the row's slot that only READS; the updates
// follow behind it (inside the cases the compiler keeps a renamed copy of phi per case and moves it back)
template <int KP, int J0, bool OWN>
__device__ __forceinline__ void pivot_row_kp(const double (&phi)[BC][NQ], const PivState& st, const double (&col)[NQ], int lp,
                                             double (&prow)[BC], double& mp, double& cp) {
    if constexpr (OWN) { mp = rdlane(st.mu[KP], lp); cp = rdlane(col[KP], lp); }
#pragma unroll
    for (int j = J0; j < BC; ++j) prow[j] = rdlane(phi[j][KP], lp);
}
template <int J0, bool OWN>
__device__ __forceinline__ void pivot_row(const double (&phi)[BC][NQ], const PivState& st, const double (&col)[NQ], int piv,
                                          double (&prow)[BC], double& mp, double& cp) {
    const int kp = piv >> 6, lp = piv & 63;
    switch (kp) {                                                     // uniform
        case 0: pivot_row_kp<0, J0, OWN>(phi, st, col, lp, prow, mp, cp); break;
        case 1: pivot_row_kp<1, J0, OWN>(phi, st, col, lp, prow, mp, cp); break;
        case 2: pivot_row_kp<2, J0, OWN>(phi, st, col, lp, prow, mp, cp); break;
        case 3: pivot_row_kp<3, J0, OWN>(phi, st, col, lp, prow, mp, cp); break;
        case 4: pivot_row_kp<4, J0, OWN>(phi, st, col, lp, prow, mp, cp); break;
        case 5: pivot_row_kp<5, J0, OWN>(phi, st, col, lp, prow, mp, cp); break;
        default: pivot_row_kp<6, J0, OWN>(phi, st, col, lp, prow, mp, cp); break;
    }
}
// rank-1 elimination of my columns J0 .. BC-1 with the pivot (col, piv, rpp):
//   Phi[:, c] -= Phi[:, 0] * (Phi[idx, c] / Phi[idx, 0])   (:260-266)
template <int J0>
__device__ __forceinline__ void elim_rows(double (&phi)[BC][NQ], const double (&col)[NQ], const double (&prow)[BC], double rpp) {
#pragma unroll
    for (int j = J0; j < BC; ++j) {
        const double qv = prow[j] * rpp;
#pragma unroll
        for (int q = 0; q < NQ; ++q) phi[j][q] = fma(-qv, col[q], phi[j][q]);
    }
}
template <int J0>
__device__ __forceinline__ void elim(double (&phi)[BC][NQ], const PivState& st, const double (&col)[NQ], int piv, double rpp) {
    double prow[BC], mp, cp;
    pivot_row<J0, false>(phi, st, col, piv, prow, mp, cp);
    elim_rows<J0>(phi, col, prow, rpp);
}

// wait for pivot s: header (alpha, 1/pivot, index) and this lane's rows of the pivot column; false = give up
__device__ __forceinline__ bool consume(rsrc_t rs, int s, int lane, double (&col)[NQ], double& alpha, double& rpp,
                                        int& piv) {
    const unsigned tag = (unsigned)s + 1u;
    const unsigned hoff = OFF_H + (unsigned)s * 64u;
    const unsigned coff = OFF_P + ((unsigned)s * NS + (unsigned)lane) * 16u;
    unsigned spins = 0;
    for (;;) {                                                        // light spin: the index word (stored last); 31
        const u32x4 g = load_granule(rs, hoff + 32u);                 // waves polling whole columns slow the producer
        if (__all(granule_ok(g, tag))) break;
        if (spin_fail(rs, spins, 0x400u + (unsigned)s)) return false;
        asm volatile("" ::: "memory");
    }
    for (;;) {
        u32x4 h0 = load_granule(rs, hoff), h1 = load_granule(rs, hoff + 16u), h2 = load_granule(rs, hoff + 32u);
        u32x4 gc[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) gc[q] = load_granule(rs, coff + (unsigned)q * 1024u);
        const bool okh = granule_ok(h0, tag) && granule_ok(h1, tag) && granule_ok(h2, tag);
        alpha = granule_val(h0); rpp = granule_val(h1);
        piv = okh ? (int)granule_val(h2) : 0;                          // (the same in every lane; made scalar below)
        bool ok = okh;
#pragma unroll
        for (int q = 0; q < NQ; ++q) { ok &= granule_ok(gc[q], tag) | (piv < 0); col[q] = granule_val(gc[q]); }
        if (__all(ok)) { piv = __builtin_amdgcn_readfirstlane(piv); return true; }
        if (spin_fail(rs, spins, 0x500u + (unsigned)s)) return false;
        asm volatile("" ::: "memory");
    }
}

__device__ __forceinline__ void publish(rsrc_t rs, int s, int lane, const double (&col)[NQ], double alpha, double rpp,
                                        int piv) {
    const unsigned tag = (unsigned)s + 1u;
    const unsigned hoff = OFF_H + (unsigned)s * 64u;
    if (piv >= 0) {
        const unsigned coff = OFF_P + ((unsigned)s * NS + (unsigned)lane) * 16u;
#pragma unroll
        for (int q = 0; q < NQ; ++q) put_granule(rs, coff + (unsigned)q * 1024u, col[q], tag);
    }
    if (lane == 0) {
        put_granule(rs, hoff, alpha, tag);
        put_granule(rs, hoff + 16u, rpp, tag);
        put_granule(rs, hoff + 32u, (double)piv, tag);
    }
}

// column JJ of my block is the pivot column: ratio test, the winner's row read once, publish, update of the weights and
// of my later columns.  One instance per column (the columns keep their registers): the loop form of rounds 2-4 moved
// the seven later columns down by one slot behind every pivot -- 49 v_mov_b64 on the producing wave's chain.
#ifdef MC_STAMPS
#define MC_PSTAMP_ARGS , unsigned long long (&acc_)[12], unsigned long long& tl_
#define MC_PSTAMP_PASS , acc_, tl_
#else
#define MC_PSTAMP_ARGS
#define MC_PSTAMP_PASS
#endif
#ifdef MC_TSTAMPS      // diagnostic build (scripts/pivot_stamps.py mc): s_memrealtime (100 MHz) of every publish, by pivot index
__device__ unsigned long long g_mc_stamps[260];
#endif
template <int JJ>
__device__ __forceinline__ bool produce(double (&phi)[BC][NQ], PivState& st, rsrc_t rs, int s, int lane, bool exact_only MC_PSTAMP_ARGS) {
    double col[NQ];
    MC_STAMP(0);
#pragma unroll
    for (int q = 0; q < NQ; ++q) col[q] = alive_lane(st, q) ? phi[JJ][q] : 0.0;
    const int piv = ratio_test(col, st, exact_only);
    double prow[BC], mp = 0.0, cp = 1.0;
    if (piv >= 0) pivot_row<JJ + 1, true>(phi, st, col, piv, prow, mp, cp);
    const double al = mp / cp, rp = 1.0 / cp;                         // (alpha and 1 / pivot: the same two divisions whichever test found the row)
    MC_STAMP(1);
    publish(rs, s, lane, col, al, rp, piv);
#ifdef MC_TSTAMPS
    { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); if (lane == 0 && s < 256) g_mc_stamps[s] = t_; }
#endif
    MC_STAMP(2);
    if (piv < 0) return false;                                        // Q6: the loop ends here (:241-242)
    mu_step(st, col, al, piv, lane);
    elim_rows<JJ + 1>(phi, col, prow, rp);
    MC_STAMP(3);
    return true;
}

__global__ __launch_bounds__(256) void k_mc_pivot(const double* __restrict__ PhiT, int N, int m,
                                                  const double* __restrict__ mu_in, int32_t* __restrict__ keep_rank,
                                                  double* __restrict__ w_star, int32_t* __restrict__ n_keep_out,
                                                  double* __restrict__ mu_out, void* comm, unsigned cbytes,
                                                  unsigned long long* dbg, int phi_groups, int exact_ratio) {
    __shared__ int lcu;
    const int lane = threadIdx.x & 63;
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(comm, 0, (int)cbytes, 0x00020000);
    if (threadIdx.x == 0) lcu = elect(comm, OFF_EL_P, PWAVES / 4, rs);
    __syncthreads();
    const int cu = __builtin_amdgcn_readfirstlane(lcu);
    if (cu < 0) return;
    const int gw = cu * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
#ifdef MC_TSTAMPS
    if (gw == 0 && lane == 0) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); g_mc_stamps[256] = t_; }
#endif
    const int K = N - m;
    const int c0 = gw * BC;
    if (c0 >= K && gw != 0) return;
    double phi[BC][NQ];
    PivState st;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int row = lane + 64 * q;
        const int nq = min(max(N - 64 * q, 0), 64);
        st.alive[q] = nq >= 64 ? ~0ull : ((1ull << nq) - 1ull);
        st.mu[q] = row < N ? mu_in[row] + 0.0 : 0.0;
#pragma unroll
        for (int j = 0; j < BC; ++j) phi[j][q] = (c0 + j < K && row < N) ? PhiT[(size_t)(c0 + j) * NS + row] : 0.0;
    }
    bool fail = load_err(rs) != 0u;                                   // the bidiagonalisation gave up
    if (phi_groups > 0)                                               // (fused launch) ... or a group of Phi's rows found no consumer
        fail |= (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs, OFF_DONE, 0, 16) != (unsigned)phi_groups;
    bool stop = false;
    MC_STAMP_DECL
    double col[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) col[q] = 0.0;
    // the pivots before my block: apply them to all my columns as they arrive
    for (int s = 0; s < min(c0, K) && !fail && !stop; ++s) {
        double al, rp;
        int piv;
        MC_STAMP(4);
        if (!consume(rs, s, lane, col, al, rp, piv)) { fail = true; break; }
        if (s + 1 == min(c0, K)) MC_STAMP(6); else MC_STAMP(5);       // 6: the wait for the hand-over
        if (piv < 0) { stop = true; break; }
        mu_step(st, col, al, piv, lane);
        elim<0>(phi, st, col, piv, rp);
        MC_STAMP(7);
    }
    // my block
    if (!fail && !stop && c0 < K) {
        bool go = true;
        const int s_end = min(c0 + BC, K);
        int sp = c0;
#define MC_STEP(JJ) if (sp < s_end && go) { go = produce<JJ>(phi, st, rs, sp, lane, exact_ratio != 0 MC_PSTAMP_PASS); ++sp; }
        MC_STEP(0) MC_STEP(1) MC_STEP(2) MC_STEP(3) MC_STEP(4) MC_STEP(5) MC_STEP(6) MC_STEP(7)
#undef MC_STEP
        static_assert(BC == 8, "one MC_STEP per column of a block");
        stop = !go;
    }
    MC_STAMP_FLUSH(dbg, 256 + gw);
    if (gw != 0) return;
    // wave 0 follows the remaining pivots for the weights and writes the result
    for (int s = c0 + BC; s < K && !fail && !stop; ++s) {
        double al, rp;
        int piv;
        if (!consume(rs, s, lane, col, al, rp, piv)) { fail = true; break; }
        if (piv < 0) { stop = true; break; }
        mu_step(st, col, al, piv, lane);
    }
    // ---------------- output: w_star = mu[mu > 0], idx_star as ranks ----------------
    int base = 0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int row = lane + 64 * q;
        const double v = alive_lane(st, q) ? st.mu[q] + 0.0 : 0.0;    // mu[idx] = 0 of :254 for every cancelled row
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
#ifdef MC_TSTAMPS
    if (lane == 0) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); g_mc_stamps[257] = t_; }
#endif
}

// lane-swap self test: out[l] = wave_allsum(in[l]), out[64 + l] = xrow_allsum(in[l])
__global__ void k_mc_report_giveup(int32_t* __restrict__ n_keep) { if (threadIdx.x == 0) *n_keep = -1; }
__global__ void k_mc_selftest(const double* __restrict__ in, double* __restrict__ out) {
    const int l = threadIdx.x;
    out[l] = wave_allsum(in[l]);
    out[64 + l] = xrow_allsum(in[l]);
}

}  // namespace mc
}  // namespace sober

extern "C" int sober_car_giveup_forced(void);
extern "C" int sober_car_mc_supported(int N, int m) {
    using namespace sober::mc;
    return (m >= 2 && N > m && N <= NS && m <= 64 * RS_MAX && N - m <= KMAX) ? 1 : 0;
}

// scratch: [communication area, zeroed per call] [reflectors m x 448] [tau, 256] [Phi^T (N - m) x 448]
extern "C" int64_t sober_car_mc_ws_bytes(int N, int m) {
    using namespace sober::mc;
    (void)N; (void)m;
    const int64_t comm = ((int64_t)comm_bytes(KMAX) + 255) / 256 * 256;
    return comm + ((int64_t)64 * RS_MAX * NS + 256 + (int64_t)KMAX * NS + DBG_WORDS) * (int64_t)sizeof(double);
}

extern "C" int sober_car_mc_device(const double* X, int ldx, int N, int m, const double* mu_in, int32_t* keep_rank,
                                   double* w_star, int32_t* n_keep, double* mu_out, double* phi_out, void* ws,
                                   int64_t ws_bytes, void* stream) {
    using namespace sober::mc;
    if (!X || !mu_in || !keep_rank || !w_star || !n_keep || !mu_out || !ws || ldx < m - 1) return SOBER_E_ARG;
    if (!sober_car_mc_supported(N, m)) return SOBER_E_DIM;
    if (ws_bytes < sober_car_mc_ws_bytes(N, m)) return SOBER_E_WS;
    hipStream_t st = (hipStream_t)stream;
    const int K = N - m;
    const unsigned cbytes = comm_bytes(K);
    const int64_t comm_pad = ((int64_t)comm_bytes(KMAX) + 255) / 256 * 256;
    char* base = (char*)ws;
    double* vws = (double*)(base + comm_pad);
    double* taup = vws + (size_t)64 * RS_MAX * NS;
    double* PhiT = taup + 256;
    unsigned long long* dbg = (unsigned long long*)(PhiT + (size_t)KMAX * NS);
    HIP_TRY(hipMemsetAsync(ws, 0, (size_t)cbytes, st));
    const bool unfused = sober::switches().car_unfused;                                // (A/B switch)
    const int fused = (phi_out == nullptr && !unfused) ? 1 : 0;
    const int G = fused ? FUSED_GRID : ELECT_GRID;
    const size_t pad = fused ? (size_t)FUSED_LDS_PAD : 0;             // (dynamic LDS nobody touches: one workgroup per CU)
    static std::atomic<unsigned long long> attr_set{0};             // (one bit per device)
    if (sober_attr_needed(attr_set)) {
        HIP_TRY(hipFuncSetAttribute((const void*)k_mc_bidiag<2>, hipFuncAttributeMaxDynamicSharedMemorySize, FUSED_LDS_PAD));
        HIP_TRY(hipFuncSetAttribute((const void*)k_mc_bidiag<3>, hipFuncAttributeMaxDynamicSharedMemorySize, FUSED_LDS_PAD));
        HIP_TRY(hipFuncSetAttribute((const void*)k_mc_bidiag<4>, hipFuncAttributeMaxDynamicSharedMemorySize, FUSED_LDS_PAD));
        sober_attr_done(attr_set);
    }
    const int RS = (m + 63) / 64;
    switch (RS) {
        case 1:
        case 2:
            hipLaunchKernelGGL(k_mc_bidiag<2>, dim3(G), dim3(256), pad, st, X, ldx, N, m, vws, taup, ws, cbytes, dbg, PhiT, fused);
            break;
        case 3:
            hipLaunchKernelGGL(k_mc_bidiag<3>, dim3(G), dim3(256), pad, st, X, ldx, N, m, vws, taup, ws, cbytes, dbg, PhiT, fused);
            break;
        default:
            hipLaunchKernelGGL(k_mc_bidiag<4>, dim3(G), dim3(256), pad, st, X, ldx, N, m, vws, taup, ws, cbytes, dbg, PhiT, fused);
            break;
    }
    LAUNCH_CHECK();
    if (!fused) {
        hipLaunchKernelGGL(k_mc_phi, dim3((K + 3) / 4), dim3(256), 0, st, vws, taup, N, m, PhiT, phi_out);
        LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_mc_pivot, dim3(ELECT_GRID), dim3(256), 0, st, PhiT, N, m, mu_in, keep_rank, w_star, n_keep,
                       mu_out, ws, cbytes, dbg, fused ? (NS + 4 * PHI_RW - 1) / (4 * PHI_RW) : 0, (int)sober::switches().car_exact_ratio);
    LAUNCH_CHECK();
    if (sober_car_giveup_forced()) {                        // (test switch: what a give-up of the exchange reports)
        hipLaunchKernelGGL(k_mc_report_giveup, dim3(1), dim3(64), 0, st, n_keep);
        LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int sober_mc_selftest(const double* in, double* out, void* stream) {
    if (!in || !out) return SOBER_E_ARG;
    hipLaunchKernelGGL(sober::mc::k_mc_selftest, dim3(1), dim3(64), 0, (hipStream_t)stream, in, out);
    LAUNCH_CHECK();
    return 0;
}

#ifdef MC_TSTAMPS
extern "C" int sober_debug_mc_stamps(unsigned long long* out260) {
    return (int)hipMemcpyFromSymbol(out260, HIP_SYMBOL(sober::mc::g_mc_stamps), sizeof(unsigned long long) * 260, 0, hipMemcpyDeviceToHost);
}
#endif
