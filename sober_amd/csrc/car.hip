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
// recomputed here.  Launches on one stream, each shaped by what bounds its phase:
//
//   k_car_bidiag_fused  workgroup 0 (256 threads, one wave per SIMD) is the bidiagonalisation: the matrix lives in
//                 VGPRs (7 x 13 doubles per thread), a matrix row inside ONE 16-lane DPP row so that row dot
//                 products never leave the wave; two workgroup barriers per step (merged form, see
//                 car_bidiag2_block).  The other workgroups on its XCD accumulate the rows of Phi = P [0; I] as the
//                 reflectors appear (forward accumulation, row by row independent).
//   k_car_bidiag + k_car_phi  the same two phases as separate launches with NO dependence between workgroups
//                 (backward accumulation, one wave per column of Phi): the test hook for phi_out and the route a
//                 fused launch that gave up is redone on (sober_car_device_ex, SOBER_CAR_SAFE).
//   k_car_pivot_stream  1 workgroup x 16 waves, Phi in VGPRs: the N-m pivots of :237-266 as a barrier-free stream
//                 through an LDS ring.
//
// Limits of these one-CU kernels: N <= 208, m <= 112, N - m <= 112 (batch <= 100).  sober_car_device hands larger
// steps (N <= 448, m <= 256: batch <= 224) to the multi-CU kernels of car_mc.hip, steps beyond those (N <= 2048) to the
// memory-resident kernels of car_big.hip; only then the engine takes the host LAPACK route (sober_car_pivot_host).
#include "common.hpp"
#include <atomic>
#include <cstdlib>

namespace sober {

constexpr int CAR_MS = 7;           // matrix row slots per thread in the bidiagonalisation: m <= 16 * 7
constexpr int CAR_CQ = 13;          // 16-lane slots along N: N <= 16 * 13
constexpr int CAR_NS = 16 * CAR_CQ; // stride of a reflector / of the LDS columns
constexpr int CAR_PC = 128;         // columns of Phi in the global scratch (N - m <= 128), kept COLUMN by column: Phi[col * CAR_NS + row]
//                                    (a pivot wave reads its seven columns as contiguous runs of rows)
constexpr int CAR_BT = 256;         // threads of k_car_bidiag

// ---- DPP cross-lane helpers (row = 16 lanes).  ds_bpermute-based __shfl costs an LDS round trip
// per step; these are plain VALU moves.
template <int CTRL>
__device__ __forceinline__ double dpp(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);          // (no `old` operand: every lane is written)
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
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

// v_rcp_f64 seed + Newton steps (<= 1 ulp) instead of an IEEE division: a third of its dependent chain, no branch
__device__ __forceinline__ double car_rcp(double c) {
    double r = __builtin_amdgcn_rcp(c);
    double e = fma(-c, r, 1.0);
    r = fma(r, e, r);
    e = fma(-c, r, 1.0);
    return fma(r, e, r);
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
typedef unsigned int car_u32x2 __attribute__((ext_vector_type(2)));
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

// ---------------- phase 1, merged form (round 3): two barriers per step, nobody waits for an owner ----------------
// The step above is a relay: the owner row builds G(i) while three waves wait (A), everyone applies it (B), every wave
// builds H(i) from the published column (C), the column sums are finished (D), H(i) is applied -- four barriers, and the
// owner's 130 dependent instructions are paid by all.  Here every thread rebuilds what it needs from what the PREVIOUS
// step left in LDS, the same reflectors in exact arithmetic (tests/test_car_algorithm.py::bidiag_merged):
//   * row i+1 and column i+1 of A' = A G(i) are published by their holders right after G(i)'s update (rowg, colg);
//     the next step's row x = rowg - tauq z and column cur = colg - f z_i follow from them in every thread, so G(i+1)'s
//     scalars (dlarfg on x) are computed redundantly by all 256 threads -- phase (A) and its barrier are gone;
//   * column i after G(i), col' = cur - tau w, is known to every thread for its own rows, so the column sums
//     Y = col'[i+2:]^T A' and |col'[i+2:]|^2 start without H(i)'s scalars; H(i)'s dlarfg runs AFTER the sums, redundantly,
//     at the top of the next iteration, where z = rowg + sc2 Y -- phase (C)'s wave-wide norm and its barrier are gone.
// One iteration = [z, H(i-1)'s update | row, column, G(i), w = A v, update, publish, partial column sums]
// barrier [16 partial sums -> Y on waves 0-2; norm + H(i)'s scalars on wave 3] barrier.  Dead rows and columns are never written (masks on the block's own slot only).
struct CarLds2 {
    double* rowg; double* colg; double* zpart; double* s2part; double* zsum; double* scal; double* hsc;
};

// timing-only switches (wrong results): what each part of the step costs  (scripts/bidiag_where.sh)
#define CB2_RSUM(v) row16_sum(v)
#define CB2_HUPD(n, o) (n)
#define CB2_GUPD(n, o) (n)
#define CB2_BARRIER() CAR_LDS_BARRIER()
#define CB2_VSTORE(e) e
#ifdef CAR_BSTAMPS
#define CB2_STAMP(K, VAL) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_), "+v"(VAL) :: "memory"); \
    cacc_[K] += t_ - ctl_; ctl_ = t_; } while (0)
#else
#define CB2_STAMP(K, VAL) do { } while (0)
#endif
// MS / CQ: row / column slots in use (m <= 16 MS, N <= 16 CQ): the instantiation is picked by the problem's size, so
// a 20 x 10 step (batch 10) does not carry the 7 x 13 slots of a 200 x 100 one through every loop
template <int S, bool FUSED, int MS, int CQ>
__device__ __forceinline__ void car_bidiag2_block(double (&a)[CAR_MS][CAR_CQ], double (&colp)[CAR_MS], int m, const CarLds2& L,
                                                  double* __restrict__ vws, double* __restrict__ taup, const CarPub& pub) {
    const int tid = threadIdx.x;
    const int R = tid >> 4, C = tid & 15;
    const int i_end = min(16 * S + 16, m - 1);
    constexpr int S1 = (S + 1 < MS) ? S + 1 : S;         // row slot of row i+1 when i is the block's last step
    const car_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(vws, 0, (int)((size_t)m * CAR_NS * sizeof(double)), 0x00020000);
    CB_DECL
    for (int i = 16 * S; i < i_end; ++i) {
        const int li = i & 15, p = i & 1, pp = p ^ 1;
        if constexpr (FUSED) car_publish_progress(pub, i);   // reflector i - 1 is complete in memory by now
        // ---- H(i-1): scalars from the finished sums, z, the rank-1 update (rows >= i, columns >= i)
        double z[CAR_CQ], x[CAR_CQ], f[CAR_MS], cur[CAR_MS];
        CB_STAMP(0);
        const double tauq = L.hsc[0], sc2 = L.hsc[1];        // H(i-1)'s scalars (wave 3 of the previous step's sums)
#pragma unroll
        for (int q = S; q < CQ; ++q) { x[q] = L.rowg[pp * CAR_NS + C + 16 * q]; z[q] = L.zsum[C + 16 * q]; }
        const double rgi = L.rowg[pp * CAR_NS + i];
#pragma unroll
        for (int k = S; k < MS; ++k) cur[k] = L.colg[pp * 128 + R + 16 * k];
#pragma unroll
        for (int q = S; q < CQ; ++q) {
            z[q] = fma(sc2, z[q], x[q]);
            x[q] = fma(-tauq, z[q], x[q]);                  // row i of the updated matrix (u_i = 1)
        }
        z[S] = (C >= li) ? z[S] : 0.0;
        x[S] = (C > li) ? x[S] : 0.0;
#pragma unroll
        for (int k = S; k < MS; ++k) f[k] = tauq * (sc2 * colp[k]);
        f[S] = (R == li) ? tauq : f[S];
        const double zi = rdlane(z[S], li);                 // z at column i (lane li of every wave holds C == li)
        const double alpha = fma(-tauq, zi, rgi);
#pragma unroll
        for (int k = S; k < MS; ++k) cur[k] = fma(-f[k], zi, cur[k]);                            // column i, my rows
#pragma unroll
        for (int q = S; q < CQ; ++q)
#pragma unroll
            for (int k = S; k < MS; ++k) a[k][q] = CB2_HUPD(fma(-f[k], z[q], a[k][q]), a[k][q]);
        // ---- G(i) from row i
        double ss0 = 0.0, ss1 = 0.0;
#pragma unroll
        for (int q = S; q < CQ; ++q) { if (q & 1) ss1 = fma(x[q], x[q], ss1); else ss0 = fma(x[q], x[q], ss0); }
        double ss = CB2_RSUM(ss0 + ss1);
        CB2_STAMP(2, ss);
        double beta, tau, sc;
        larfg_vt(alpha, ss, tau, sc); (void)beta;
        CB2_STAMP(3, tau);
        // w = A v = column i + sc * (A x) over my rows
        double tG[CAR_MS];
#pragma unroll
        for (int k = S; k < MS; ++k) {
            double w0 = 0.0, w1 = 0.0;
#pragma unroll
            for (int q = S; q < CQ; ++q) { if (q & 1) w1 = fma(a[k][q], x[q], w1); else w0 = fma(a[k][q], x[q], w0); }
            tG[k] = tau * fma(sc, CB2_RSUM(w0 + w1), cur[k]);
        }
        tG[S] = (R > li) ? tG[S] : 0.0;                      // rows <= i stay
        CB2_STAMP(4, tG[S]);
#pragma unroll
        for (int k = S; k < MS; ++k) colp[k] = cur[k] - tG[k];      // column i after G(i)  (v_i = 1)
        colp[S] = (R > li) ? colp[S] : 0.0;
#pragma unroll
        for (int q = S; q < CQ; ++q) x[q] *= sc;         // x becomes v
        if (R == li) {                                       // the reflector goes to the scratch (Phi follows it)
            // (round 6, from the ISA -- profiles/r06_bidiag_isa.txt: in block 0 li == i, so inside this branch the compiler
            //  knows i == R and rewrote the scalar offset below in terms of the LANE's R: thirteen waterfall loops
            //  (readfirstlane / compare / saveexec / store / branch) per step.  The step index is laundered as what it is, a
            //  wave-uniform value; and the lane's byte offset is made opaque to the known-bits analysis so that 128 q stays an
            //  ADD and folds into the store's immediate offset: one address register instead of thirteen kept alive
            //  through the whole kernel.)
            const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane(i) * (unsigned)(CAR_NS * 8);
            unsigned c8 = (unsigned)C * 8u;
            asm volatile("" : "+v"(c8));
#pragma unroll
            for (int q = 0; q < CAR_CQ; ++q) {            // (whole reflectors: the consumers read every slot)
                double v;
                if (q < S || q >= CQ) v = 0.0;
                else if (q == S) v = (C < li) ? 0.0 : ((C == li) ? 1.0 : x[S]);
                else v = x[q];
                car_u32x2 g;
                g.x = (unsigned)__double2loint(v); g.y = (unsigned)__double2hiint(v);
                CB2_VSTORE(__builtin_amdgcn_raw_buffer_store_b64(g, vrs, c8 + (unsigned)(128 * q), so, 0));
            }
            if (C == 0) taup[i] = tau;
        }
#pragma unroll
        for (int k = S; k < MS; ++k)
#pragma unroll
            for (int q = S; q < CQ; ++q) a[k][q] = CB2_GUPD(fma(-tG[k], x[q], a[k][q]), a[k][q]);
        // ---- what the next step is built from: row i+1 and column i+1 of A', and col'[i+1]
        const bool last = li == 15;
        const int n1 = (li + 1) & 15;
        if (!last) {                                         // (uniform)
            if (R == n1) {
#pragma unroll
                for (int q = S; q < CQ; ++q) L.rowg[p * CAR_NS + C + 16 * q] = a[S][q];
                if (C == 0) L.scal[p] = colp[S];
            }
            if (C == n1) {
#pragma unroll
                for (int k = S; k < MS; ++k) L.colg[p * 128 + R + 16 * k] = a[k][S];
            }
        } else {
            if (R == 0) {
#pragma unroll
                for (int q = S; q < CQ; ++q) L.rowg[p * CAR_NS + C + 16 * q] = a[S1][q];
                if (C == 0) L.scal[p] = colp[S1];
            }
            if (C == 0) {
#pragma unroll
                for (int k = S; k < MS; ++k) L.colg[p * 128 + R + 16 * k] = a[k][S + 1];
            }
        }
        CB2_STAMP(5, a[S][S]);
        // ---- partial column sums of H(i): rows >= i + 2 of col' against A'
        double cm[CAR_MS];
#pragma unroll
        for (int k = S; k < MS; ++k) cm[k] = colp[k];
        cm[S] = (R > li + 1) ? cm[S] : 0.0;
        if constexpr (S + 1 < MS) cm[S + 1] = (last && R == 0) ? 0.0 : cm[S + 1];
        double yp[CAR_CQ], s2p = 0.0;
#pragma unroll
        for (int q = S; q < CQ; ++q) yp[q] = cm[S] * a[S][q];
#pragma unroll
        for (int k = S + 1; k < MS; ++k)
#pragma unroll
            for (int q = S; q < CQ; ++q) yp[q] = fma(cm[k], a[k][q], yp[q]);
#pragma unroll
        for (int k = S; k < MS; ++k) s2p = fma(cm[k], cm[k], s2p);
#pragma unroll
        for (int q = S; q < CQ; ++q) L.zpart[R * CAR_NS + C + 16 * q] = yp[q];
        if (C == 0) L.s2part[R] = s2p;
        CB_STAMP(6);
        CB2_BARRIER();
        CB_STAMP(7);
        // the 16 partial sums per live column: column 16 S + t belongs to thread t of waves 0 .. 2 (192 threads cover the
        // live columns from block 1 on).  WAVE 3 has no column of its own then: it sums the norm and runs H(i)'s scalar
        // chain (dlarfg: ~35 dependent instructions) HERE, beside the column sums, instead of every thread running it at
        // the top of the next iteration with nothing to overlap it (block 0: wave 3 also takes columns 192 .. 207)
        if (tid < 192) {                                     // (wave-uniform)
            const int c = 16 * S + tid;
            if (c < 16 * CQ) {
                double z0 = 0.0, z1 = 0.0;
#pragma unroll
                for (int w = 0; w < 16; w += 2) {
                    z0 += L.zpart[w * CAR_NS + c];
                    z1 += L.zpart[(w + 1) * CAR_NS + c];
                }
                L.zsum[c] = z0 + z1;
            }
        } else {
            if constexpr (S == 0) {
                if (tid < 16 * CQ) {
                    double z0 = 0.0, z1 = 0.0;
#pragma unroll
                    for (int w = 0; w < 16; w += 2) {
                        z0 += L.zpart[w * CAR_NS + tid];
                        z1 += L.zpart[(w + 1) * CAR_NS + tid];
                    }
                    L.zsum[tid] = z0 + z1;
                }
            }
            const double s2n = row16_sum(L.s2part[C]);       // (every 16-lane row of the wave: the same sum, same order)
            double tq, s2c;
            larfg_vt(L.scal[p], s2n, tq, s2c);
            if (tid == 192) { L.hsc[0] = tq; L.hsc[1] = s2c; }
        }
        CB_STAMP(8);
        CB2_BARRIER();
        CB_STAMP(9);
    }
    CB_FLUSH(S);
}

// the last reflector, G(m-1): row m-1 after H(m-2) -- no matrix work at all
template <int S, bool FUSED, int MS, int CQ>
__device__ __forceinline__ void car_bidiag2_last(int m, const CarLds2& L, double* __restrict__ vws, double* __restrict__ taup,
                                                 const CarPub& pub) {
    const int tid = threadIdx.x;
    const int R = tid >> 4, C = tid & 15;
    const int i = m - 1, li = i & 15, pp = (i & 1) ^ 1;
    if constexpr (FUSED) car_publish_progress(pub, i);
    double x[CAR_CQ], zS = 0.0;
    const double tauq = L.hsc[0], sc2 = L.hsc[1];
#pragma unroll
    for (int q = S; q < CQ; ++q) {
        const double rg = L.rowg[pp * CAR_NS + C + 16 * q];
        const double zq = fma(sc2, L.zsum[C + 16 * q], rg);
        if (q == S) zS = zq;
        x[q] = fma(-tauq, zq, rg);
    }
    x[S] = (C > li) ? x[S] : 0.0;
    const double zi = rdlane(zS, li);
    const double alpha = fma(-tauq, zi, L.rowg[pp * CAR_NS + i]);
    double ss0 = 0.0, ss1 = 0.0;
#pragma unroll
    for (int q = S; q < CQ; ++q) { if (q & 1) ss1 = fma(x[q], x[q], ss1); else ss0 = fma(x[q], x[q], ss0); }
    const double ss = row16_sum(ss0 + ss1);
    double beta, tau, sc;
    larfg_vt(alpha, ss, tau, sc); (void)beta;
    if (R == li) {
#pragma unroll
        for (int q = 0; q < CAR_CQ; ++q) {
            double v;
            if (q < S || q >= CQ) v = 0.0;
            else if (q == S) v = (C < li) ? 0.0 : ((C == li) ? 1.0 : x[S] * sc);
            else v = x[q] * sc;
            vws[(size_t)i * CAR_NS + C + 16 * q] = v;
        }
        if (C == 0) taup[i] = tau;
    }
}

template <bool FUSED, int MS, int CQ>
__device__ __forceinline__ void car_bidiag2_body(const double* __restrict__ X, int ldx, int N, int m,
                                                 double* __restrict__ vws, double* __restrict__ taup, const CarPub& pub) {
    __shared__ double rowg[2 * CAR_NS];
    __shared__ double colg[2 * 128];
    __shared__ double zpart[16 * CAR_NS];
    __shared__ double s2part[16];
    __shared__ double zsum[CAR_NS + 8];
    __shared__ double scal[2];
    __shared__ double hsc[2];
    const int tid = threadIdx.x;
    const int R = tid >> 4, C = tid & 15;
    double a[CAR_MS][CAR_CQ], colp[CAR_MS];
#pragma unroll
    for (int k = 0; k < CAR_MS; ++k) {
        colp[k] = 0.0;
#pragma unroll
        for (int q = 0; q < CAR_CQ; ++q) {
            const int r = R + 16 * k, c = C + 16 * q;
            a[k][q] = 0.0;                                   // (slots beyond MS / CQ: constants, never touched again)
            if (k < MS && q < CQ) a[k][q] = (c < N && r < m) ? ((r == 0) ? 1.0 : X[(size_t)c * ldx + (r - 1)]) : 0.0;
        }
    }
    // "step -1" left the matrix untouched: its row 0 and column 0, no H
    if (tid < CAR_NS) { rowg[CAR_NS + tid] = (tid < N) ? 1.0 : 0.0; zsum[tid] = 0.0; }
    if (tid < 128) colg[128 + tid] = (tid < m && N > 0) ? ((tid == 0) ? 1.0 : X[tid - 1]) : 0.0;
    if (tid < 8) zsum[CAR_NS + tid] = 0.0;
    if (tid < 2) { scal[tid] = 0.0; hsc[tid] = 0.0; }    // (tau = 0: "H(-1)" is the identity)
    __syncthreads();
    const CarLds2 L{rowg, colg, zpart, s2part, zsum, scal, hsc};
    car_bidiag2_block<0, FUSED, MS, CQ>(a, colp, m, L, vws, taup, pub);
    if constexpr (MS > 1) if (m > 17) car_bidiag2_block<1, FUSED, MS, CQ>(a, colp, m, L, vws, taup, pub);
    if constexpr (MS > 2) if (m > 33) car_bidiag2_block<2, FUSED, MS, CQ>(a, colp, m, L, vws, taup, pub);
    if constexpr (MS > 3) if (m > 49) car_bidiag2_block<3, FUSED, MS, CQ>(a, colp, m, L, vws, taup, pub);
    if constexpr (MS > 4) if (m > 65) car_bidiag2_block<4, FUSED, MS, CQ>(a, colp, m, L, vws, taup, pub);
    if constexpr (MS > 5) if (m > 81) car_bidiag2_block<5, FUSED, MS, CQ>(a, colp, m, L, vws, taup, pub);
    if constexpr (MS > 6) if (m > 97) car_bidiag2_block<6, FUSED, MS, CQ>(a, colp, m, L, vws, taup, pub);
    const int sl = (m - 1) >> 4;
    if (sl == 0) car_bidiag2_last<0, FUSED, MS, CQ>(m, L, vws, taup, pub);
    if constexpr (MS > 1) if (sl == 1) car_bidiag2_last<1, FUSED, MS, CQ>(m, L, vws, taup, pub);
    if constexpr (MS > 2) if (sl == 2) car_bidiag2_last<2, FUSED, MS, CQ>(m, L, vws, taup, pub);
    if constexpr (MS > 3) if (sl == 3) car_bidiag2_last<3, FUSED, MS, CQ>(m, L, vws, taup, pub);
    if constexpr (MS > 4) if (sl == 4) car_bidiag2_last<4, FUSED, MS, CQ>(m, L, vws, taup, pub);
    if constexpr (MS > 5) if (sl == 5) car_bidiag2_last<5, FUSED, MS, CQ>(m, L, vws, taup, pub);
    if constexpr (MS > 6) if (sl >= 6) car_bidiag2_last<6, FUSED, MS, CQ>(m, L, vws, taup, pub);
}

template <int MS, int CQ>
__global__ __launch_bounds__(CAR_BT) void k_car_bidiag(const double* __restrict__ X, int ldx, int N, int m,
                                                       double* __restrict__ vws, double* __restrict__ taup) {
    car_bidiag2_body<false, MS, CQ>(X, ldx, N, m, vws, taup, CarPub{});
}

// ---------------- phases 1 + 2 in one launch ----------------
// Phi = G(0) ... G(m-1) [0; I] is also the last N - m columns of P = G(0) G(1) ... G(m-1) accumulated FORWARD, and in
// that order every ROW of P is independent: p <- p - tau (p . v) v^T as each reflector appears.  So the 24 us that
// k_car_phi spent after the bidiagonalisation (100 dependent steps per column, on an otherwise idle chip) move BESIDE
// it: workgroup 0 is the bidiagonalisation and publishes every reflector as tagged granules; the other workgroups
// that find themselves on ITS XCD (hardware id; the rest exit at once) take tickets for groups of four rows, one wave
// per row, and follow the reflectors as they appear -- one step behind the producer, finished ~1 us after it.
// Tags carry an epoch (one per launch), so nothing has to be cleared between launches; spins are bounded (error word:
// the pivot kernel then reports n_keep = -1 and the caller redoes the step with the launches that do not depend on
// partner workgroups: sober_car_device_ex, SOBER_CAR_SAFE).
#ifdef SP_TSTAMPS     // diagnostic build: the launches' own timeline on the chip-wide 100 MHz clock (scripts/pivot_stamps.py)
__device__ unsigned long long g_car_rt[8];   // 0 producer in, 1 last reflector out, 2 producer done, 3 last consumer done
#define CAR_RT(K) do { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); g_car_rt[K] = t_; } while (0)
#define CAR_RT_MAX(K) do { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); atomicMax(&g_car_rt[K], t_); } while (0)
#else
#define CAR_RT(K) do { } while (0)
#define CAR_RT_MAX(K) do { } while (0)
#endif
template <int MS, int CQ>
__global__ __launch_bounds__(CAR_BT) void k_car_bidiag_fused(const double* __restrict__ X, int ldx, int N, int m,
                                                             double* __restrict__ vws, double* __restrict__ taup,
                                                             double* __restrict__ Phi, void* __restrict__ comm,
                                                             unsigned cbytes, unsigned epoch, unsigned spin_limit) {
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
#ifdef SP_TSTAMPS
        if (threadIdx.x == 0) { CAR_RT(0); g_car_rt[3] = 0ull; }
#endif
        car_bidiag2_body<true, MS, CQ>(X, ldx, N, m, vws, taup, pub);
        __syncthreads();
        if (threadIdx.x == 0) CAR_RT(1);
        car_publish_progress(pub, m);                                  // the last reflector (this time the wait is real)
        if (threadIdx.x == 0) CAR_RT(2);
        return;
    }
    // ---- consumers ----
    __shared__ int s_group;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) {
        unsigned spins = 0, w;
        int grp = -1;
        while (((w = __hip_atomic_load(words + CARF_XCD / 4, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) >> 4) != epoch) {
            if (++spins > CARF_SPIN_LIMIT) { w = 0; break; }      // (the election is not what the test switch shortens)
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
            if (spin_limit == 0u) {                                    // (the test switch: give up at once, through the real path --
                __builtin_amdgcn_raw_buffer_store_b32(1, rs, CARF_ERR, 0, 16);   //  a limit of one poll stopped being a give-up when
                failed = true;                                         //  a small step's producer got ahead of its consumers)
            }
            for (int i = 0; i < m && !failed; ++i) {
                unsigned spins = 0;
                while (have <= i) {                                    // (one broadcast load per poll)
                    const unsigned w = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs, CARF_PROGRESS, 0, 16);
                    if ((w >> 7) == epoch) have = (int)(w & 127u);
                    if (have > i) break;
                    if (++spins > spin_limit ||
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
                if (c < CAR_NS && col >= 0 && col < CAR_PC) Phi[(size_t)col * CAR_NS + r] = (r < N && col < NC) ? p[w][q] : 0.0;
            }
            // (columns beyond what the lanes above cover)
            for (int col = max(CAR_NS - m, 0) + lane; col < CAR_PC; col += 64) Phi[(size_t)col * CAR_NS + r] = 0.0;
        }
        // this group of rows is in memory: the pivot kernel refuses to run on a Phi with a group missing
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
        if (tid == 0 && !failed)
            __hip_atomic_fetch_add(words + CARF_DONE / 4, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) CAR_RT_MAX(3);
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
            Phi[(size_t)col * CAR_NS + r] = (r < N) ? phi[q] : 0.0;
            if (phi_out != nullptr && col < NC && r < N) phi_out[(size_t)r * NC + col] = phi[q];
        }
    }
}


// ---------------- phase 3: the pivots of SOBER/_rchq.py:237-266 ----------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_min_u32(unsigned v) {
    // (`old` = the minimum's identity: lanes a row mask leaves out read it, which is what they should combine with --
    //  and the DPP combiner may then fold the move into the v_min_u32 itself)
    const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, CTRL, ROW_MASK, 0xf, false);
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

// ---------------- phase 3, streaming form ----------------
// The same pivots without a workgroup barrier per pivot (the design of car_mc.hip's k_mc_pivot, through LDS instead of
// L2).  A wave owns SP_BC CONSECUTIVE columns and keeps its own copy of the weights.  While a column of its block is the
// pivot column it runs the ratio test on its registers, publishes (column, index, alpha, 1/pivot) in an LDS ring and goes
// straight on to its next column -- seven of eight pivots need no hand-over at all; every other wave applies the
// published pivots to its columns as they arrive, at low priority, in the issue slots the owner's dependent chain
// leaves empty.  (The round-1 kernel rotated the ownership with every pivot -- column c -> wave c mod 16 --: one barrier,
// one hand-over of the weights and one LDS round trip of the pivot column per pivot, 1.14 us each.)
constexpr int SP_W = 16, SP_BC = 7, SP_RING = 32;

#ifdef SP_TSTAMPS     // diagnostic build (scripts/pivot_stamps.py): s_memrealtime (100 MHz) of every publish, by pivot index
__device__ unsigned long long g_sp_stamps[260];
constexpr int SP_W_MAX_DBG = 16;   // (= SP_W)
__device__ unsigned long long g_sp_seg[SP_W_MAX_DBG * 8];   // per wave: ticks summed by segment of the produce step
#define SP_SEG(K, DEP_CONSTRAINT, DEP) do { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_), DEP_CONSTRAINT(DEP) :: "memory"); \
        sp_seg_[K] += t_ - sp_last_; sp_last_ = t_; } while (0)
#else
#define SP_SEG(K, DEP_CONSTRAINT, DEP) do { } while (0)
#endif
struct SpSlot { double col[256]; double alpha, rpp; int piv; int tag; };
// tags and progress words through LDS-typed pointers: a volatile access through a generic pointer becomes a FLAT
// instruction followed by s_waitcnt vmcnt(0) -- the publishing wave waited for its own tag store on every pivot
typedef __attribute__((address_space(3))) volatile int sp_lds_vint;
typedef __attribute__((address_space(3))) int sp_lds_int;
__device__ __forceinline__ sp_lds_vint* sp_lds_v(volatile int* p) { return (sp_lds_vint*)p; }
// alive[q]: rows 64 q + lane that are inside the step and not cancelled yet -- a wave-uniform lane mask (a pair of
// scalar registers used directly as the select's condition; per-lane booleans cost a compare, a byte and/or and a
// move per slot and pivot on the chain)
struct SpState { double mu[4]; unsigned long long alive[4]; };
__device__ __forceinline__ bool sp_alive(const SpState& st, int q) { return __builtin_amdgcn_inverse_ballot_w64(st.alive[q]); }
constexpr int SP_BAND = 4;          // screened ratio test: candidates = high words within SP_BAND of the minimum

#ifdef SP_TSTAMPS
#define SP_DBG_ARGS0 , unsigned long long (&sp_seg_)[8], unsigned long long& sp_last_
#else
#define SP_DBG_ARGS0
#endif
// The ratio test of :240-247.  Screened form (round 5): the quotients' HIGH WORDS from mu * v_rcp_f64(col) (relative
// error e < 2^-22, tests/test_hip_round5.py pins it) decide the winner whenever exactly one lane lies within SP_BAND
// high-word steps (>= 2^-21 each) of the minimum: every other lane's exact quotient is then larger than the winner's
// by more than 2 e -- the same index as the exact argmin, and alpha and 1/pivot are computed exactly for that lane
// only (two divisions per pivot instead of four IEEE divisions and four order-preserving 64-bit keys).  Anything else
// -- two lanes inside the band, a quotient that is negative, infinite or NaN (keys below 0x80100000 after the shift
// that wraps them), no candidate -- takes the exact test below, unchanged.
template <int NQ>
__device__ __forceinline__ int sp_ratio_test(const double (&col)[4], const SpState& st, bool exact_only SP_DBG_ARGS0) {
    if (!exact_only) {                                                // (SOBER_CAR_EXACT_RATIO: the A/B of the tests)
        unsigned sk[4];
        unsigned smin = 0xffffffffu;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const double r = st.mu[q] * __builtin_amdgcn_rcp(col[q]);
            const unsigned k = (unsigned)__double2hiint(r) + 0x80100000u;   // [+0, DBL_MAX] -> [0x80100000, 0xffffffff]; the rest below
            sk[q] = (col[q] > 0.0) ? k : 0xffffffffu;                        // (col is 0 on rows that are not alive)
            smin = min(smin, sk[q]);
        }
        SP_SEG(5, "+v", smin);                                       // the screen's keys, the lane's minimum
        unsigned H = wave_min_u32(smin);
        SP_SEG(6, "+s", H);                                          // the wave's minimum
        if (__builtin_expect(H >= 0x80100000u && H < 0xffffffffu - (unsigned)SP_BAND, 1)) {
            unsigned long long mb[4] = {0ull, 0ull, 0ull, 0ull};
            int cnt = 0;
#pragma unroll
            for (int q = 0; q < NQ; ++q) { mb[q] = __ballot(sk[q] <= H + (unsigned)SP_BAND); cnt += __popcll(mb[q]); }
            if (__builtin_expect(cnt == 1, 1)) {                      // one bit in all: slot and lane without a branch
                const int f = __ffsll((long long)(mb[0] | mb[1] | mb[2] | mb[3])) - 1;
                const int kp = (mb[1] != 0ull ? 1 : 0) + (mb[2] != 0ull ? 2 : 0) + (mb[3] != 0ull ? 3 : 0);
                return f + 64 * kp;
            }
        }
    }
    // the exact test: IEEE quotients, order-preserving 64-bit keys, first index among equal ones
    double rt[4];
    unsigned kh[4], kl[4];
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
    unsigned H = wave_min_u32(hmin);
    if (H == 0xffffffffu) return -1;                                  // uniform: no candidate (:241-242)
    unsigned long long mb[4];
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
// mu[:] = mu - alpha * Phi[:, 0]; mu[idx] = 0  (:253-254; the product and the subtraction contract to ONE fma under
// -ffp-contract=fast -- half an ulp from the tensor expression's two roundings, inside what the parity tests bound)
template <int NQ>
__device__ __forceinline__ void sp_mu_step(SpState& st, const double (&col)[4], double alpha, int piv, int lane) {
    const unsigned long long bit = 1ull << (piv & 63);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        st.alive[q] &= ~((piv >> 6) == q ? bit : 0ull);               // scalar
        // (no select: col is 0 on cancelled rows, so their weights stay what they are; the row cancelled NOW keeps a
        //  rounding residue instead of the 0 of :254 -- nobody reads it again, its col is 0 from here on, and the
        //  output applies the mask)
        st.mu[q] = __dsub_rn(st.mu[q], __dmul_rn(alpha, col[q]));
    }
}
// what the pivot needs from row `piv`: its entries in my columns J0 .. (and, for the wave that found it, the row's weight
// and its entry of the pivot column) -- ONE uniform switch over the row's slot, reads only
// (with the updates inside its cases the compiler keeps a renamed copy of phi per case and moves it back behind the
//  switch -- 24 to 52 v_mov_b64 per pivot on the producing wave's chain; picking the slot with three selects per column
//  makes it index a copy of phi in scratch memory -- 2x slower --, picking it arithmetically with four 0/1 weights
//  measured 3.5 us per step slower than the switch)
template <int KP, int J0, bool OWN>
__device__ __forceinline__ void sp_pivot_row_kp(const double (&phi)[SP_BC][4], const SpState& st, const double (&col)[4], int lp,
                                                double (&prow)[SP_BC], double& mp, double& cp) {
    if constexpr (OWN) { mp = rdlane(st.mu[KP], lp); cp = rdlane(col[KP], lp); }
#pragma unroll
    for (int j = J0; j < SP_BC; ++j) prow[j] = rdlane(phi[j][KP], lp);
}
template <int J0, int NQ, bool OWN>
__device__ __forceinline__ void sp_pivot_row(const double (&phi)[SP_BC][4], const SpState& st, const double (&col)[4], int piv,
                                             double (&prow)[SP_BC], double& mp, double& cp) {
    const int kp = piv >> 6, lp = piv & 63;
    if constexpr (NQ == 1) {
        sp_pivot_row_kp<0, J0, OWN>(phi, st, col, lp, prow, mp, cp);
    } else if constexpr (NQ == 2) {
        if (kp == 0) sp_pivot_row_kp<0, J0, OWN>(phi, st, col, lp, prow, mp, cp);          // uniform
        else sp_pivot_row_kp<1, J0, OWN>(phi, st, col, lp, prow, mp, cp);
    } else {
        switch (kp) {                                                 // uniform (a two-level if tree measured 2 us slower)
            case 0: sp_pivot_row_kp<0, J0, OWN>(phi, st, col, lp, prow, mp, cp); break;
            case 1: sp_pivot_row_kp<1, J0, OWN>(phi, st, col, lp, prow, mp, cp); break;
            case 2: sp_pivot_row_kp<2, J0, OWN>(phi, st, col, lp, prow, mp, cp); break;
            default: sp_pivot_row_kp<3, J0, OWN>(phi, st, col, lp, prow, mp, cp); break;
        }
    }
}
//   Phi[:, c] -= Phi[:, 0] * (Phi[idx, c] / Phi[idx, 0])   (:260-266), my columns J0 .. SP_BC-1
template <int J0, int NQ>
__device__ __forceinline__ void sp_elim_rows(double (&phi)[SP_BC][4], const double (&col)[4], const double (&prow)[SP_BC], double rpp) {
#pragma unroll
    for (int j = J0; j < SP_BC; ++j) {
        const double qv = prow[j] * rpp;
#pragma unroll
        for (int q = 0; q < NQ; ++q) phi[j][q] = fma(-qv, col[q], phi[j][q]);
    }
}
template <int J0, int NQ>
__device__ __forceinline__ void sp_elim(double (&phi)[SP_BC][4], const SpState& st, const double (&col)[4], int piv, double rpp) {
    double prow[SP_BC], mp, cp;
    sp_pivot_row<J0, NQ, false>(phi, st, col, piv, prow, mp, cp);
    sp_elim_rows<J0, NQ>(phi, col, prow, rpp);
}
// wait for pivot s in the ring; false = give up (bounded).  {piv, tag} travel as ONE 8-byte word.
// EAGER (the wave whose block is next: its wait IS the hand-over, on the critical chain): every poll reads the whole
// record behind the tag word -- a wave's LDS reads return in order, so when the tag is there the data read behind it
// is the pivot's -- instead of a second round trip after the tag has been seen; no sleep between polls.
typedef __attribute__((address_space(3))) volatile unsigned long long sp_lds_vu64;
typedef __attribute__((address_space(3))) unsigned long long sp_lds_u64;
typedef __attribute__((address_space(3))) volatile double sp_lds_vf64;
template <int NQ, bool EAGER>
__device__ __forceinline__ bool sp_consume(SpSlot* ring, int s, int lane, double (&col)[4], double& alpha, double& rpp, int& piv) {
    SpSlot& e = ring[s % SP_RING];
    sp_lds_vu64* pt = (sp_lds_vu64*)&e.piv;
    unsigned spins = 0;
    if constexpr (EAGER) {
        sp_lds_vf64* ar = (sp_lds_vf64*)&e.alpha;
        sp_lds_vf64* ec = (sp_lds_vf64*)&e.col[lane];
        for (;;) {
            const unsigned long long v = *pt;
            alpha = ar[0]; rpp = ar[1];
#pragma unroll
            for (int q = 0; q < NQ; ++q) col[q] = ec[64 * q];
            if (__builtin_amdgcn_readfirstlane((int)(v >> 32)) == s + 1) { piv = __builtin_amdgcn_readfirstlane((int)(unsigned)v); return true; }
            if (++spins > (1u << 24)) return false;
        }
    } else {
        unsigned long long v;
        while (v = *pt, __builtin_amdgcn_readfirstlane((int)(v >> 32)) != s + 1) {   // (uniform exit: what the loop carries -- the lane masks -- stays scalar)
            if (++spins > (1u << 24)) return false;
            __builtin_amdgcn_s_sleep(1);
        }
        piv = __builtin_amdgcn_readfirstlane((int)(unsigned)v);
        alpha = e.alpha; rpp = e.rpp;
#pragma unroll
        for (int q = 0; q < NQ; ++q) col[q] = e.col[lane + 64 * q];
        return true;
    }
}

// pivot `sp` from column JJ of my block: ratio test, publish, update of my weights and of my columns behind JJ
#ifdef SP_TSTAMPS
#define SP_DBG_ARGS , unsigned long long (&sp_seg_)[8], unsigned long long& sp_last_
#define SP_DBG_PASS , sp_seg_, sp_last_
#else
#define SP_DBG_ARGS
#define SP_DBG_PASS
#endif
template <int JJ, int NQ>
__device__ __forceinline__ void sp_produce_step(double (&phi)[SP_BC][4], SpState& st, SpSlot* ring, int sp, int lane,
                                                double (&col)[4], bool& stop, bool exact_only SP_DBG_ARGS) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) col[q] = sp_alive(st, q) ? phi[JJ][q] : 0.0;
    SP_SEG(0, "+v", col[0]);                                  // (since the previous stamp: the elimination behind the last pivot)
    int piv = sp_ratio_test<NQ>(col, st, exact_only SP_DBG_PASS);
    SpSlot& e = ring[sp % SP_RING];
    if (__builtin_expect(piv < 0, 0)) {                       // Q6: no candidate, the loop ends here (:241-242)
        if (lane == 0) *(sp_lds_u64*)&e.piv = 0xffffffffull | ((unsigned long long)(unsigned)(sp + 1) << 32);
        stop = true;
        return;
    }
    // the winner's row, once: its weight and pivot entry (alpha and 1 / pivot by the same two IEEE divisions whichever
    // test found it) and its entries of my remaining columns
    double prow[SP_BC], mp = 0.0, cp = 1.0;
    sp_pivot_row<JJ + 1, NQ, true>(phi, st, col, piv, prow, mp, cp);
    SP_SEG(7, "+v", cp);                                      // ballots, the winner's lane, the read-outs
    double al = mp / cp, rp = 1.0 / cp;
    SP_SEG(1, "+v", al);                                      // the two divisions
#pragma unroll
    for (int q = 0; q < NQ; ++q) e.col[lane + 64 * q] = col[q];
#ifdef SP_TSTAMPS
    { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); if (lane == 0) g_sp_stamps[sp] = t_; }
#endif
    asm volatile("" ::: "memory");                            // (program order; a wave's LDS operations execute in order:
    if (lane == 0) {                                          //  the tag word lands after the data it releases -- no wait)
        e.alpha = al; e.rpp = rp;
        *(sp_lds_u64*)&e.piv = (unsigned long long)(unsigned)piv | ((unsigned long long)(unsigned)(sp + 1) << 32);
    }
    asm volatile("" ::: "memory");
    SP_SEG(2, "+s", piv);                                     // publish
    sp_mu_step<NQ>(st, col, al, piv, lane);
    SP_SEG(3, "+v", st.mu[0]);                                // weights
    sp_elim_rows<JJ + 1, NQ>(phi, col, prow, rp);
}

// NQ: 64-row slots in use (N <= 64 NQ): a 20-point step does not run the ratio test of a 200-point one
template <int NQ>
__global__ __launch_bounds__(SP_W * 64) void k_car_pivot_stream(const double* __restrict__ Phi, int N, int m,
                                                                const double* __restrict__ mu_in,
                                                                int32_t* __restrict__ keep_rank,
                                                                double* __restrict__ w_star,
                                                                int32_t* __restrict__ n_keep_out,
                                                                double* __restrict__ mu_out,
                                                                const unsigned* __restrict__ err, int exact_ratio) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sp_lds[];
    SpSlot* ring = (SpSlot*)sp_lds;
    sp_lds_vint* prog = sp_lds_v((volatile int*)(sp_lds + sizeof(SpSlot) * SP_RING));   // [SP_W]: pivots consumed so far
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = N - m;
    const int c0 = w * SP_BC;
#ifdef SP_TSTAMPS
    if (tid == 0) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); g_sp_stamps[258] = t_; }
#endif
    if (tid < SP_RING) ring[tid].tag = 0;
    if (tid < SP_W) prog[tid] = 0;
    // the fused launch in front gave up on a reflector, or not every group of Phi's rows found a consumer
    const bool broken = err != nullptr && (err[CARF_ERR / 4] != 0u || err[CARF_DONE / 4] != (unsigned)(CAR_NS / 8));
    __syncthreads();                                                  // (the only workgroup barrier)
#ifdef SP_TSTAMPS
    if (tid == 0) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); g_sp_stamps[256] = t_; }
#endif
    if (broken) { if (tid == 0) *n_keep_out = -1; return; }
    if (c0 >= K && w != 0) return;
    double phi[SP_BC][4];
    SpState st;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int row = lane + 64 * q;
        const int nq = min(max(N - 64 * q, 0), 64);
        st.alive[q] = nq >= 64 ? ~0ull : ((1ull << nq) - 1ull);
        st.mu[q] = row < N ? mu_in[row] + 0.0 : 0.0;
#pragma unroll
        for (int j = 0; j < SP_BC; ++j)
            phi[j][q] = (c0 + j < K && row < N) ? Phi[(size_t)(c0 + j) * CAR_NS + row] : 0.0;
    }
    bool fail = false, stop = false;
    double col[4] = {0.0, 0.0, 0.0, 0.0};
    // the pivots before my block: apply them to all my columns as they arrive
    const int s_mine = min(c0, K);
    for (int s = 0; s < s_mine && !fail && !stop; ++s) {
        double al, rp;
        int piv;
        if (s == s_mine - SP_BC) __builtin_amdgcn_s_setprio(2);      // on deck: the hand-over is on the critical chain
        // (two copies of the wait, picked by a uniform test: the eager one only for the last SP_BC pivots before my block)
        const bool got = (s >= s_mine - SP_BC) ? sp_consume<NQ, true>(ring, s, lane, col, al, rp, piv)
                                               : sp_consume<NQ, false>(ring, s, lane, col, al, rp, piv);
        if (!got) { fail = true; break; }
        if (lane == 0) prog[w] = s + 1;
        if (piv < 0) { stop = true; break; }
        sp_mu_step<NQ>(st, col, al, piv, lane);
        sp_elim<0, NQ>(phi, st, col, piv, rp);
    }
    // my block
    if (!fail && !stop && c0 < K) {
        __builtin_amdgcn_s_setprio(3);                                // the critical chain of the whole kernel
        const int s_end = min(c0 + SP_BC, K);
        // a ring slot is reused SP_RING pivots later: everybody who still follows must be past what I overwrite
        if (s_end > SP_RING) {
            const int need = s_end - SP_RING;
            unsigned spins = 0;
            // (lane o looks at wave o: one LDS read and a ballot -- sixteen dependent reads by every lane were 1 us of
            //  every block's hand-over, 10 % of the kernel: scripts/pivot_stamps.py)
            const int o = lane & (SP_W - 1);
            const bool follows = (o > w && o * SP_BC < K) || (o == 0 && w != 0);
            for (;;) {
                if (__ballot(follows && prog[o] < need) == 0ull) break;
                if (++spins > (1u << 22)) { fail = true; break; }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        // one instance of the step per column of my block: the columns keep their registers (rotating them down after
        // every pivot -- 24 doubles moved behind the elimination they depend on -- cost 15 % of the kernel)
        int sp = c0;
#ifdef SP_TSTAMPS
        unsigned long long sp_seg_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sp_last_;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(sp_last_) :: "memory");
#endif
#define SP_STEP(JJ) if (sp < s_end && !stop) { sp_produce_step<JJ, NQ>(phi, st, ring, sp, lane, col, stop, exact_ratio != 0 SP_DBG_PASS); ++sp; }
        SP_STEP(0) SP_STEP(1) SP_STEP(2) SP_STEP(3) SP_STEP(4) SP_STEP(5) SP_STEP(6)
#undef SP_STEP
        static_assert(SP_BC == 7, "one SP_STEP per column of a block");
#ifdef SP_TSTAMPS
        SP_SEG(4, "+v", col[0]);                              // the elimination behind the block's last pivot (nothing left: ~ a stamp's own cost)
        if (lane == 0) for (int k_ = 0; k_ < 8; ++k_) g_sp_seg[w * 8 + k_] = sp_seg_[k_];
#endif
        __builtin_amdgcn_s_setprio(0);
    }
    if (w != 0) return;
    // wave 0 follows the remaining pivots for the weights and writes the result
    for (int s = c0 + SP_BC; s < K && !fail && !stop; ++s) {
        double al, rp;
        int piv;
        if (!sp_consume<NQ, false>(ring, s, lane, col, al, rp, piv)) { fail = true; break; }
        if (lane == 0) prog[0] = s + 1;
        if (piv < 0) { stop = true; break; }
        sp_mu_step<NQ>(st, col, al, piv, lane);
    }
    int base = 0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int row = lane + 64 * q;
        const double v = sp_alive(st, q) ? st.mu[q] + 0.0 : 0.0;     // mu[idx] = 0 of :254 for every cancelled row; -0.0 -> +0.0
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
#ifdef SP_TSTAMPS
    if (lane == 0) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); g_sp_stamps[257] = t_; }
#endif
}

// ---------------- the extra elimination of the acquisition-guided branch (SOBER/_rchq.py:87-106, :177-196) ----------
// After the Caratheodory step with one more test function (the objective), n1 = b + 1 points are left; their
// weights move along the null vector w_null of [X_p; 1] (functions x points, a 1-dimensional null space) in the
// direction that does not decrease sum w * calc_obj until one more weight reaches zero.  w_null arrives as the
// one-column null-space basis of sober_car_device's phi_out (unit norm, sign arbitrary -- the reference fixes the
// sign by the objective, :92-93, so ANY null vector gives its result).  One workgroup; n1 <= 512.
//   in : phi[n1], objp[n1], w1[n1] (ranks of the first step), rank1[Nsets] (set -> rank or -1)
//   out: keep_rank[Nsets] (new ranks), w_star[n_keep], *n_keep_out
// BY_ROW: phi and objp are indexed by SET (sober_null_vector's output; the level's objective column) and brought to rank
// order here; the first step must have left exactly n1 sets (*n_keep1 == n1), else *n_keep_out = -2 and nothing is written
// (the caller's host route takes the level: the reference then reads a singular vector of a full-rank matrix); -1: it gave up.
template <bool BY_ROW>
__global__ __launch_bounds__(512) void k_second_elim(const double* __restrict__ phi, const double* __restrict__ objp,
                                                     const double* __restrict__ w1, const int32_t* __restrict__ rank1,
                                                     int n1, int Nsets, int32_t* __restrict__ keep_rank,
                                                     double* __restrict__ w_star, int32_t* __restrict__ n_keep_out,
                                                     const int32_t* __restrict__ n_keep1, const int32_t* __restrict__ status) {
    __shared__ double red[512];
    __shared__ unsigned long long kmin[512];
    __shared__ int kidx[512];
    __shared__ int cnt[512];
    const int t = threadIdx.x;
    double ph, ob;
    if constexpr (BY_ROW) {
        if (*n_keep1 != n1) { if (t == 0) *n_keep_out = (*n_keep1 < 0) ? -1 : -2; return; }      // (uniform)
        // (the null-vector kernel's verdict: a rank-deficient A2 has no null LINE -- the caller's host route)
        if (status != nullptr && *status != 0) { if (t == 0) *n_keep_out = -2; return; }
        __shared__ double s_ph[512], s_ob[512];
        for (int sidx = t; sidx < Nsets; sidx += 512) {
            const int r1 = rank1[sidx];
            if (r1 >= 0 && r1 < n1) { s_ph[r1] = phi[sidx]; s_ob[r1] = objp[sidx]; }
        }
        __syncthreads();
        ph = t < n1 ? s_ph[t] : 0.0; ob = t < n1 ? s_ob[t] : 0.0;
    } else {
        ph = t < n1 ? phi[t] : 0.0; ob = t < n1 ? objp[t] : 0.0;
    }
    const double w = t < n1 ? w1[t] : 0.0;
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
    // w_star = w_star - alpha[idx_sp] * w_null; w_star[idx_sp] = 0   (:99-100; one fma under -ffp-contract=fast)
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
extern "C" int sober_car_big_supported(int N, int m);
extern "C" int64_t sober_car_big_ws_bytes(int N, int m);
extern "C" int sober_car_big_device(const double* X, int ldx, int N, int m, const double* mu_in, int32_t* keep_rank,
                                    double* w_star, int32_t* n_keep, double* mu_out, double* phi_out, void* ws,
                                    int64_t ws_bytes, void* stream);

// one compute unit (batch <= 100)
static int car_one_cu(int N, int m) {
    return (m >= 2 && N > m && N <= sober::CAR_NS && m <= 16 * sober::CAR_MS && N - m <= sober::SP_W * sober::SP_BC) ? 1 : 0;
}

extern "C" int sober_car_supported(int N, int m) {
    return (car_one_cu(N, m) || sober_car_mc_supported(N, m) || sober_car_big_supported(N, m)) ? 1 : 0;
}

// scratch: reflectors (m x 208), tau (m, padded to 128), Phi (208 x 128)
// (a workspace sized for (N, m) also serves every (N' <= N, m): the final direct level)
extern "C" int64_t sober_car_ws_bytes(int N, int m) {
    const int64_t one = ((int64_t)m * sober::CAR_NS + 128 + (int64_t)sober::CAR_NS * sober::CAR_PC + 512) * (int64_t)sizeof(double)   // (+512: stamp block)
                        + sober::carf_bytes(0);                                       // + the fused launch's words
    if (car_one_cu(N, m)) return one;
    // (beyond one compute unit the memory-resident route of car_big.hip is the SOBER_CAR_SAFE rung of the multi-CU sizes too)
    int64_t need = sober_car_big_supported(N, m) ? sober_car_big_ws_bytes(N, m) : 0;
    if (need < one) need = one;
    // (some N' <= N may be a multi-CU size although N is not: the final direct level of a batch-250 run has 251 .. 448 points)
    if (sober_car_mc_supported(N < m + 1 ? N : m + 1, m)) { const int64_t mc = sober_car_mc_ws_bytes(N, m); if (mc > need) need = mc; }
    return need;
}

// the per-device function attribute of the streaming pivot kernel (dynamic LDS beyond 64 KB)
static int car_pivot_attr(size_t sp_bytes) {
    static std::atomic<unsigned long long> done{0};
    if (sober_attr_needed(done)) {
        HIP_TRY(hipFuncSetAttribute((const void*)sober::k_car_pivot_stream<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sp_bytes));
        HIP_TRY(hipFuncSetAttribute((const void*)sober::k_car_pivot_stream<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sp_bytes));
        HIP_TRY(hipFuncSetAttribute((const void*)sober::k_car_pivot_stream<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sp_bytes));
        sober_attr_done(done);
    }
    return 0;
}

// SOBER_CAR_FORCE_GIVEUP (any value; read when the library is loaded and by sober_reload_switches): the test switch that makes the launches which depend on
// partner workgroups give up -- the fused launch's consumers give up at once (spin limit 0), the multi-CU route reports
// n_keep = -1 -- so that the recovery of the callers can be exercised (tests/test_hip_parity.py).
extern "C" int sober_car_giveup_forced(void) { return sober::switches().car_force_giveup ? 1 : 0; }

extern "C" int sober_car_safe_supported(int N, int m) { return (car_one_cu(N, m) || sober_car_big_supported(N, m)) ? 1 : 0; }

// the bidiagonalisation's instantiation by size: row / column slots in use (m <= 16 MS_, N <= 16 CQ_)
#define CAR_BY_SIZE(m_, N_, LAUNCH)                                                                    \
    do {                                                                                               \
        if ((m_) <= 16 && (N_) <= 32) { constexpr int MS_ = 1, CQ_ = 2; LAUNCH; }                      \
        else if ((m_) <= 32 && (N_) <= 64) { constexpr int MS_ = 2, CQ_ = 4; LAUNCH; }                 \
        else if ((m_) <= 64 && (N_) <= 112) { constexpr int MS_ = 4, CQ_ = 7; LAUNCH; }                \
        else if ((N_) <= 112) { constexpr int MS_ = sober::CAR_MS, CQ_ = 7; LAUNCH; }                  \
        else { constexpr int MS_ = sober::CAR_MS, CQ_ = sober::CAR_CQ; LAUNCH; }                       \
    } while (0)

// the pivot kernel's instantiation by size: 64-row slots in use
#define CAR_PIVOT_BY_SIZE(N_, LAUNCH)                                  \
    do {                                                               \
        if ((N_) <= 64) { constexpr int NQ_ = 1; LAUNCH; }             \
        else if ((N_) <= 128) { constexpr int NQ_ = 2; LAUNCH; }       \
        else { constexpr int NQ_ = 4; LAUNCH; }                        \
    } while (0)

// v_rcp_f64 as the screened ratio test uses it (the test pins its relative error below the band's 2^-22 per SP_BAND step)
namespace sober {
__global__ __launch_bounds__(256) void k_probe_rcp(const double* __restrict__ x, double* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_rcp(x[i]);
}
}
extern "C" int sober_probe_rcp(const double* x, double* out, int64_t n, void* stream) {
    if (n < 0 || (n > 0 && (!x || !out))) return SOBER_E_ARG;
    if (n == 0) return 0;
    hipLaunchKernelGGL(sober::k_probe_rcp, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, out, n);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int sober_car_device_ex(const double* X, int ldx, int N, int m, const double* mu_in,
                                   int32_t* keep_rank, double* w_star, int32_t* n_keep, double* mu_out,
                                   double* phi_out, void* ws, int64_t ws_bytes, int mode, void* stream) {
    if (!X || !mu_in || !keep_rank || !w_star || !n_keep || !mu_out || !ws || ldx < m - 1) return SOBER_E_ARG;
    if (mode != SOBER_CAR_DEFAULT && mode != SOBER_CAR_SAFE) return SOBER_E_ARG;
    if (!car_one_cu(N, m)) {
        // beyond one compute unit: car_mc.hip (matrix in the registers of nine compute units, batch <= 224; every launch there
        // depends on partner workgroups), else -- and in SOBER_CAR_SAFE mode -- car_big.hip (matrix in memory, a launch per
        // dependency: nothing to give up)
        if (mode == SOBER_CAR_DEFAULT && sober_car_mc_supported(N, m))
            return sober_car_mc_device(X, ldx, N, m, mu_in, keep_rank, w_star, n_keep, mu_out, phi_out, ws, ws_bytes, stream);
        return sober_car_big_device(X, ldx, N, m, mu_in, keep_rank, w_star, n_keep, mu_out, phi_out, ws, ws_bytes, stream);
    }
    if (ws_bytes < sober_car_ws_bytes(N, m)) return SOBER_E_WS;
    hipStream_t st = (hipStream_t)stream;
    double* vws = (double*)ws;
    double* taup = vws + (size_t)m * sober::CAR_NS;
    double* Phi = taup + 128;
    const bool unfused = sober::switches().car_unfused;                                // (same-box A/B of the fused launch)
    const size_t sp_bytes = sizeof(sober::SpSlot) * sober::SP_RING + sober::SP_W * sizeof(int);
    { const int rc = car_pivot_attr(sp_bytes); if (rc != 0) return rc; }
    if (phi_out != nullptr || unfused || mode == SOBER_CAR_SAFE) {
        // three launches, no workgroup waits for another one: nothing here can give up
        CAR_BY_SIZE(m, N, hipLaunchKernelGGL((sober::k_car_bidiag<MS_, CQ_>), dim3(1), dim3(sober::CAR_BT), 0, st, X, ldx, N, m, vws, taup));
        LAUNCH_CHECK();
        hipLaunchKernelGGL(sober::k_car_phi, dim3(sober::CAR_PC / 4), dim3(256), 0, st, vws, taup, N, m, Phi, phi_out);
        LAUNCH_CHECK();
        CAR_PIVOT_BY_SIZE(N, hipLaunchKernelGGL((sober::k_car_pivot_stream<NQ_>), dim3(1), dim3(sober::SP_W * 64), sp_bytes, st, Phi, N, m,
                                                mu_in, keep_rank, w_star, n_keep, mu_out, (const unsigned*)nullptr, (int)sober::switches().car_exact_ratio));
        LAUNCH_CHECK();
        return 0;
    }
    // bidiagonalisation and Phi in ONE launch: workgroup 0 produces the reflectors, the workgroups that land on its XCD
    // (an eighth of the rest: 26 are needed, 40 are offered) accumulate the rows of Phi as the reflectors appear
    static std::atomic<unsigned> epoch_ctr{0};
    const unsigned epoch = (epoch_ctr.fetch_add(1) + 1u) & 0x1FFFFFFu;
    void* comm = (void*)(Phi + (size_t)sober::CAR_NS * sober::CAR_PC + 512);
    constexpr int per_xcd = 40;
    const unsigned spin_limit = sober_car_giveup_forced() ? 0u : sober::CARF_SPIN_LIMIT;
    CAR_BY_SIZE(m, N, hipLaunchKernelGGL((sober::k_car_bidiag_fused<MS_, CQ_>), dim3(1 + 8 * per_xcd), dim3(sober::CAR_BT), 0, st, X, ldx,
                                         N, m, vws, taup, Phi, comm, (unsigned)sober::carf_bytes(0), epoch, spin_limit));
    LAUNCH_CHECK();
    CAR_PIVOT_BY_SIZE(N, hipLaunchKernelGGL((sober::k_car_pivot_stream<NQ_>), dim3(1), dim3(sober::SP_W * 64), sp_bytes, st, Phi, N, m,
                                            mu_in, keep_rank, w_star, n_keep, mu_out, (const unsigned*)comm, (int)sober::switches().car_exact_ratio));
    LAUNCH_CHECK();
    return 0;
}

extern "C" int sober_car_device(const double* X, int ldx, int N, int m, const double* mu_in,
                                int32_t* keep_rank, double* w_star, int32_t* n_keep, double* mu_out,
                                double* phi_out, void* ws, int64_t ws_bytes, void* stream) {
    return sober_car_device_ex(X, ldx, N, m, mu_in, keep_rank, w_star, n_keep, mu_out, phi_out, ws, ws_bytes,
                               SOBER_CAR_DEFAULT, stream);
}

extern "C" int sober_second_elimination(const double* phi, const double* objp, const double* w1, const int32_t* rank1,
                                        int n1, int Nsets, int32_t* keep_rank, double* w_star, int32_t* n_keep,
                                        void* stream) {
    if (!phi || !objp || !w1 || !rank1 || !keep_rank || !w_star || !n_keep || n1 <= 0 || Nsets <= 0) return SOBER_E_ARG;
    if (n1 > 512) return SOBER_E_DIM;
    hipLaunchKernelGGL(sober::k_second_elim<false>, dim3(1), dim3(512), 0, (hipStream_t)stream, phi, objp, w1, rank1, n1, Nsets,
                       keep_rank, w_star, n_keep, (const int32_t*)nullptr, (const int32_t*)nullptr);
    LAUNCH_CHECK();
    return 0;
}

// ... with the null vector and the objective BY SET (sober_null_vector's output; the level's objective column) and the
// first step's verdict read on the device: *n_keep = -2 when it did not leave exactly n1 sets (-1: it gave up)
// (status: sober_null_vector's verdict word, or NULL; != 0 there reads as "not regular": *n_keep = -2)
extern "C" int sober_second_elimination_rows(const double* null_row, const double* obj_row, const double* w1,
                                             const int32_t* rank1, const int32_t* n_keep1, const int32_t* status, int n1,
                                             int Nsets, int32_t* keep_rank, double* w_star, int32_t* n_keep, void* stream) {
    if (!null_row || !obj_row || !w1 || !rank1 || !n_keep1 || !keep_rank || !w_star || !n_keep || n1 <= 0 || Nsets <= 0)
        return SOBER_E_ARG;
    if (n1 > 512) return SOBER_E_DIM;
    hipLaunchKernelGGL(sober::k_second_elim<true>, dim3(1), dim3(512), 0, (hipStream_t)stream, null_row, obj_row, w1, rank1, n1,
                       Nsets, keep_rank, w_star, n_keep, n_keep1, status);
    LAUNCH_CHECK();
    return 0;
}

#ifdef SP_TSTAMPS
extern "C" int sober_debug_sp_stamps(unsigned long long* out260) {
    return (int)hipMemcpyFromSymbol(out260, HIP_SYMBOL(sober::g_sp_stamps), sizeof(unsigned long long) * 260, 0, hipMemcpyDeviceToHost);
}
extern "C" int sober_debug_car_rt(unsigned long long* out8) {
    return (int)hipMemcpyFromSymbol(out8, HIP_SYMBOL(sober::g_car_rt), sizeof(unsigned long long) * 8, 0, hipMemcpyDeviceToHost);
}
extern "C" int sober_debug_sp_segments(unsigned long long* out128) {
    return (int)hipMemcpyFromSymbol(out128, HIP_SYMBOL(sober::g_sp_seg), sizeof(unsigned long long) * 128, 0, hipMemcpyDeviceToHost);
}
#endif
