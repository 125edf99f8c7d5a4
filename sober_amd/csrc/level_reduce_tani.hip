// K1-K3 for bit-packed fingerprints (Tanimoto kernel, SOBER/_drug_modelling.py:15-25,37) on the INT8 matrix cores.
//
//   k(x, y) = (<x, y> + eps) / (|x| + |y| - <x, y> + eps),   <x, y> = popcount(x & y) = sum_k x_k y_k,  x_k, y_k in {0, 1}
//
// The dot product of 0/1 vectors is an integer GEMM: with the bits expanded to bytes, v_mfma_i32_16x16x64_i8 does
// 16 x 16 x 64 multiply-adds in 16 cycles per SIMD -- one 64-bit word of 16 rows against the same word of 16
// candidates per instruction, exact in int32.  The VALU form (level_reduce.hip: AND + v_bcnt per 32 bits, operands
// broadcast from LDS) needs ~130 integer instructions per (row, candidate) pair at 2048 bits; here a pair costs 1/8 of
// an MFMA pass and the vector unit is left with the Tanimoto quotient and the weighted FP64 accumulation.
//
// Layout.  Workgroup = 4 waves x 32 rows (two 16-row MFMA tiles per wave) x 16 consecutive sets x one element chunk.
// The rows' bits are expanded ONCE into registers (A fragments: 2 tiles x DT words x 16 bytes per lane = 256 VGPRs at
// 2048 bits -- one wave per SIMD, the whole register file).  The 16 candidates of an element arrive bit-packed from
// HBM (256 B each instead of 16 KB as the reference's FP64 0/1 matrix), are expanded to bytes by all 256 threads
// (16 bits -> one ds_write_b128) into a double-buffered LDS tile with padded rows, and every wave reads its B
// fragments with ds_read_b128.  A and B use the same bit -> byte position map, which is all a dot product needs.
// C/D map (dtype independent on gfx950): col = lane & 15 (one candidate = one set), row = 4 (lane >> 4) + reg:
// every lane owns 8 FP64 accumulators (2 tiles x 4 rows) of its set -- no atomics, fixed summation order.
#include "common.hpp"

namespace sober {

typedef int int4_t __attribute__((ext_vector_type(4)));

constexpr int LT_RW = 8;     // waves per workgroup (two per SIMD)
constexpr int LT_RT = 1;     // 16-row MFMA tiles per wave: 128 VGPRs of A fragments at 2048 bits, nothing in AGPRs
constexpr int LT_SB = 16;    // sets per workgroup = MFMA N
constexpr int LT_TE = 2;     // elements (of 16 candidates) staged per tile
constexpr int LT_ROWS = LT_RW * LT_RT * 16;   // 128 rows per workgroup

// 16 bits -> 16 bytes of 0 / 1 (dword q holds bits 4q .. 4q+3, one per byte)
__device__ __forceinline__ int4_t expand16(unsigned bits) {
    int4_t r;
#pragma unroll
    for (int q = 0; q < 4; ++q) r[q] = (int)((((bits >> (4 * q)) & 0xFu) * 0x00204081u) & 0x01010101u);
    return r;
}

// (dot + eps) / (|x| + |y| - dot + eps): reciprocal seed + ONE Newton step (relative error ~2^-50), quotient, and one
// correction step with the exact residual -- within 1 ulp of the IEEE quotient at a third of its instructions (four
// quotients per lane and element are the vector unit's main arithmetic here).  Numerator and denominator are positive
// (eps > 0), so the reference's clamp_min_(0) (SOBER/_drug_modelling.py:37) never acts; the output scale is folded into
// the candidate's weight when the tile is staged.
__device__ __forceinline__ double tani_fast(double dot, double nx, double ny) {
    const double eps = 1e-6;
    const double den = ((eps + nx) + ny) - dot;
    double r = __builtin_amdgcn_rcp(den);
    r = fma(fma(-den, r, 1.0), r, r);
    const double num = dot + eps;
    const double q = num * r;
    return fma(fma(-q, den, num), r, q);
}

#define LT_EXPAND(x) expand16(x)
template <int DT>      // 64-bit words per fingerprint
__global__ __launch_bounds__(LT_RW * 64) void k_level_reduce_tani(
    const unsigned long long* __restrict__ rows, const double* __restrict__ rows_norm, int n_rows,
    const unsigned long long* __restrict__ cand, const double* __restrict__ cand_norm,
    const int32_t* __restrict__ idx, int64_t pos0, int64_t count, int S,
    const double* __restrict__ mu, const double* __restrict__ wmul, double os,
    int64_t e_first, int e_total, int e_per_chunk,
    double* __restrict__ partG, int ldg, int col0,
    double* __restrict__ partTot, int64_t tot_limit,
    const int64_t* __restrict__ dR, int S_main, int leftover,
    int S_x, double* __restrict__ partG_x, int ldg_x, double* __restrict__ partTot_x) {
    constexpr int SB = LT_SB;
    int bx = blockIdx.x;                        // my set group
    if (dR != nullptr) {                        // queued level (level_exec.cpp): the launch was sized from an UPPER BOUND of
        // the live positions; the exact number R sits in device memory.  leftover = 0: positions [0, R), set masses over
        // [0, E S) (S = S_main); leftover = 1: the leftover positions [E S_main, R) over S pseudo-sets.  The chunk count
        // follows from the exact size (the formula the host uses: level_chunks_for); surplus workgroups leave.
        const int64_t R = __hip_atomic_load(dR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (R <= S_main) return;
        const int64_t ES = (R / S_main) * S_main;
        if (leftover == 2) {                    // both placements in one grid: the set groups behind the main launch's are
            const int g_main = (S_main + SB - 1) / SB;                 // the leftover launch's (S_x pseudo-sets, its own sums)
            if (bx >= g_main) { leftover = 1; bx -= g_main; S = S_x; partG = partG_x; ldg = ldg_x; partTot = partTot_x; }
            else leftover = 0;
        }
        if (!leftover) { count = R; tot_limit = ES; }
        else { idx += ES; count = R - ES; tot_limit = count; }
        if (count <= 0) return;
        pos0 = 0; e_first = 0;
        e_total = (int)((count + S - 1) / S);
        const int n_chunks = level_chunks_for(n_rows, e_total, S);
        if ((int)blockIdx.y >= n_chunks) return;
        e_per_chunk = (e_total + n_chunks - 1) / n_chunks;
    }
    constexpr int ROWB = DT * 64 + 32;                 // bytes per candidate in LDS: row stride = 8 dwords mod 64 banks -> the four lane groups of a ds_read_b128 are conflict-free (a 16-byte pad leaves a 2-way conflict in each)
    constexpr int TE = LT_TE;
    constexpr int NC = TE * SB;                        // candidates per tile
    constexpr int UNITS = NC * DT * 4;                 // 16-bit units per tile
    constexpr int NTH = LT_RW * 64;                    // threads
    constexpr int UPT = (UNITS + NTH - 1) / NTH;       // units per thread
    // dynamic LDS (66 KB at 2048 bits: beyond the static limit): [2][SB * ROWB] candidate bytes, then [2][SB] weights
    // and [2][SB] popcounts
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    unsigned char (*s_b)[NC * ROWB] = (unsigned char (*)[NC * ROWB])s_dyn;
    double (*s_w)[NC] = (double (*)[NC])(s_dyn + 2 * NC * ROWB);
    double (*s_ny)[NC] = (double (*)[NC])(s_dyn + 2 * NC * ROWB + 2 * NC * sizeof(double));

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lj = lane & 15, lk = lane >> 4;
    // (an XCD-aware one-dimensional grid -- the row blocks of one (set group, chunk) on ONE XCD, as in the FP64 level
    //  kernel -- was measured 16 % SLOWER here: 809 vs 695 us at level 0 of configuration 5)
    const int s0 = bx * SB;
    const int chunk = blockIdx.y;
    const int row0 = blockIdx.z * LT_ROWS + wave * (LT_RT * 16);
    const int e0 = chunk * e_per_chunk;
    const int e1 = min(e0 + e_per_chunk, e_total);

    // A fragments: lane holds bits [16 lk, 16 lk + 16) of every word of row (row0 + 16 t + lj), as bytes
    int4_t afr[LT_RT][DT];
#pragma unroll
    for (int t = 0; t < LT_RT; ++t) {
        const int r = row0 + 16 * t + lj;
#pragma unroll
        for (int ks = 0; ks < DT; ++ks) {
            const unsigned long long wv = (r < n_rows) ? rows[(size_t)r * DT + ks] : 0ull;
            afr[t][ks] = expand16((unsigned)(wv >> (16 * lk)) & 0xFFFFu);
        }
    }
    double nxr[LT_RT][4];
#pragma unroll
    for (int t = 0; t < LT_RT; ++t)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int r = row0 + 16 * t + 4 * lk + v;
            nxr[t][v] = (r < n_rows) ? rows_norm[r] : 0.0;
        }
    double acc[LT_RT][4];
#pragma unroll
    for (int t = 0; t < LT_RT; ++t)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[t][v] = 0.0;

    // staging of one tile (TE elements of 16 consecutive list positions each).  TPC = 16 threads share a candidate:
    // ONE list position, one validity flag and one base address per thread and tile, then UPT 16-bit units of that
    // candidate at fixed strides (unit j * TPC + cp: word 4 j + cp / 4, quarter cp % 4) -- loads and LDS writes with
    // immediate offsets.  (A first form spread a thread's units over UPT different candidates: their positions, flags
    // and addresses were more than half of the kernel's instruction stream.)  The address chain idx -> candidate words is
    // two global round trips, so the index is fetched TWO tiles ahead and the words ONE ahead; every load is
    // unconditional (clamped position, masked afterwards) so that the compiler counts outstanding loads instead of
    // draining them; expansion and LDS writes follow the current tile's MFMAs.  Positions are 32-bit offsets into this
    // launch's index list (count < 2^31).
    constexpr int TPC = NTH / NC;                      // threads per candidate
    static_assert(NTH % NC == 0 && TPC % 4 == 0 && (DT * 4) % TPC == 0 && UPT == DT * 4 / TPC, "unit map");
    unsigned long long stw[UPT];
    int cpre = 0;
    bool ok_w = false, ok_t = false;
    double tot_acc = 0.0, m_raw = 0.0, wm_raw = 0.0, ny_raw = 0.0;
    const double* wm_ptr = wmul ? wmul : mu;
    const int cnt32 = (int)count;
    const int rel0 = (int)(e_first * S + s0 - pos0);                   // list offset of (element 0, set s0)
    const int tl32 = (int)min(tot_limit - pos0, (int64_t)0x7fffffff);
    const int cq = tid / TPC, cp = tid % TPC;                           // my candidate of the tile, my share of it
    const int qe = cq / SB, qs = cq % SB;
    const int wsel = cp / 4, part = cp % 4;
    const bool qs_ok = s0 + qs < S;
    unsigned char* const wr_base = &s_b[0][0] + cq * ROWB + wsel * 64 + part * 16;
#define LT_REL(e_, R, OK)                                                                  \
    const int R = rel0 + ((e_) + qe) * S + qs;                                             \
    const bool OK = qs_ok & ((e_) + qe < e1) & (R >= 0) & (R < cnt32);
#define LT_PREFETCH_IDX(e_)                                                                \
    {                                                                                      \
        LT_REL(e_, r_, okp_)                                                               \
        cpre = idx[okp_ ? r_ : 0];                                                         \
    }
#define LT_STAGE_LOAD(e_)                                                                  \
    {                                                                                      \
        LT_REL(e_, r_, okl_)                                                               \
        const int c_ = okl_ ? cpre : 0;                                                    \
        const unsigned long long* src_ = cand + (size_t)c_ * DT + wsel;                    \
        _Pragma("unroll") for (int u = 0; u < UPT; ++u) stw[u] = src_[u * (TPC / 4)];      \
        m_raw = mu[c_]; wm_raw = wm_ptr[c_]; ny_raw = cand_norm[c_];                       \
        ok_w = okl_; ok_t = okl_ & (r_ < tl32);                                            \
    }
#define LT_STAGE_WRITE(buf_)                                                               \
    {                                                                                      \
        unsigned char* dst_ = wr_base + (buf_) * (NC * ROWB);                              \
        _Pragma("unroll") for (int u = 0; u < UPT; ++u)                                    \
            *(int4_t*)(dst_ + u * (TPC / 4) * 64) =                                        \
                LT_EXPAND(ok_w ? (unsigned)(stw[u] >> (16 * part)) & 0xFFFFu : 0u);        \
        if (cp == 0) {                                                                     \
            s_w[buf_][cq] = ok_w ? (wmul ? m_raw * wm_raw : m_raw) * os : 0.0;             \
            s_ny[buf_][cq] = ok_w ? ny_raw : 0.0;                                          \
            tot_acc += ok_t ? m_raw : 0.0;                                                 \
        }                                                                                  \
    }

    int buf = 0;
    LT_PREFETCH_IDX(e0)
    __builtin_amdgcn_s_waitcnt(0);
    LT_STAGE_LOAD(e0)
    LT_PREFETCH_IDX(e0 + TE)
    LT_STAGE_WRITE(0)
    __syncthreads();
    const bool rows_live = row0 < n_rows;
    for (int e = e0; e < e1; e += TE) {
        LT_STAGE_LOAD(e + TE)                                           // (padding when there is no next tile)
        LT_PREFETCH_IDX(e + 2 * TE)
        const int te_cnt = rows_live ? min(TE, e1 - e) : 0;
        for (int te = 0; te < te_cnt; ++te) {
            // B fragments LT_BD k-steps ahead of their MFMAs and two accumulators per tile: with a single accumulator and
            // the compiler's own order (two reads, wait, two dependent MFMAs) an element was a chain of 16 LDS round trips
            int4_t cc[LT_RT], cd[LT_RT];
#pragma unroll
            for (int t = 0; t < LT_RT; ++t) { cc[t] = (int4_t){0, 0, 0, 0}; cd[t] = cc[t]; }
            __builtin_amdgcn_sched_barrier(0);
            const unsigned char* bp = &s_b[buf][(te * SB + lj) * ROWB + lk * 16];
            constexpr int LT_BD = 8;
            int4_t bq[LT_BD];
#pragma unroll
            for (int j = 0; j < LT_BD && j < DT; ++j) bq[j] = *(const int4_t*)(bp + j * 64);
#pragma unroll
            for (int ks = 0; ks < DT; ++ks) {
                const int4_t bfr = bq[ks % LT_BD];
                if (ks + LT_BD < DT) bq[ks % LT_BD] = *(const int4_t*)(bp + (ks + LT_BD) * 64);
#pragma unroll
                for (int t = 0; t < LT_RT; ++t)
                {
                    if (ks & 1) cd[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(afr[t][ks], bfr, cd[t], 0, 0, 0);
                    else cc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(afr[t][ks], bfr, cc[t], 0, 0, 0);
                }
            }
            // (the scheduler would pull every read back to just in front of its MFMA: pin the order -- LT_BD reads, then
            //  one MFMA per further read)
            __builtin_amdgcn_sched_group_barrier(0x100, LT_BD, 0);
#pragma unroll
            for (int ks = 0; ks < DT - LT_BD; ++ks) {
                __builtin_amdgcn_sched_group_barrier(0x008, LT_RT, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, LT_RT * LT_BD, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < LT_RT; ++t) cc[t] += cd[t];
            const double w = s_w[buf][te * SB + lj], ny = s_ny[buf][te * SB + lj];
#pragma unroll
            for (int t = 0; t < LT_RT; ++t)
#pragma unroll
                for (int v = 0; v < 4; ++v)
                    acc[t][v] = fma(tani_fast((double)cc[t][v], nxr[t][v], ny), w, acc[t][v]);
        }
        LT_STAGE_WRITE(buf ^ 1)
        __syncthreads();
        buf ^= 1;
    }
#undef LT_STAGE_LOAD
#undef LT_STAGE_WRITE
#undef LT_PREFETCH_IDX
#undef LT_REL

    if (s0 + lj < S) {
#pragma unroll
        for (int t = 0; t < LT_RT; ++t)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int row = row0 + 16 * t + 4 * lk + v;
                if (row < n_rows) partG[((size_t)chunk * n_rows + row) * ldg + col0 + s0 + lj] = acc[t][v];
            }
    }
    if (partTot != nullptr && blockIdx.z == 0) {
        __syncthreads();
        if (cp == 0) s_w[0][cq] = tot_acc;                              // (the weight buffers are free now)
        __syncthreads();
        if (tid < SB && s0 + tid < S) {
            double tt = 0.0;
#pragma unroll
            for (int te = 0; te < TE; ++te) tt += s_w[0][te * SB + tid];
            partTot[(size_t)chunk * ldg + col0 + s0 + tid] = tt;
        }
    }
}

template <int DT>
static int launch_lt(const void* rows, const double* rows_norm, int n_rows, const void* cand, const double* cand_norm,
                     const int32_t* idx, int64_t pos0, int64_t count, int S, const double* mu, const double* wmul,
                     double os, int n_chunks, double* partG, int ldg, int col0, double* partTot, int64_t tot_limit,
                     hipStream_t st, const int64_t* dR = nullptr, int S_main = 0, int leftover = 0, int S_x = 0, int n_xchunks = 0,
                     double* partG_x = nullptr, int ldg_x = 0, double* partTot_x = nullptr) {
    const int64_t e_first = pos0 / S;
    const int e_total = (int)((pos0 + count + S - 1) / S - e_first);
    const int e_per_chunk = (e_total + n_chunks - 1) / n_chunks;
    dim3 grid((S + LT_SB - 1) / LT_SB, n_chunks, (n_rows + LT_ROWS - 1) / LT_ROWS);
    if (leftover == 2) {                                               // + the leftover launch's set groups and chunks
        grid.x += (unsigned)((S_x + LT_SB - 1) / LT_SB);
        if ((unsigned)n_xchunks > grid.y) grid.y = (unsigned)n_xchunks;
    }
    const size_t lds = (size_t)2 * LT_TE * LT_SB * (DT * 64 + 32) + 4 * LT_TE * LT_SB * sizeof(double);
    HIP_TRY(hipFuncSetAttribute((const void*)k_level_reduce_tani<DT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    SOBER_LAUNCH_TIMED((k_level_reduce_tani<DT>), grid, dim3(LT_RW * 64), lds, st, (const unsigned long long*)rows,
                       rows_norm, n_rows, (const unsigned long long*)cand, cand_norm, idx, pos0, count, S, mu, wmul, os,
                       e_first, e_total, e_per_chunk, partG, ldg, col0, partTot, tot_limit, dR, S_main, leftover, S_x, partG_x,
                       ldg_x, partTot_x);
    LAUNCH_CHECK();
    return 0;
}

}  // namespace sober

using namespace sober;

extern "C" int sober_level_reduce_tani_supported(int dt) { return (dt == 8 || dt == 16 || dt == 32) ? 1 : 0; }

extern "C" int sober_level_reduce_tani(const void* rows, const double* rows_norm, int n_rows, const void* cand,
                                       const double* cand_norm, int dt, const int32_t* idx, int64_t pos0,
                                       int64_t count, int S, const double* mu, const double* wmul, double outputscale,
                                       int n_chunks, double* partG, int ldg, int col0, double* partTot,
                                       int64_t tot_limit, void* stream) {
    if (!rows || !rows_norm || !cand || !cand_norm || !idx || !mu || !partG) return SOBER_E_ARG;
    if (n_rows <= 0 || pos0 < 0 || count <= 0 || S <= 0 || n_chunks <= 0 || ldg < col0 + S) return SOBER_E_ARG;
    if (n_chunks > (pos0 + count + S - 1) / S - pos0 / S) return SOBER_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    switch (dt) {
        case 8: return launch_lt<8>(rows, rows_norm, n_rows, cand, cand_norm, idx, pos0, count, S, mu, wmul, outputscale,
                                    n_chunks, partG, ldg, col0, partTot, tot_limit, st);
        case 16: return launch_lt<16>(rows, rows_norm, n_rows, cand, cand_norm, idx, pos0, count, S, mu, wmul, outputscale,
                                      n_chunks, partG, ldg, col0, partTot, tot_limit, st);
        case 32: return launch_lt<32>(rows, rows_norm, n_rows, cand, cand_norm, idx, pos0, count, S, mu, wmul, outputscale,
                                      n_chunks, partG, ldg, col0, partTot, tot_limit, st);
        default: return SOBER_E_DIM;
    }
}

// queued form (see sober_level_loop): the level size is read from device memory; the launch is sized for count_ub
// positions and n_chunks_ub chunks (sober_level_chunks_cap)
extern "C" int sober_level_reduce_tani_queued(const void* rows, const double* rows_norm, int n_rows, const void* cand,
                                              const double* cand_norm, int dt, const int32_t* idx, int64_t count_ub, int S,
                                              int S_main, int leftover, const double* mu, const double* wmul,
                                              double outputscale, int n_chunks_ub, double* partG, int ldg, double* partTot,
                                              const int64_t* dR, void* stream) {
    if (!rows || !rows_norm || !cand || !cand_norm || !idx || !mu || !partG || !dR) return SOBER_E_ARG;
    if (n_rows <= 0 || count_ub <= 0 || S <= 0 || S_main <= 0 || n_chunks_ub <= 0 || ldg < S || (!leftover && S != S_main))
        return SOBER_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    switch (dt) {
        case 8: return launch_lt<8>(rows, rows_norm, n_rows, cand, cand_norm, idx, 0, count_ub, S, mu, wmul, outputscale,
                                    n_chunks_ub, partG, ldg, 0, partTot, 0, st, dR, S_main, leftover);
        case 16: return launch_lt<16>(rows, rows_norm, n_rows, cand, cand_norm, idx, 0, count_ub, S, mu, wmul, outputscale,
                                      n_chunks_ub, partG, ldg, 0, partTot, 0, st, dR, S_main, leftover);
        case 32: return launch_lt<32>(rows, rows_norm, n_rows, cand, cand_norm, idx, 0, count_ub, S, mu, wmul, outputscale,
                                      n_chunks_ub, partG, ldg, 0, partTot, 0, st, dR, S_main, leftover);
        default: return SOBER_E_DIM;
    }
}

// both placements of a queued level in ONE launch: the leftover launch's set groups ride behind the main ones
extern "C" int sober_level_reduce_tani_queued_pair(const void* rows, const double* rows_norm, int n_rows, const void* cand,
                                                   const double* cand_norm, int dt, const int32_t* idx, int64_t count_ub,
                                                   int S, int n_xcols, const double* mu, const double* wmul,
                                                   double outputscale, int n_chunks_ub, double* partG, int ldg,
                                                   double* partTot, int n_xchunks_ub, double* extraG, double* extraTot,
                                                   const int64_t* dR, void* stream) {
    if (!rows || !rows_norm || !cand || !cand_norm || !idx || !mu || !partG || !dR || !extraG || !extraTot) return SOBER_E_ARG;
    if (n_rows <= 0 || count_ub <= 0 || S <= 1 || n_xcols <= 0 || n_chunks_ub <= 0 || n_xchunks_ub <= 0 || ldg < S) return SOBER_E_ARG;
    hipStream_t st = (hipStream_t)stream;
#define LT_PAIR(T) launch_lt<T>(rows, rows_norm, n_rows, cand, cand_norm, idx, 0, count_ub, S, mu, wmul, outputscale, n_chunks_ub, \
                                partG, ldg, 0, partTot, 0, st, dR, S, 2, n_xcols, n_xchunks_ub, extraG, n_xcols, extraTot)
    switch (dt) {
        case 8: return LT_PAIR(8);
        case 16: return LT_PAIR(16);
        case 32: return LT_PAIR(32);
        default: return SOBER_E_DIM;
    }
#undef LT_PAIR
}
