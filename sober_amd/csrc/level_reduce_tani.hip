// K1-K3 for bit-packed fingerprints (Tanimoto kernel, SOBER/_drug_modelling.py:15-25,37) on the FP4 matrix cores.
//
//   k(x, y) = (<x, y> + eps) / (|x| + |y| - <x, y> + eps),   <x, y> = popcount(x & y) = sum_k x_k y_k,  x_k, y_k in {0, 1}
//
// The dot product of 0/1 vectors is a GEMM whose operands need ONE bit of precision: with every bit expanded to an E2M1
// nibble (1.0 = 0x2, 0.0 = 0x0) and unit block scales, v_mfma_scale_f32_16x16x128_f8f6f4 does 16 x 16 x 128 multiply-adds
// in 16 cycles per SIMD -- two 64-bit words of 16 rows against the same words of 16 candidates per instruction, exact in
// FP32 (sums <= 2048 << 2^24).  Round 2-3 ran this on the INT8 cores (a bit per BYTE, v_mfma_i32_16x16x64_i8): half the
// rate (scripts/fp4_popcount_probe.hip: 8.6 against 4.5 P bit-pair operations per second chip-wide), twice the bytes in
// LDS and in the A fragments -- and the A fragments are what decides how many ROWS a workgroup covers (rows x 2048 bits x
// bytes per bit must fit eight waves' registers): 128 rows at a byte per bit, i.e. six row blocks for the 700-row table,
// each of which fetched, expanded and staged every candidate again (round 3's review: 437 MB fetched against 68 MB of
// candidates at level 0); 256 rows at a nibble per bit: three.  The VALU form (level_reduce.hip: AND + v_bcnt per 32
// bits) serves rows shorter than 512 bits.
//
// Layout.  Workgroup = 8 waves x 32 rows (two 16-row MFMA tiles per wave) x 16 consecutive sets x one element chunk.
// The rows' bits are expanded ONCE into registers (A fragments: 2 tiles x DT/2 k-steps x 16 bytes per lane = 128 VGPRs at
// 2048 bits).  The 32 candidates of a tile arrive bit-packed from HBM (256 B each instead of 16 KB as the reference's
// FP64 0/1 matrix), are expanded to nibbles by all 512 threads (32 bits -> one ds_write_b128) into a double-buffered LDS
// tile with padded rows, and every wave reads its B fragments with ds_read_b128 -- one read feeds both row tiles.  A and B
// use the same bit -> nibble position map, which is all a dot product needs.
// C/D map (dtype independent on gfx950): col = lane & 15 (one candidate = one set), row = 4 (lane >> 4) + reg:
// every lane owns 8 FP64 accumulators (2 tiles x 4 rows) of its set -- no atomics, fixed summation order; the sums are
// the same bits as the INT8 kernel's (exact dot products, the same quotient, the same order over the elements).
#include "common.hpp"

namespace sober {

typedef int int4_t __attribute__((ext_vector_type(4)));
typedef int int8_t_ __attribute__((ext_vector_type(8)));
typedef float float4_t __attribute__((ext_vector_type(4)));

constexpr int LT_RW = 8;     // waves per workgroup (two per SIMD)
constexpr int LT_RT = 2;     // 16-row MFMA tiles per wave: 128 VGPRs of A fragments at 2048 bits (a nibble per bit), nothing in AGPRs
constexpr int LT_SB = 16;    // sets per workgroup = MFMA N
constexpr int LT_TE = 2;     // elements (of 16 candidates) staged per tile
constexpr int LT_ROWS = LT_RW * LT_RT * 16;   // 256 rows per workgroup

// 32 bits -> 32 E2M1 nibbles of 1.0 (0x2) / 0.0 (dword q holds bits 8q .. 8q+7: bit k of the low four -> nibble 2k, of the
// high four -> nibble 2k + 1; the multiplier puts four bits at byte distance, already shifted to the 0x2 position)
__device__ __forceinline__ int4_t expand32(unsigned bits) {
    int4_t r;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned b8 = (bits >> (8 * q)) & 0xFFu;
        const unsigned lo = ((b8 & 0xFu) * 0x00408102u) & 0x02020202u, hi = ((b8 >> 4) * 0x00408102u) & 0x02020202u;
        r[q] = (int)(lo | (hi << 4));
    }
    return r;
}
// 16 x 16 x 128 multiply-adds of E2M1 operands with unit scales (E8M0 127 in every byte of the scale registers)
__device__ __forceinline__ float4_t mfma_f4(const int4_t a, const int4_t b, const float4_t c) {
    const int8_t_ a8 = {a.x, a.y, a.z, a.w, 0, 0, 0, 0}, b8 = {b.x, b.y, b.z, b.w, 0, 0, 0, 0};
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, c, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
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
        // follows from the exact size (the formula the host uses: level_chunks_tani_for); surplus workgroups leave.
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
        const int n_chunks = level_chunks_tani_for(n_rows, e_total, S);
        if ((int)blockIdx.z >= n_chunks) return;
        e_per_chunk = (e_total + n_chunks - 1) / n_chunks;
    }
    static_assert(DT % 2 == 0, "a k-step is two 64-bit words");
    constexpr int KS = DT / 2;                         // k-steps (128 bits each) per fingerprint
    constexpr int ROWB = DT * 32 + 32;                 // bytes per candidate in LDS: row stride = 8 dwords mod 64 banks -> the four lane groups of a ds_read_b128 are conflict-free (a 16-byte pad leaves a 2-way conflict in each)
    constexpr int TE = LT_TE;
    constexpr int NC = TE * SB;                        // candidates per tile
    constexpr int UNITS = NC * DT * 2;                 // 32-bit units per tile
    constexpr int NTH = LT_RW * 64;                    // threads
    constexpr int UPT = (UNITS + NTH - 1) / NTH;       // units per thread
    // dynamic LDS (66 KB at 2048 bits: beyond the static limit): [2][NC * ROWB] candidate nibbles, then [2][NC] weights
    // and [2][NC] popcounts
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    unsigned char (*s_b)[NC * ROWB] = (unsigned char (*)[NC * ROWB])s_dyn;
    double (*s_w)[NC] = (double (*)[NC])(s_dyn + 2 * NC * ROWB);
    double (*s_ny)[NC] = (double (*)[NC])(s_dyn + 2 * NC * ROWB + 2 * NC * sizeof(double));

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lj = lane & 15, lk = lane >> 4;
    // (an XCD-aware one-dimensional grid -- the row blocks of one (set group, chunk) on ONE XCD, as in the FP64 level
    //  kernel -- was measured 16 % SLOWER here: 809 vs 695 us at level 0 of configuration 5)
    const int s0 = bx * SB;
    const int chunk = blockIdx.z;                      // (the slowest dimension: a queued launch's surplus chunks come last)
    const int row0 = blockIdx.y * LT_ROWS + wave * (LT_RT * 16);
    const int e0 = chunk * e_per_chunk;
    const int e1 = min(e0 + e_per_chunk, e_total);

    // A fragments: lane holds bits [32 lk, 32 lk + 32) of every k-step (two words) of row (row0 + 16 t + lj), as nibbles
    int4_t afr[LT_RT][KS];
#pragma unroll
    for (int t = 0; t < LT_RT; ++t) {
        const int r = row0 + 16 * t + lj;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const unsigned long long wv = (r < n_rows) ? rows[(size_t)r * DT + 2 * ks + (lk >> 1)] : 0ull;
            afr[t][ks] = expand32((unsigned)(wv >> (32 * (lk & 1))));
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
    // ONE list position, one validity flag and one base address per thread and tile, then UPT 32-bit units of that
    // candidate at fixed strides (unit j * TPC + cp: the cp-th 32 bits of the j-th run of 512) -- loads and LDS writes
    // with immediate offsets.  (A first form spread a thread's units over UPT different candidates: their positions, flags
    // and addresses were more than half of the kernel's instruction stream.)  The address chain idx -> candidate words is
    // two global round trips, so the index is fetched TWO tiles ahead and the words ONE ahead; every load is
    // unconditional (clamped position, masked afterwards) so that the compiler counts outstanding loads instead of
    // draining them; expansion and LDS writes follow the current tile's MFMAs.  Positions are 32-bit offsets into this
    // launch's index list (count < 2^31).
    constexpr int TPC = NTH / NC;                      // threads per candidate
    static_assert(NTH % NC == 0 && (DT * 2) % TPC == 0 && UPT == DT * 2 / TPC, "unit map");
    unsigned stw[UPT];
    int cpre = 0;
    bool ok_w = false, ok_t = false;
    double tot_acc = 0.0, m_raw = 0.0, wm_raw = 0.0, ny_raw = 0.0;
    const double* wm_ptr = wmul ? wmul : mu;
    const int cnt32 = (int)count;
    const int rel0 = (int)(e_first * S + s0 - pos0);                   // list offset of (element 0, set s0)
    const int tl32 = (int)min(tot_limit - pos0, (int64_t)0x7fffffff);
    const int cq = tid / TPC, cp = tid % TPC;                           // my candidate of the tile, my share of it
    const int qe = cq / SB, qs = cq % SB;
    const bool qs_ok = s0 + qs < S;
    unsigned char* const wr_base = &s_b[0][0] + cq * ROWB + cp * 16;
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
        const unsigned* src_ = (const unsigned*)(cand + (size_t)c_ * DT) + cp;             \
        _Pragma("unroll") for (int u = 0; u < UPT; ++u) stw[u] = src_[u * TPC];            \
        m_raw = mu[c_]; wm_raw = wm_ptr[c_]; ny_raw = cand_norm[c_];                       \
        ok_w = okl_; ok_t = okl_ & (r_ < tl32);                                            \
    }
#define LT_STAGE_WRITE(buf_)                                                               \
    {                                                                                      \
        unsigned char* dst_ = wr_base + (buf_) * (NC * ROWB);                              \
        _Pragma("unroll") for (int u = 0; u < UPT; ++u)                                    \
            *(int4_t*)(dst_ + u * TPC * 16) = expand32(ok_w ? stw[u] : 0u);                \
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
            // B fragments LT_BD k-steps ahead of their MFMAs; the two row tiles are the two independent accumulator chains
            // (left to itself the compiler reads a fragment, waits, and issues its MFMAs: an LDS round trip per k-step)
            float4_t cc[LT_RT];
#pragma unroll
            for (int t = 0; t < LT_RT; ++t) cc[t] = (float4_t){0.f, 0.f, 0.f, 0.f};
            __builtin_amdgcn_sched_barrier(0);
            const unsigned char* bp = &s_b[buf][(te * SB + lj) * ROWB + lk * 16];
            constexpr int LT_BD = KS < 8 ? KS : 8;
            int4_t bq[LT_BD];
#pragma unroll
            for (int j = 0; j < LT_BD; ++j) bq[j] = *(const int4_t*)(bp + j * 64);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int4_t bfr = bq[ks % LT_BD];
                if (ks + LT_BD < KS) bq[ks % LT_BD] = *(const int4_t*)(bp + (ks + LT_BD) * 64);
#pragma unroll
                for (int t = 0; t < LT_RT; ++t) cc[t] = mfma_f4(afr[t][ks], bfr, cc[t]);
            }
            // (the scheduler would pull every read back to just in front of its MFMAs: pin the order -- LT_BD reads, then
            //  LT_RT MFMAs per further read)
            __builtin_amdgcn_sched_group_barrier(0x100, LT_BD, 0);
#pragma unroll
            for (int ks = 0; ks < KS - LT_BD; ++ks) {
                __builtin_amdgcn_sched_group_barrier(0x008, LT_RT, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, LT_RT * LT_BD, 0);
            __builtin_amdgcn_sched_barrier(0);
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
    if (partTot != nullptr && blockIdx.y == 0) {
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
    dim3 grid((S + LT_SB - 1) / LT_SB, (n_rows + LT_ROWS - 1) / LT_ROWS, n_chunks);
    if (leftover == 2) {                                               // + the leftover launch's set groups and chunks
        grid.x += (unsigned)((S_x + LT_SB - 1) / LT_SB);
        if ((unsigned)n_xchunks > grid.z) grid.z = (unsigned)n_xchunks;
    }
    const size_t lds = (size_t)2 * LT_TE * LT_SB * (DT * 32 + 32) + 4 * LT_TE * LT_SB * sizeof(double);
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

// the element chunks of a launch of this kernel (common.hpp: level_chunks_tani_for), and the largest count a launch
// sized for up to e_total_ub elements per set can find on the device
extern "C" int sober_level_chunks_tani(int n_rows, int64_t pos0, int64_t count, int S) {
    if (n_rows <= 0 || pos0 < 0 || count <= 0 || S <= 0) return SOBER_E_ARG;
    return level_chunks_tani_for(n_rows, (pos0 + count + S - 1) / S - pos0 / S, S);
}
extern "C" int sober_level_chunks_tani_cap(int n_rows, int64_t e_total_ub, int S) {
    if (n_rows <= 0 || e_total_ub <= 0 || S <= 0) return SOBER_E_ARG;
    return level_chunks_tani_cap(n_rows, e_total_ub, S);
}

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
// positions and n_chunks_ub chunks (>= sober_level_chunks_tani_cap: the kernel takes the exact count from the level's size)
extern "C" int sober_level_reduce_tani_queued(const void* rows, const double* rows_norm, int n_rows, const void* cand,
                                              const double* cand_norm, int dt, const int32_t* idx, int64_t count_ub, int S,
                                              int S_main, int leftover, const double* mu, const double* wmul,
                                              double outputscale, int n_chunks_ub, double* partG, int ldg, double* partTot,
                                              const int64_t* dR, void* stream) {
    if (!rows || !rows_norm || !cand || !cand_norm || !idx || !mu || !partG || !dR) return SOBER_E_ARG;
    if (n_rows <= 0 || count_ub <= 0 || S <= 0 || S_main <= 0 || n_chunks_ub <= 0 || ldg < S || (!leftover && S != S_main))
        return SOBER_E_ARG;
    if (n_chunks_ub < sober::level_chunks_tani_cap(n_rows, (count_ub + S - 1) / S, S)) return SOBER_E_ARG;
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
    if (n_chunks_ub < sober::level_chunks_tani_cap(n_rows, (count_ub + S - 1) / S, S)) return SOBER_E_ARG;
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
