// K1-K3, matrix-core variant for the continuous kernels (RBF, Matern-5/2).
//
//   k(x, y) depends on  a = -1/2 |x~ - y~|^2  =  x~.y~ - 1/2|x~|^2 - 1/2|y~|^2        (x~ = (x - c)/l)
//
// With augmented points  X' = L [x~, -1/2|x~|^2, 1, 0..]  and  Y' = [y~, 1, -1/2|y~|^2, 0..]  (the two extra slots live
// in the zero padding of the 4-aligned dimension; L = 256/ln2), X'.Y' = a L is a plain GEMM with
// K = DA = roundup4(d + 2): three v_mfma_f64_16x16x4_f64 per 16x16 tile at d = 10.  The matrix pipe produces the
// exponent argument already scaled for the table, the vector pipe does the exponential and the weighted accumulation.
//
// exp(): no v_exp_f64 exists.  n = rint(a L) by the magic-number add, 256-entry table 2^(j/256) (exp_table.inc,
// correctly rounded constants), degree-4 polynomial, the power of two as one integer add on the table value's high
// dword: 8 FP64 + 4 integer instructions per value, max relative error ~2e-16 (exp_tab4 below).
//
// Wave tile: 64 rows (4 MFMA row tiles) x 16 sets; lane l holds column j = l & 15 (one candidate -> one set) and rows
// (l >> 4) + 4*reg of each tile.  A wave is autonomous (k_level_reduce_wave): it feeds its own B fragments from L2,
// no LDS staging, no barrier in the element loop.  Accumulators stay in VGPRs over the wave's element range: fixed
// order, no atomics.
#include "common.hpp"
#include <cstdlib>

namespace sober {

#include "kern_exp.inc"

// ---- wave-autonomous variant ----------------------------------------------------------------------------------
// The same arithmetic without a workgroup in the inner loop.  A wave owns a 64-row x 16-set tile over a contiguous
// range of elements and feeds its own B fragments straight from L2: lane (lj, lk) reads KT CONTIGUOUS doubles of its
// candidate's augmented row (the contraction index is permuted, k = lk KT + ks, on both operands), its weight and its
// list index -- index three elements ahead, row and weight two ahead, MFMAs one ahead of the exponentials.  No LDS
// staging, no barrier in the loop.
// On gfx950 v_mfma_f64 occupies the vector ALU for its 64 cycles -- neither FP64 nor integer vector instructions of
// any wave of the SIMD overlap with it (scripts/dp_rate_probe.hip) -- so the kernel is bound by its instruction count
// (12 MFMAs + ~270 vector instructions per element and wave at d = 10), not by latencies: two waves per SIMD are enough,
// and what matters is that ALL waves of a launch are resident at once and evenly loaded.  Hence the split of
// common.hpp: one line of waves, `wpt` per tile with element ranges that differ by at most one element, cut into
// workgroups of 4 that may straddle two tiles; the waves of one tile inside a workgroup are summed through LDS in wave
// order (fixed order: bit-reproducible) and leave ONE partial sum, so a tile leaves <= 5 in HBM instead of 13.
// XCD-aware: consecutive workgroup ids go round the 8 XCDs, so id -> (xcd, slot) and XCD x takes the x-th CONTIGUOUS
// eighth of the line -- the row tiles of at most three set groups: a candidate row is fetched by one or two L2s,
// not by all eight.
template <int KIND, int KT>
__global__ __launch_bounds__(SOBER_LW_W * 64, 8 / SOBER_LW_W) void k_level_reduce_wave(
    const double* __restrict__ rows, int n_rows, const double* __restrict__ cand,
    const int32_t* __restrict__ idx, int64_t pos0, int64_t count, int S,
    const double* __restrict__ mu, const double* __restrict__ wmul, double os,
    int64_t e_first, int e_total,
    double* __restrict__ partG, int ldg, int col0,
    double* __restrict__ partTot, int64_t tot_limit,
    const int64_t* __restrict__ dR, int S_main, int leftover,
    int S_x, double* __restrict__ partG_x, int ldg_x, double* __restrict__ partTot_x, int grid_main, int wpt_ov) {
    constexpr int DA = 4 * KT;
    constexpr int W = SOBER_LW_W;
#ifdef LW_STAMPS      // diagnostic build (scripts/level_stamps.py): s_memrealtime (100 MHz, one origin for the whole chip) of every wave at eight points -> partG, slot 6 on
    unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long* const st_out_ = (unsigned long long*)(partG + (size_t)6 * n_rows * ldg) + ((size_t)blockIdx.x * W + (threadIdx.x >> 6)) * 8;
#define LW_STAMP(K) do { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); st_[K] = t_; } while (0)
#define LW_STAMP_V(K, VAL) do { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_), "+v"(VAL) :: "memory"); st_[K] = t_; } while (0)
#define LW_STAMP_FLUSH() do { if ((threadIdx.x & 63) == 0) for (int k_ = 0; k_ < 8; ++k_) st_out_[k_] = st_[k_]; } while (0)
    LW_STAMP(0);
#else
#define LW_STAMP(K) do { } while (0)
#define LW_STAMP_V(K, VAL) do { } while (0)
#define LW_STAMP_FLUSH() do { } while (0)
#endif
    __shared__ double s_T[EXP_TAB];              // 2^(j/256), exponent-adjusted (exp_tab_entry)
    __shared__ double s_tot[W * 16];
    __shared__ double s_red[W * 16 * 64];

    int bid = blockIdx.x, gdim = gridDim.x;     // my place in my launch, and its size
    if (dR != nullptr) {                        // queued level (level_exec.cpp): the launch was sized from an UPPER BOUND of
        // the live positions; the exact number R sits in device memory (written by the previous level's update).
        // leftover = 0: the main launch over positions [0, R) (S = S_main); leftover = 1: positions [E S_main, R)
        // spread over S pseudo-sets; leftover = 2: BOTH in one grid -- the workgroups behind the first grid_main are
        // the leftover launch (S_x pseudo-sets, its own partial sums), same arithmetic as two launches.
        const int64_t R = __hip_atomic_load(dR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (R <= S_main) return;
        const int64_t ES = (R / S_main) * S_main;
        if (leftover == 2) {
            if (bid >= grid_main) {
                leftover = 1; bid -= grid_main; gdim -= grid_main;
                S = S_x; partG = partG_x; ldg = ldg_x; partTot = partTot_x;
            } else {
                leftover = 0; gdim = grid_main;
            }
        }
        if (!leftover) { count = R; tot_limit = ES; }
        else { idx += ES; count = R - ES; tot_limit = count; }
        if (count <= 0) return;
        pos0 = 0; e_first = 0;
        e_total = (int)((count + S - 1) / S);
    }
#ifdef LW_STAMPS
    { unsigned hw_, xc_; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_)); asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xc_));
      st_[1] = ((unsigned long long)xc_ << 32) | hw_; }          // where the wave runs
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // (scalar: the element loop is uniform)
    const int lj = lane & 15, lk = lane >> 4;
    const int G = (S + 15) >> 4, RT = (n_rows + 63) >> 6;
    // (wpt_ov > 0: the class launch of level_exec.cpp -- S is then 2^D x the level's sets and the line of waves may be
    //  several rounds long, level_class_wpt)
    const int wpt = wpt_ov > 0 ? wpt_ov : level_wave_wpt(n_rows, e_total, S);
    const int n_slots = level_wave_slots(wpt);
    const int n_waves = G * RT * wpt;
    const int per_xcd = gdim >> 3;                                      // (the grid is a multiple of 8)
    const int q = (bid & 7) * per_xcd + (bid >> 3);                     // my place in the line of workgroups
    if (q * W >= n_waves) return;                                       // (whole workgroup)
    const int v = q * W + wave;
    const bool live = v < n_waves;
    const int tile = live ? v / wpt : -1, sub = live ? v - tile * wpt : 0;
    const int g = live ? tile / RT : 0, rt = live ? tile - g * RT : 0;
    const int e_base = e_total / wpt, e_rem = e_total - e_base * wpt;   // element ranges differ by at most one
    const int ea = live ? sub * e_base + min(sub, e_rem) : 0;
    const int eb = live ? ea + e_base + (sub < e_rem ? 1 : 0) : 0;
    const int row0 = rt * 64;
    const int s = g * 16 + lj;
    const bool s_ok = s < S;

    static_assert(SOBER_LW_W * 64 >= EXP_TAB, "one table entry per thread");
    if (tid < EXP_TAB) s_T[tid] = exp_tab_entry(tid);

    double afr[4][KT];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int r = min(row0 + 16 * t + lj, n_rows - 1);             // (rows past the table: computed, never stored)
#pragma unroll
        for (int ks = 0; ks < KT; ++ks) afr[t][ks] = rows[(size_t)r * DA + lk * KT + ks];
    }
    double4_t acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (double4_t){0.0, 0.0, 0.0, 0.0};
    double tot_acc = 0.0;

    const double* wm_ptr = wmul ? wmul : mu;
    // list position of my candidate of element e, relative to the first position of this launch: 32-bit (the list
    // holds int32 indices).  Without a candidate (set past S, element past my range, position outside the list) the
    // loads go to a clamped position -- some valid candidate, whose finite kernel value meets a zero weight: every
    // load is unconditional, so the compiler can count what is in flight.  (Lanes of a set past S compute on such
    // stand-ins throughout and are never stored.)
    const int rel0 = s - (int)(pos0 - e_first * S);
    const int cnt32 = (int)count;
    const int tl32 = (int)max((int64_t)INT32_MIN, min((int64_t)INT32_MAX, tot_limit - pos0));
#define LW_REL(e, R, OK)                                                                   \
    const int R = (e) * S + rel0;                                                          \
    const bool OK = s_ok && (e) < eb && R >= 0 && R < cnt32;
    // (addresses are clamped with integer min/max, not selected on OK: a select on a loaded value's address turns
    // into a branch, and with control flow in the loop the waits degrade to "everything")
#define LW_IDX(e, C) C = idx[min(max((e) * S + rel0, 0), cnt32 - 1)];
#define LW_LOAD(e, C, Q)                                                                   \
    {                                                                                      \
        LW_REL(e, rl_, okl_)                                                               \
        m[Q] = mu[C];                                                                      \
        wv[Q] = wm_ptr[C];                                                                 \
        const double* src_ = cand + (size_t)C * DA + lk * KT;                              \
        _Pragma("unroll") for (int ks = 0; ks < KT; ++ks) b[Q][ks] = src_[ks];             \
        ok[Q] = okl_;                                                                      \
        okt[Q] = okl_ && rl_ < tl32;                                                       \
    }
    // matrix-core part of one element: its operands were loaded one half-step ago
#define LW_MFMA(Q)                                                                         \
    {                                                                                      \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) cc[Q][t] = (double4_t){0.0, 0.0, 0.0, 0.0}; \
        _Pragma("unroll") for (int ks = 0; ks < KT; ++ks)                                  \
            _Pragma("unroll") for (int t = 0; t < 4; ++t)                                  \
                cc[Q][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(afr[t][ks], b[Q][ks], cc[Q][t], 0, 0, 0); \
        w[Q] = ok[Q] ? (wmul ? m[Q] * wv[Q] : m[Q]) * os : 0.0;                            \
        tot_acc += okt[Q] ? m[Q] : 0.0;                                                    \
    }
    // one element: operands of e + 2 requested, index of e + 3 requested, MFMAs of e + 1, exponentials of e
#define LW_HALF(e, Q, QN)                                                                  \
    {                                                                                      \
        asm volatile("" : "+v"(cF));       /* the index is awaited HERE, not where it was requested */ \
        int cG_;                                                                           \
        LW_IDX((e) + 3, cG_)                                                               \
        LW_LOAD((e) + 2, cF, Q)                                                            \
        cF = cG_;                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        LW_MFMA(QN)                                                                        \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                    \
            double k_[4];                                                                  \
            const double4_t c_ = ccv[Q][t];                                                \
            kern_from_arg4<KIND>(c_, s_T, k_);                                             \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) acc[t][r] = fma(k_[r], wcv[Q], acc[t][r]); \
            /* the vector work of a row tile stays inside its tile (four chains are enough to keep the pipe fed), the matrix */ \
            /* instructions may still cross: with all 16 values of an element in flight three instantiations spilled -- RBF */ \
            /* KT = 4 / 5, Matern KT = 3: 12 / 148 / 56 bytes of scratch, reloaded in the hot loop; d = 15..18 ran 6 % SLOWER */ \
            /* than d = 20 -- and the others sat at 244-256 registers.  Round 5: -11..-14 % on the spilling ones, -0.4..-4 % */ \
            /* on the rest (scripts/level_kernel_sweep.py), same arithmetic */                 \
            __builtin_amdgcn_sched_barrier(0x0008);                                        \
        }                                                                                  \
        LW_INTERLEAVE                                                                      \
    }

    // LW_SCHED = n: ask the scheduler for one MFMA per n vector instructions instead of its own order (which
    // leaves the MFMAs of an element in two or three clumps)
#if defined(LW_SCHED) && LW_SCHED > 0
#define LW_INTERLEAVE                                                                      \
    _Pragma("unroll") for (int i_ = 0; i_ < 4 * KT; ++i_) {                                \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                 \
        __builtin_amdgcn_sched_group_barrier(0x002, LW_SCHED, 0);                          \
    }
#else
#define LW_INTERLEAVE
#endif
    double4_t cc[2][4];
    double w[2];
    double b[2][KT], m[2], wv[2];
    bool ok[2], okt[2];
    int cA = 0, cB = 0, cF = 0;
    LW_IDX(ea, cA)
    LW_IDX(ea + 1, cB)
    LW_IDX(ea + 2, cF)
    __syncthreads();                                   // the table (the only barrier before the final sum)
    LW_STAMP(2);
    LW_STAMP_V(3, afr[3][KT - 1]);
    LW_LOAD(ea, cA, 0)
    LW_LOAD(ea + 1, cB, 1)
    LW_STAMP_V(4, b[1][KT - 1]);
    LW_MFMA(0)
    // (the exponentials of a half-step read cc / w of the element the PREVIOUS half-step multiplied)
#define ccv cc
#define wcv w
    int e = ea;
    for (; e + 1 < eb; e += 2) {
        LW_HALF(e, 0, 1)
        LW_HALF(e + 1, 1, 0)
    }
    if (e < eb) LW_HALF(e, 0, 1)
    LW_STAMP_V(5, acc[3][3]);
#undef ccv
#undef wcv
#undef LW_REL
#undef LW_IDX
#undef LW_LOAD
#undef LW_MFMA
#undef LW_HALF
#undef LW_INTERLEAVE

    // the waves of one tile inside the workgroup hold consecutive element ranges: the first of them sums the run in
    // wave order and stores it as the tile's partial sum number (this workgroup - the tile's first workgroup)
    __shared__ int s_tile[W];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) s_red[(wave * 16 + t * 4 + r) * 64 + lane] = acc[t][r];
    if (lk == 0) s_tot[wave * 16 + lj] = tot_acc;
    if (lane == 0) s_tile[wave] = tile;
    __syncthreads();
    LW_STAMP(6);
    if (!live || (wave > 0 && s_tile[wave - 1] == tile)) { LW_STAMP_FLUSH(); return; }      // not the first wave of its run
    for (int w = wave + 1; w < W && s_tile[w] == tile; ++w) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[t][r] += s_red[(w * 16 + t * 4 + r) * 64 + lane];
        tot_acc += s_tot[w * 16 + lj];
    }
    const int slot = q - (tile * wpt) / W;
    const int last_q = (tile * wpt + wpt - 1) / W;                     // the workgroup of the tile's last wave
    if (!s_ok) { LW_STAMP_FLUSH(); return; }
    // (a tile whose waves touch fewer workgroups than n_slots: its last run also clears the slots nobody writes)
    for (int sl = slot; sl < (q == last_q ? n_slots : slot + 1); ++sl) {
        const bool mine = sl == slot;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + 16 * t + lk + 4 * r;
                if (row < n_rows) partG[((size_t)sl * n_rows + row) * ldg + col0 + s] = mine ? acc[t][r] : 0.0;
            }
        if (partTot != nullptr && rt == 0 && lk == 0) partTot[(size_t)sl * ldg + col0 + s] = mine ? tot_acc : 0.0;
    }
    LW_STAMP(7);
    LW_STAMP_FLUSH();
}

// points -> augmented, centred, scaled rows.  side 1 (pool): [y~, 1, -|y~|^2/2, 0..];
// side 0 (row table): L * [x~, -|x~|^2/2, 1, 0..] with L = 256/ln2, so that X'.Y' = -|x~ - y~|^2/2 * L is
// directly the table-exp's scaled argument
// One 16-lane DPP row per point (lane l holds coordinates l and l + 16: DA <= 32): the row of X is read and the
// augmented row written as contiguous 128-byte segments, |x~|^2 is an in-row DPP reduction.
template <int CTRL>
__device__ __forceinline__ double aug_dpp(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);          // (no `old` operand: every lane is written)
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__global__ __launch_bounds__(256) void k_augment_points(const double* __restrict__ X, int64_t n, int d, int64_t ldx,
                                                        const double* __restrict__ ls, int ls_len,
                                                        const double* __restrict__ center, int side,
                                                        double* __restrict__ out, int da) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int l16 = threadIdx.x & 15;
    const bool live = i < n;
    const int64_t ir = live ? i : n - 1;
    const double sc = (side == 0) ? 369.3299304675746 : 1.0;       // 256/ln2: exp_tab4's pre-scaling
    double v[2], nrm = 0.0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int j = l16 + 16 * h, jc = min(j, d - 1);
        const double x = (X[ir * ldx + jc] - center[jc]) / ls[ls_len == 1 ? 0 : jc];
        v[h] = (j < d) ? x : 0.0;
        nrm = fma(v[h], v[h], nrm);
    }
    nrm += aug_dpp<0x128>(nrm);                       // row_ror 8, 4, 2, 1: every lane gets the row total
    nrm += aug_dpp<0x124>(nrm);
    nrm += aug_dpp<0x122>(nrm);
    nrm += aug_dpp<0x121>(nrm);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int j = l16 + 16 * h;
        double o = v[h] * sc;
        if (j == d + side) o = -0.5 * nrm * sc;
        else if (j == d + 1 - side) o = sc;
        else if (j >= d) o = 0.0;
        if (live && j < da) out[i * da + j] = o;
    }
}

// The plan's augmented tables in TWO launches (round 6; they were a concatenation, a mean, and one k_augment_points per
// table): k_center_mean -- the column means of the Nystrom points, summed in a fixed order: "any shift works; this one keeps
// |x~| small" -- and k_augment_all, one grid over the row table [X_nys; X_obs] (side 0) followed by the pool (side 1), each
// read from where it lies.
__global__ __launch_bounds__(1024) void k_center_mean(const double* __restrict__ X, int64_t n, int d, int64_t ldx,
                                                      double* __restrict__ center) {
    __shared__ double s_p[32][33];
    const int j = threadIdx.x & 31, g = threadIdx.x >> 5;         // coordinate, row group (32 groups: a chain of n / 32 loads each)
    double acc = 0.0;
    if (j < d) {
#pragma unroll 4
        for (int64_t i = g; i < n; i += 32) acc += X[i * ldx + j];
    }
    s_p[g][j] = acc;
    __syncthreads();
    if (g == 0 && j < d) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < 32; ++q) t += s_p[q][j];
        center[j] = t / (double)n;
    }
}

__global__ __launch_bounds__(256) void k_augment_all(const double* __restrict__ Xa, int64_t na, int64_t lda,
                                                     const double* __restrict__ Xb, int64_t nb, int64_t ldb,
                                                     const double* __restrict__ Xc, int64_t nc, int64_t ldc, int d,
                                                     const double* __restrict__ ls, int ls_len,
                                                     const double* __restrict__ center, double* __restrict__ rows_out,
                                                     double* __restrict__ cand_out, int da) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int l16 = threadIdx.x & 15;
    const int64_t n_rows = na + nb, n_all = n_rows + nc;
    const bool live = i < n_all;
    const int64_t ir = live ? i : n_all - 1;
    const int side = ir < n_rows ? 0 : 1;
    const double* src = ir < na ? Xa + ir * lda : (ir < n_rows ? Xb + (ir - na) * ldb : Xc + (ir - n_rows) * ldc);
    double* dst = side == 0 ? rows_out + ir * da : cand_out + (ir - n_rows) * da;
    const double sc = (side == 0) ? 369.3299304675746 : 1.0;       // 256/ln2: exp_tab4's pre-scaling
    double v[2], nrm = 0.0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int j = l16 + 16 * h, jc = min(j, d - 1);
        const double x = (src[jc] - center[jc]) / ls[ls_len == 1 ? 0 : jc];
        v[h] = (j < d) ? x : 0.0;
        nrm = fma(v[h], v[h], nrm);
    }
    nrm += aug_dpp<0x128>(nrm);
    nrm += aug_dpp<0x124>(nrm);
    nrm += aug_dpp<0x122>(nrm);
    nrm += aug_dpp<0x121>(nrm);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int j = l16 + 16 * h;
        double o = v[h] * sc;
        if (j == d + side) o = -0.5 * nrm * sc;
        else if (j == d + 1 - side) o = sc;
        else if (j >= d) o = 0.0;
        if (live && j < da) dst[j] = o;
    }
}

template <int KIND, int KT>
static int launch_lm(const double* rows, int n_rows, const double* cand, const int32_t* idx, int64_t pos0,
                     int64_t count, int S, const double* mu, const double* wmul, double os, int n_chunks,
                     double* partG, int ldg, int col0, double* partTot, int64_t tot_limit, hipStream_t st,
                     const int64_t* dR = nullptr, int S_main = 0, int leftover = 0, int S_x = 0, int n_xchunks = 0,
                     double* partG_x = nullptr, int ldg_x = 0, double* partTot_x = nullptr, int wpt_ov = 0) {
    const int64_t e_first = pos0 / S;
    const int e_total = (int)((pos0 + count + S - 1) / S - e_first);
    if (wpt_ov < 0 || wpt_ov > e_total || (wpt_ov > 0 && (dR != nullptr || leftover))) return SOBER_E_ARG;
    const int wpt = wpt_ov > 0 ? wpt_ov : level_wave_wpt(n_rows, e_total, S);
    if (n_chunks != level_wave_slots(wpt)) return SOBER_E_ARG;         // (sober_level_parts_mfma: the slots per tile)
    if (count + 2 * (int64_t)S > 0x7fffffffLL) return SOBER_E_ARG;     // (positions inside a launch are 32-bit, like the list's entries)
    const int64_t n_waves = level_wave_tiles(n_rows, S) * wpt;
    const int64_t n_wg = (n_waves + SOBER_LW_W - 1) / SOBER_LW_W;
    const int grid_main = (int)(8 * ((n_wg + 7) / 8));
    int grid_x = 0;
    if (leftover == 2) {                                               // + the leftover launch's workgroups (at most S_main - 1 positions)
        const int e_x = (S_main - 1 + S_x - 1) / S_x;
        const int wpt_x = level_wave_wpt(n_rows, e_x, S_x);
        if (dR == nullptr || S_x <= 0 || n_xchunks != level_wave_slots(wpt_x) || !partG_x || ldg_x < S_x) return SOBER_E_ARG;
        const int64_t n_wg_x = (level_wave_tiles(n_rows, S_x) * wpt_x + SOBER_LW_W - 1) / SOBER_LW_W;
        grid_x = (int)(8 * ((n_wg_x + 7) / 8));
    }
    SOBER_LAUNCH_TIMED((k_level_reduce_wave<KIND, KT>), dim3((unsigned)(grid_main + grid_x)), dim3(SOBER_LW_W * 64), 0,
                       st, rows, n_rows, cand, idx, pos0, count, S, mu, wmul, os, e_first, e_total, partG, ldg, col0,
                       partTot, tot_limit, dR, S_main, leftover, S_x, partG_x, ldg_x, partTot_x, grid_main, wpt_ov);
    LAUNCH_CHECK();
    return 0;
}

}  // namespace sober

using namespace sober;

extern "C" int sober_level_parts_mfma(int n_rows, int64_t pos0, int64_t count, int S) {
    if (n_rows <= 0 || pos0 < 0 || count <= 0 || S <= 0) return SOBER_E_ARG;
    return level_parts_mfma_for(n_rows, (pos0 + count + S - 1) / S - pos0 / S, S);
}

extern "C" int sober_level_parts_mfma_cap(int n_rows, int64_t e_total_ub, int S) {
    if (n_rows <= 0 || e_total_ub <= 0 || S <= 0) return SOBER_E_ARG;
    return level_parts_mfma_cap(n_rows, e_total_ub, S);
}

extern "C" int sober_aug_dim(int d) {
    if (d <= 0) return SOBER_E_ARG;
    const int da = ((d + 2 + 3) / 4) * 4;
    return da <= 32 ? da : SOBER_E_DIM;
}

extern "C" int sober_augment_points(const double* X, int64_t n, int d, int64_t ldx, const double* lengthscale,
                                    int ls_len, const double* center, int side, double* out, int da,
                                    void* stream) {
    if (!X || !lengthscale || !center || !out || n <= 0 || d <= 0 || ldx < d || da < d + 2) return SOBER_E_ARG;
    if ((ls_len != 1 && ls_len != d) || (side != 0 && side != 1)) return SOBER_E_ARG;
    if (da > 32) return SOBER_E_DIM;
    hipLaunchKernelGGL(k_augment_points, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, (hipStream_t)stream, X, n,
                       d, ldx, lengthscale, ls_len, center, side, out, da);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int sober_augment_plan(const double* X_nys, int64_t M, int64_t ld_nys, const double* X_obs, int64_t n_obs,
                                  int64_t ld_obs, const double* X_cand, int64_t N, int64_t ld_cand, int d,
                                  const double* lengthscale, int ls_len, double* center, double* rows_aug, double* cand_aug,
                                  int da, void* stream) {
    if (!X_nys || !X_cand || !lengthscale || !center || !rows_aug || !cand_aug || M <= 0 || N <= 0 || n_obs < 0 || d <= 0 ||
        (n_obs > 0 && !X_obs) || ld_nys < d || ld_cand < d || (n_obs > 0 && ld_obs < d) || da < d + 2)
        return SOBER_E_ARG;
    if ((ls_len != 1 && ls_len != d) || d > 30) return SOBER_E_ARG;
    if (da > 32) return SOBER_E_DIM;
    hipLaunchKernelGGL(k_center_mean, dim3(1), dim3(1024), 0, (hipStream_t)stream, X_nys, M, d, ld_nys, center);
    LAUNCH_CHECK();
    const int64_t n_all = M + n_obs + N;
    hipLaunchKernelGGL(k_augment_all, dim3((unsigned)((n_all + 15) / 16)), dim3(256), 0, (hipStream_t)stream, X_nys, M, ld_nys,
                       X_obs, n_obs, ld_obs, X_cand, N, ld_cand, d, lengthscale, ls_len, center, rows_aug, cand_aug, da);
    LAUNCH_CHECK();
    return 0;
}

static int level_reduce_mfma_impl(int kind, const double* rows, int n_rows, const double* cand, int da,
                                  const int32_t* idx, int64_t pos0, int64_t count, int S, const double* mu,
                                  const double* wmul, double outputscale, int n_chunks, double* partG,
                                  int ldg, int col0, double* partTot, int64_t tot_limit, void* stream,
                                  const int64_t* dR, int S_main, int leftover, int S_x = 0, int n_xchunks = 0,
                                  double* partG_x = nullptr, int ldg_x = 0, double* partTot_x = nullptr, int wpt_ov = 0);

extern "C" int sober_level_reduce_mfma(int kind, const double* rows, int n_rows, const double* cand, int da,
                                       const int32_t* idx, int64_t pos0, int64_t count, int S, const double* mu,
                                       const double* wmul, double outputscale, int n_chunks, double* partG,
                                       int ldg, int col0, double* partTot, int64_t tot_limit, void* stream) {
    return level_reduce_mfma_impl(kind, rows, n_rows, cand, da, idx, pos0, count, S, mu, wmul, outputscale, n_chunks,
                                  partG, ldg, col0, partTot, tot_limit, stream, nullptr, 0, 0);
}

extern "C" int sober_level_reduce_mfma_queued(int kind, const double* rows, int n_rows, const double* cand, int da,
                                              const int32_t* idx, int64_t count_ub, int S, int S_main,
                                              int leftover, const double* mu, const double* wmul,
                                              double outputscale, int n_chunks_ub, double* partG, int ldg,
                                              double* partTot, const int64_t* dR, void* stream) {
    if (!dR || S_main <= 0 || (!leftover && S != S_main)) return SOBER_E_ARG;
    return level_reduce_mfma_impl(kind, rows, n_rows, cand, da, idx, 0, count_ub, S, mu, wmul, outputscale,
                                  n_chunks_ub, partG, ldg, 0, partTot, 0, stream, dR, S_main, leftover);
}

// The class launch (round 6, level_exec.cpp: lx_class_first): the same kernel over S = 2^D x (the level's sets) with an
// explicit number of waves per tile -- sober_level_class_wpt -- whose line may be several rounds of the chip long.
// n_chunks = sober_level_class_slots(wpt).  Positions [0, count) of the list, no leftovers (count % S == 0).
extern "C" int sober_level_class_wpt(int n_rows, int64_t e_total, int S) {
    if (n_rows <= 0 || e_total <= 0 || S <= 0) return SOBER_E_ARG;
    return level_class_wpt(n_rows, e_total, S);
}
extern "C" int sober_level_class_slots(int wpt) { return wpt > 0 ? level_wave_slots(wpt) : SOBER_E_ARG; }
extern "C" int sober_level_reduce_mfma_wpt(int kind, const double* rows, int n_rows, const double* cand, int da,
                                           const int32_t* idx, int64_t count, int S, const double* mu,
                                           const double* wmul, double outputscale, int wpt, int n_chunks, double* partG,
                                           int ldg, double* partTot, void* stream) {
    if (wpt <= 0 || count <= 0 || S <= 0 || count % S != 0) return SOBER_E_ARG;
    return level_reduce_mfma_impl(kind, rows, n_rows, cand, da, idx, 0, count, S, mu, wmul, outputscale, n_chunks, partG,
                                  ldg, 0, partTot, count, stream, nullptr, 0, 0, 0, 0, nullptr, 0, nullptr, wpt);
}

// both placements of a queued level in ONE launch: the leftover launch's workgroups ride behind the main ones
extern "C" int sober_level_reduce_mfma_queued_pair(int kind, const double* rows, int n_rows, const double* cand, int da,
                                                   const int32_t* idx, int64_t count_ub, int S, int n_xcols,
                                                   const double* mu, const double* wmul, double outputscale,
                                                   int n_chunks_ub, double* partG, int ldg, double* partTot,
                                                   int n_xchunks_ub, double* extraG, double* extraTot,
                                                   const int64_t* dR, void* stream) {
    if (!dR || S <= 1 || n_xcols <= 0 || !extraG || !extraTot) return SOBER_E_ARG;
    return level_reduce_mfma_impl(kind, rows, n_rows, cand, da, idx, 0, count_ub, S, mu, wmul, outputscale,
                                  n_chunks_ub, partG, ldg, 0, partTot, 0, stream, dR, S, 2, n_xcols, n_xchunks_ub, extraG,
                                  n_xcols, extraTot);
}

static int level_reduce_mfma_impl(int kind, const double* rows, int n_rows, const double* cand, int da,
                                  const int32_t* idx, int64_t pos0, int64_t count, int S, const double* mu,
                                  const double* wmul, double outputscale, int n_chunks, double* partG,
                                  int ldg, int col0, double* partTot, int64_t tot_limit, void* stream,
                                  const int64_t* dR, int S_main, int leftover, int S_x, int n_xchunks,
                                  double* partG_x, int ldg_x, double* partTot_x, int wpt_ov) {
    if (!rows || !cand || !idx || !mu || !partG) return SOBER_E_ARG;
    if (n_rows <= 0 || pos0 < 0 || count <= 0 || S <= 0 || n_chunks <= 0 || ldg < col0 + S) return SOBER_E_ARG;
    hipStream_t st = (hipStream_t)stream;
#define LM_CASE(K, T)                                                                                          \
    case 4 * T:                                                                                                \
        return launch_lm<K, T>(rows, n_rows, cand, idx, pos0, count, S, mu, wmul, outputscale, n_chunks, partG, \
                               ldg, col0, partTot, tot_limit, st, dR, S_main, leftover, S_x, n_xchunks, partG_x,   \
                               ldg_x, partTot_x, wpt_ov);
    switch (kind) {
        case SOBER_KIND_RBF:
            switch (da) { LM_CASE(SOBER_KIND_RBF, 1) LM_CASE(SOBER_KIND_RBF, 2) LM_CASE(SOBER_KIND_RBF, 3)
                          LM_CASE(SOBER_KIND_RBF, 4) LM_CASE(SOBER_KIND_RBF, 5) LM_CASE(SOBER_KIND_RBF, 6)
                          LM_CASE(SOBER_KIND_RBF, 7) LM_CASE(SOBER_KIND_RBF, 8) default: return SOBER_E_DIM; }
        case SOBER_KIND_MATERN52:
            switch (da) { LM_CASE(SOBER_KIND_MATERN52, 1) LM_CASE(SOBER_KIND_MATERN52, 2)
                          LM_CASE(SOBER_KIND_MATERN52, 3) LM_CASE(SOBER_KIND_MATERN52, 4)
                          LM_CASE(SOBER_KIND_MATERN52, 5) LM_CASE(SOBER_KIND_MATERN52, 6)
                          LM_CASE(SOBER_KIND_MATERN52, 7) LM_CASE(SOBER_KIND_MATERN52, 8)
                          default: return SOBER_E_DIM; }
        default: return SOBER_E_ARG;
    }
#undef LM_CASE
}
