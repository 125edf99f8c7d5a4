// Shared device helpers for libsober_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/sober_hip.h"

// Diagnostic builds: the in-kernel time stamps (*_STAMPS switches: same results, slower kernels, debug exports) exist
// only behind SOBER_DIAG_BUILD, which `make all` never defines.  Such a library says so (sober_diag_build() = 1) and
// sober_amd._native.load() refuses it unless SOBER_ALLOW_DIAG_LIB=1 (the stamp scripts under scripts/ set it).  There
// are no switches that change RESULTS: the timing-only experiments of rounds 2-3 are out of the source (their numbers:
// profiles/r03_bidiag_where.txt).
#if (defined(CAR_BSTAMPS) || defined(SP_TSTAMPS) || defined(MC_STAMPS) || defined(MC_TSTAMPS) || defined(CM_STAMPS) || \
     defined(CH_STAMPS) || defined(LW_STAMPS) || defined(CG_STAMPS)) && !defined(SOBER_DIAG_BUILD)
#error "in-kernel stamps are diagnostic builds: make stamps, or EXTRA='-DSOBER_DIAG_BUILD -D..._STAMPS' with a BUILD/OUT of its own"
#endif

#define SOBER_MAX_DT 32          // largest padded dimension with a register-tiled instantiation
#define SOBER_WAVE 64

#define HIP_TRY(expr)                                   \
    do {                                                \
        hipError_t _e = (expr);                         \
        if (_e != hipSuccess) return (int)_e;           \
    } while (0)

// Function attributes (dynamic LDS beyond 64 KB) are per DEVICE: one bit per device ordinal, set after the attribute
// calls for that device went through.  Usage:  static std::atomic<unsigned long long> done{0};
//   if (sober_attr_needed(done)) { HIP_TRY(hipFuncSetAttribute(...)); sober_attr_done(done); }
#include <atomic>
static inline bool sober_attr_needed(std::atomic<unsigned long long>& done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return true;
    return (done.load(std::memory_order_acquire) & (1ull << (dev & 63))) == 0;
}
static inline void sober_attr_done(std::atomic<unsigned long long>& done) {
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) done.fetch_or(1ull << (dev & 63), std::memory_order_release);
}

// Kernel launches that can carry a pair of events IN THEIR DISPATCH (hipExtLaunchKernelGGL): start / stop then hold the
// kernel's own begin / end timestamps -- the duration rocprofv3 reports for it -- with no marker packets on the stream.
// sober_set_launch_events arms the pair for the NEXT such launch of the calling thread (the level kernels).
#include <hip/hip_ext.h>
namespace sober {
struct LaunchEvents { hipEvent_t start = nullptr, stop = nullptr; };
LaunchEvents& launch_events();                      // (thread-local, misc.hip)
}
#define SOBER_LAUNCH_TIMED(kernel, grid, block, shmem, stream, ...)                                        \
    do {                                                                                                    \
        sober::LaunchEvents& le_ = sober::launch_events();                                                  \
        if (le_.start != nullptr && le_.stop != nullptr) {                                                  \
            hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, le_.start, le_.stop, 0, __VA_ARGS__); \
            le_ = sober::LaunchEvents{};                                                                    \
        } else {                                                                                            \
            hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                            \
        }                                                                                                   \
    } while (0)

#include "switches.hpp"

#define LAUNCH_CHECK()                                  \
    do {                                                \
        hipError_t _e = hipGetLastError();              \
        if (_e != hipSuccess) return (int)_e;           \
    } while (0)

#ifndef SOBER_CHUNK_TARGET
#define SOBER_CHUNK_TARGET 512      // workgroups a level launch aims for (~2 per CU): 1024 doubles the partial-sum traffic and is 9 % slower, 256 starves the 1M-row level (measured, scripts/chunk_target_ab.py)
#endif

namespace sober {

// Element chunks of one level launch (sober_level_chunks): ~4 workgroups per CU on 256 CUs, at most 64, no empty
// chunk.  Host and device use the same formula: a launch sized from an upper bound of the live positions finds
// its own chunk count from the exact number it reads on the device (level_exec.cpp: the queued loop).
__host__ __device__ inline int level_chunks_for(int n_rows, int64_t e_total, int S) {
    const int sb = (S + 15) / 16;
    const int rb = (n_rows + 255) / 256;
    int64_t n = SOBER_CHUNK_TARGET / ((int64_t)sb * rb);
    if (n < 1) n = 1;
    if (n > 64) n = 64;
    if (n > e_total) n = e_total;
    if (n < 1) return 0;
    const int64_t epc = (e_total + n - 1) / n;
    return (int)((e_total + epc - 1) / epc);
}

// Element chunks of a launch of the FINGERPRINT level kernel (level_reduce_tani.hip): one workgroup per compute unit (512
// threads, ~240 VGPRs), all of a launch's workgroups equally long -- a launch costs ROUNDS x one workgroup's time, and a
// workgroup's time is a fixed part (the rows' bits expanded into registers, two dependent gathers to the first tile, 32 KB
// of partial sums out: ~3 tile times) plus its tiles of two elements.  So the count aims at ONE round of the 256 compute
// units, INCLUDING the set group of the leftover positions that rides in the same grid on the queued levels: a second
// round never pays (2 x (fixed + tiles / 2) > fixed + tiles).  And the chunk index is the SLOWEST grid dimension: a
// queued launch is sized from an upper bound, its surplus workgroups leave at once -- but to be dispatched each of them
// needs a whole compute unit's registers and LDS like a working one, in order: placed between working ones (chunk as the
// middle dimension, a launch sized for 12 chunks of which 6 exist) they wait for a unit to come free and everything behind
// them waits too -- 275 us against 156 (scripts/tani_queued_time.py; an empty grid of 2048 such workgroups costs nothing
// by itself).  Round 3 sized for 512 workgroups without the leftover group: 13 chunks x 42 workgroups = 546 = THREE
// rounds on the queued levels (levels 1-9 of cfg-5 at 2/3 of their speed, profiles/r04_pmc_cfg5.csv) and 13 partial-sum
// slots of the whole table whatever the level's size.  Host and device use the same formula; scripts/tani_chunk_sweep.py
// has the sweep (6 chunks is the fastest at every level size of cfg-5).
__host__ __device__ inline int level_chunks_tani_cap(int n_rows, int64_t e_total_ub, int S) {
    if (e_total_ub < 1) return 0;
    const int64_t per = (int64_t)((S + 15) / 16 + 1) * ((n_rows + 255) / 256);      // workgroups per chunk
    int64_t n = 256 / per;
    if (n < 1) n = 1;
    if (n > 64) n = 64;
    if (n > e_total_ub) n = e_total_ub;
    // (the leftover positions' launch -- 16 pseudo-sets, <= 13 elements: k_sum_partials folds its chunks x 16 columns into
    //  set S - 1 in ONE thread per row, 208 dependent loads = 19 us at 13 chunks; two chunks: 32)
    if (S <= 16 && n > (e_total_ub + 5) / 6) n = (e_total_ub + 5) / 6;
    return (int)n;
}
__host__ __device__ inline int level_chunks_tani_for(int n_rows, int64_t e_total, int S) {
    if (e_total < 1) return 0;
    const int64_t n = level_chunks_tani_cap(n_rows, e_total, S);
    const int64_t epc = (e_total + n - 1) / n;
    return (int)((e_total + epc - 1) / epc);
}

// Split of one level launch of the wave-autonomous matrix-core kernel (level_reduce_mfma.hip: k_level_reduce_wave).
// A wave owns a 64-row x 16-set tile over a contiguous range of elements.  All waves of a launch form ONE line, tile
// after tile (row tiles of a set group adjacent), `wpt` waves per tile; a workgroup is SOBER_LW_W consecutive waves
// of that line, so it may straddle two tiles: the waves of one tile inside a workgroup are summed through LDS in wave
// order, and a tile leaves one partial sum per workgroup it touches -- at most level_wave_slots() of them.
// At most SOBER_WAVE_TARGET waves per launch: two per SIMD on 1024 SIMDs, all resident at once (the kernel is
// instruction-issue bound; a second round of workgroups would double its time).
#ifndef SOBER_WAVE_TARGET
#define SOBER_WAVE_TARGET 2048
#endif
#ifndef SOBER_LW_W
#define SOBER_LW_W 4
#endif
__host__ __device__ inline int64_t level_wave_tiles(int n_rows, int S) {
    return (int64_t)((S + 15) / 16) * ((n_rows + 63) / 64);
}
// waves per tile: monotone in e_total (a launch sized from an upper bound of e_total covers the exact one)
__host__ __device__ inline int level_wave_wpt(int n_rows, int64_t e_total, int S) {
    int64_t wpt = SOBER_WAVE_TARGET / level_wave_tiles(n_rows, S);
    if (wpt < 1) wpt = 1;
    if (wpt > e_total) wpt = e_total;
    return (int)wpt;
}
// partial-sum slots per tile = the most workgroups `wpt` consecutive waves can touch
__host__ __device__ inline int level_wave_slots(int wpt) { return wpt <= 0 ? 0 : (wpt + SOBER_LW_W - 2) / SOBER_LW_W + 1; }

// Waves per tile of the CLASS launch (round 6: level 0 over 2^D x S sets, whose element-class sums give the next D levels
// without a kernel evaluation -- level_exec.cpp).  Its tiles are 2^D times the level's: a one-round line (level_wave_wpt)
// would leave the chip half empty (1100 tiles at 8 classes of cfg-4: one wave each) or spill into a short second round.
// The line may be several rounds long instead; wpt minimises  rounds x (elements per wave x t_e + t_0)  with the per-wave
// figures of profiles/r03_level_stamps.txt (1.7 us per element and wave at two waves per SIMD, 6 us fixed).  Host-known
// sizes only (no queued launch uses it), so it need not be monotone.
__host__ __device__ inline int level_class_wpt(int n_rows, int64_t e_total, int S) {
    const int64_t tiles = level_wave_tiles(n_rows, S);
    int best = 1;
    double best_t = 1e300;
    for (int w = 1; w <= 16 && w <= e_total; ++w) {
        const int64_t rounds = (tiles * w + SOBER_WAVE_TARGET - 1) / SOBER_WAVE_TARGET;
        const double t = (double)rounds * ((double)((e_total + w - 1) / w) * 1.7 + 6.0);
        if (t < best_t * 0.97) { best_t = t; best = w; }            // (a longer line has to pay for its partial sums)
    }
    return best;
}

// partial sums per tile of the matrix-core level kernel (monotone in e_total: a launch sized from an upper bound covers
// the exact one)
__host__ __device__ inline int level_parts_mfma_for(int n_rows, int64_t e_total, int S) {
    return level_wave_slots(level_wave_wpt(n_rows, e_total, S));
}
__host__ __device__ inline int level_parts_mfma_cap(int n_rows, int64_t e_total, int S) {
    return level_wave_slots(level_wave_wpt(n_rows, e_total, S));
}

#ifdef __HIPCC__
// dlarfg: reflector H = I - tau [1; v][1; v]^T with H [alpha; x] = [beta; 0], ss = |x|^2, v = x * scal -- computed
// from 1 / norm alone (beta itself is never needed here):
//   tau = (beta - alpha) / beta = 1 + |alpha| / norm,   scal = 1 / (alpha - beta) = sign(alpha) / (norm tau)
// Branch-free (a taken branch costs a single resident wave 50-80 cycles of instruction fetch): operands far outside the
// normal range are rescaled by 2^(+-300) with selects, ss = 0 gives the identity (tau = 0).
// -- one refined rsq, one fma, one refined rcp of a number in [1, 2]: 15 dependent operations instead of 23
// (Round 5: the rescaling kept off the common path -- the rsq started on the unscaled norm^2 in front of a wave-uniform
//  test, the scaled sequence only for tiny / huge / NaN operands; bit-identical by construction and on 103 steps -- measured
//  184.7 against 183.1 us for the 200 x 100 bidiagonalisation: the five dependent operations it removes are not what the
//  step waits for, the branch costs more.  Dropped; scripts/car_ab3.sh has the A/B.)
__device__ __forceinline__ void larfg_vt(double alpha, double ss, double& tau, double& scal) {
    const double n2r = fma(alpha, alpha, ss);
    const bool tiny = n2r < 1e-200, huge = n2r > 1e200;
    const double f = tiny ? 0x1p300 : (huge ? 0x1p-300 : 1.0);
    const double al = alpha * f;
    const double n2 = fma(al, al, (ss * f) * f);
    double r = __builtin_amdgcn_rsq(n2);
    double h = 0.5 * r;
    double e = fma(-(n2 * r), h, 0.5);
    r = fma(r, e, r);
    h = 0.5 * r;
    e = fma(-(n2 * r), h, 0.5);
    r = fma(r, e, r);
    const double t = fma(fabs(al), r, 1.0);
    double y = __builtin_amdgcn_rcp(t);
    e = fma(-t, y, 1.0);
    y = fma(y, e, y);
    e = fma(-t, y, 1.0);
    y = fma(y, e, y);
    const bool none = ss == 0.0;
    tau = none ? 0.0 : t;
    scal = none ? 0.0 : copysign((r * f) * y, al);
}

#endif

// k(x, y) from the squared scaled distance (continuous kernels).
// RBF:      gpytorch RBFKernel  -> exp(-sq / 2)                       [SURVEY App. D]
// Matern52: gpytorch MaternKernel(nu=2.5): r = sqrt(max(sq, 1e-30)),
//           (sqrt5 r + 1 + 5/3 r^2) * exp(-sqrt5 r)                   [SURVEY App. D]
// ScaleKernel multiplies by outputscale afterwards.
template <int KIND>
__device__ __forceinline__ double kern_from_sq(double sq, double os) {
    if constexpr (KIND == SOBER_KIND_RBF) {
        return exp(-0.5 * sq) * os;
    } else {
        const double s5 = 2.23606797749978969641;
        double r = sqrt(fmax(sq, 1e-30));
        double c = (s5 * r + 1.0) + (5.0 / 3.0) * (r * r);
        return (c * exp(-s5 * r)) * os;
    }
}

// Tanimoto similarity of bit vectors, SOBER/_drug_modelling.py:23-25 + clamp_min_(0) of :37.
__device__ __forceinline__ double kern_tanimoto(double dot, double nx, double ny, double os) {
    const double eps = 1e-6;
    double res = (dot + eps) / (((eps + nx) + ny) - dot);
    return fmax(res, 0.0) * os;
}

template <int KIND, int DT>
__device__ __forceinline__ double kern_eval(const double* x, double nx, const double* y, double ny,
                                            double os) {
    if constexpr (KIND == SOBER_KIND_TANIMOTO) {
        int dot = 0;
#pragma unroll
        for (int j = 0; j < DT; ++j)
            dot += __popcll(__double_as_longlong(x[j]) & __double_as_longlong(y[j]));
        return kern_tanimoto((double)dot, nx, ny, os);
    } else {
        double sq = 0.0;
#pragma unroll
        for (int j = 0; j < DT; ++j) {
            double df = x[j] - y[j];
            sq = fma(df, df, sq);
        }
        return kern_from_sq<KIND>(sq, os);
    }
}

// generic-dimension variant (runtime dt, operands in memory)
template <int KIND>
__device__ __forceinline__ double kern_eval_rt(const double* x, double nx, const double* y,
                                               double ny, int dt, double os) {
    if constexpr (KIND == SOBER_KIND_TANIMOTO) {
        int dot = 0;
        for (int j = 0; j < dt; ++j)
            dot += __popcll(__double_as_longlong(x[j]) & __double_as_longlong(y[j]));
        return kern_tanimoto((double)dot, nx, ny, os);
    } else {
        double sq = 0.0;
        for (int j = 0; j < dt; ++j) {
            double df = x[j] - y[j];
            sq = fma(df, df, sq);
        }
        return kern_from_sq<KIND>(sq, os);
    }
}

}  // namespace sober
