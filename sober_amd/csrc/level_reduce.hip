// K1-K3: fused "kernel evaluation x weight -> set sums" (SOBER/_rchq.py:116-136,152).
//
// Mapping (CDNA4, wave64): lanes <-> rows of the stacked table [X_nys; X_obs] (each lane keeps its
// row's DT scaled coordinates in VGPRs for the whole kernel), the workgroup walks a block of SB
// consecutive sets over its element chunk; the SB candidates of one element are consecutive list
// positions, so their gather through idx_story is (at level 0 exactly, later nearly) a contiguous
// read.  Candidates are staged through LDS once per workgroup and read back as wave-uniform
// (broadcast) operands; every lane owns SB FP64 accumulators, so there are no atomics and the
// summation order is fixed by (n_chunks, element order) alone -> bit-reproducible.
#include "common.hpp"

namespace sober {

constexpr int LR_RW = 4;    // row waves per workgroup (256 rows)
constexpr int LR_SB = 16;   // sets per workgroup

template <int DT>
struct LrTile {
    static constexpr int TE = (DT <= 16) ? 8 : 4;   // elements staged per tile
    static constexpr int NT = TE * LR_SB;           // candidates per tile
};

template <int KIND, int DT>
__global__ __launch_bounds__(LR_RW * 64) void k_level_reduce(
    const double* __restrict__ rows, const double* __restrict__ rows_norm, int n_rows,
    const double* __restrict__ cand, const double* __restrict__ cand_norm,
    const int32_t* __restrict__ idx, int64_t pos0, int64_t count, int S,
    const double* __restrict__ mu, const double* __restrict__ wmul, double os,
    int64_t e_first, int e_total, int e_per_chunk,
    double* __restrict__ partG, int ldg, int col0,
    double* __restrict__ partTot, int64_t tot_limit) {
    constexpr int SB = LR_SB;
    constexpr int TE = LrTile<DT>::TE;
    constexpr int NT = LrTile<DT>::NT;
    __shared__ double s_pts[2][NT][DT];
    __shared__ double s_w[2][NT];
    __shared__ double s_ny[2][NT];
    __shared__ double s_tot[NT];

    const int tid = threadIdx.x;
    const int s0 = blockIdx.x * SB;
    const int chunk = blockIdx.y;
    const int row = blockIdx.z * (LR_RW * 64) + tid;
    const int e0 = chunk * e_per_chunk;
    const int e1 = min(e0 + e_per_chunk, e_total);

    double x[DT];
    double nx = 0.0;
    if (row < n_rows) {
#pragma unroll
        for (int j = 0; j < DT; ++j) x[j] = rows[(size_t)row * DT + j];
        if constexpr (KIND == SOBER_KIND_TANIMOTO) nx = rows_norm[row];
    } else {
#pragma unroll
        for (int j = 0; j < DT; ++j) x[j] = 0.0;
    }

    double acc[SB];
#pragma unroll
    for (int i = 0; i < SB; ++i) acc[i] = 0.0;

    const bool stager = tid < NT;
    const int st_te = tid / SB, st_i = tid % SB;
    double st[DT];
    double st_w = 0.0, st_ny = 0.0, tot_acc = 0.0;

    // ---- stage_load: global -> registers for the tile starting at element e_tile ----
#define LR_STAGE_LOAD(e_tile)                                                              \
    if (stager) {                                                                          \
        const int e_ = (e_tile) + st_te;                                                   \
        const int s_ = s0 + st_i;                                                          \
        const int64_t p_ = (e_first + e_) * S + s_;          /* global list position */    \
        const bool ok_ = (s_ < S) && (e_ < e1) && (p_ >= pos0) && (p_ < pos0 + count);     \
        st_w = 0.0; st_ny = 0.0;                                                           \
        if (ok_) {                                                                         \
            const int c_ = idx[p_ - pos0];                                                 \
            const double m_ = mu[c_];                                                      \
            st_w = wmul ? m_ * wmul[c_] : m_;                                              \
            if (p_ < tot_limit) tot_acc += m_;                                             \
            const double* src_ = cand + (size_t)c_ * DT;                                   \
            _Pragma("unroll") for (int j = 0; j < DT; ++j) st[j] = src_[j];                \
            if constexpr (KIND == SOBER_KIND_TANIMOTO) st_ny = cand_norm[c_];              \
        } else {                                                                           \
            _Pragma("unroll") for (int j = 0; j < DT; ++j) st[j] = 0.0;                    \
        }                                                                                  \
    }
#define LR_STAGE_WRITE(buf)                                                                \
    if (stager) {                                                                          \
        _Pragma("unroll") for (int j = 0; j < DT; ++j) s_pts[buf][tid][j] = st[j];         \
        s_w[buf][tid] = st_w;                                                              \
        s_ny[buf][tid] = st_ny;                                                            \
    }

    int buf = 0;
    if (e0 < e1) {
        LR_STAGE_LOAD(e0);
        LR_STAGE_WRITE(0);
    }
    __syncthreads();

    for (int et = e0; et < e1; et += TE) {
        const bool more = (et + TE) < e1;
        if (more) { LR_STAGE_LOAD(et + TE); }
        const int te_cnt = min(TE, e1 - et);
        for (int te = 0; te < te_cnt; ++te) {
#pragma unroll
            for (int i = 0; i < SB; ++i) {
                const int q = te * SB + i;
                double y[DT];
#pragma unroll
                for (int j = 0; j < DT; ++j) y[j] = s_pts[buf][q][j];
                const double k = kern_eval<KIND, DT>(x, nx, y, s_ny[buf][q], os);
                acc[i] = fma(k, s_w[buf][q], acc[i]);
            }
        }
        if (more) { LR_STAGE_WRITE(buf ^ 1); }
        __syncthreads();
        buf ^= 1;
    }
#undef LR_STAGE_LOAD
#undef LR_STAGE_WRITE

    if (row < n_rows) {
        double* o = partG + ((size_t)chunk * n_rows + row) * ldg + col0 + s0;
#pragma unroll
        for (int i = 0; i < SB; ++i)
            if (s0 + i < S) o[i] = acc[i];
    }
    if (partTot != nullptr && blockIdx.z == 0) {
        if (stager) s_tot[tid] = tot_acc;
        __syncthreads();
        if (tid < SB && s0 + tid < S) {
            double t = 0.0;
#pragma unroll
            for (int te = 0; te < TE; ++te) t += s_tot[te * SB + tid];
            partTot[(size_t)chunk * ldg + col0 + s0 + tid] = t;
        }
    }
}

template <int KIND, int DT>
static int launch_lr(const void* rows, const double* rows_norm, int n_rows, const void* cand,
                     const double* cand_norm, const int32_t* idx, int64_t pos0, int64_t count, int S,
                     const double* mu, const double* wmul, double os, int n_chunks, double* partG,
                     int ldg, int col0, double* partTot, int64_t tot_limit, hipStream_t st) {
    const int64_t e_first = pos0 / S;
    const int e_total = (int)((pos0 + count + S - 1) / S - e_first);
    const int e_per_chunk = (e_total + n_chunks - 1) / n_chunks;
    dim3 grid((S + LR_SB - 1) / LR_SB, n_chunks, (n_rows + LR_RW * 64 - 1) / (LR_RW * 64));
    hipLaunchKernelGGL((k_level_reduce<KIND, DT>), grid, dim3(LR_RW * 64), 0, st,
                       (const double*)rows, rows_norm, n_rows, (const double*)cand, cand_norm, idx,
                       pos0, count, S, mu, wmul, os, e_first, e_total, e_per_chunk, partG, ldg, col0,
                       partTot, tot_limit);
    LAUNCH_CHECK();
    return 0;
}

}  // namespace sober

using namespace sober;

// the largest value sober_level_chunks can take up to e_total_ub elements per set (what a launch sized ahead of time is
// given: sober_level_reduce_tani_queued)
extern "C" int sober_level_chunks_cap(int n_rows, int64_t e_total_ub, int S) {
    if (n_rows <= 0 || e_total_ub <= 0 || S <= 0) return SOBER_E_ARG;
    int best = 0;
    // (the count is not monotone in the number of elements: n = min(target, 64, e) chunks of ceil(e / n) elements)
    const int sb = (S + 15) / 16, rb = (n_rows + 255) / 256;
    int64_t n = SOBER_CHUNK_TARGET / ((int64_t)sb * rb);
    if (n < 1) n = 1;
    if (n > 64) n = 64;
    if (n > e_total_ub) n = e_total_ub;
    best = (int)n;
    return best;
}

extern "C" int sober_level_chunks(int n_rows, int64_t pos0, int64_t count, int S) {
    if (n_rows <= 0 || pos0 < 0 || count <= 0 || S <= 0) return SOBER_E_ARG;
    const int64_t e_total = (pos0 + count + S - 1) / S - pos0 / S;
    static_assert(LR_SB == 16 && LR_RW == 4, "level_chunks_for assumes 16 sets x 256 rows per workgroup");
    return sober::level_chunks_for(n_rows, e_total, S);
}

extern "C" int sober_level_reduce(int kind, const void* rows, const double* rows_norm, int n_rows,
                                  const void* cand, const double* cand_norm, int dt,
                                  const int32_t* idx, int64_t pos0, int64_t count, int S, const double* mu,
                                  const double* wmul, double outputscale, int n_chunks,
                                  double* partG, int ldg, int col0, double* partTot,
                                  int64_t tot_limit, void* stream) {
    if (!rows || !cand || !idx || !mu || !partG) return SOBER_E_ARG;
    if (n_rows <= 0 || pos0 < 0 || count <= 0 || S <= 0 || n_chunks <= 0 || ldg < col0 + S)
        return SOBER_E_ARG;
    if (n_chunks > (pos0 + count + S - 1) / S - pos0 / S) return SOBER_E_ARG;
    if (kind == SOBER_KIND_TANIMOTO && (!rows_norm || !cand_norm)) return SOBER_E_ARG;
    hipStream_t st = (hipStream_t)stream;
#define LR_CASE(K, D)                                                                            \
    case D:                                                                                      \
        return launch_lr<K, D>(rows, rows_norm, n_rows, cand, cand_norm, idx, pos0, count, S, mu, wmul, \
                               outputscale, n_chunks, partG, ldg, col0, partTot, tot_limit, st);
    switch (kind) {
        case SOBER_KIND_RBF:
            switch (dt) { LR_CASE(SOBER_KIND_RBF, 4) LR_CASE(SOBER_KIND_RBF, 8)
                          LR_CASE(SOBER_KIND_RBF, 12) LR_CASE(SOBER_KIND_RBF, 16)
                          LR_CASE(SOBER_KIND_RBF, 20) LR_CASE(SOBER_KIND_RBF, 24)
                          LR_CASE(SOBER_KIND_RBF, 32) default: return SOBER_E_DIM; }
        case SOBER_KIND_MATERN52:
            switch (dt) { LR_CASE(SOBER_KIND_MATERN52, 4) LR_CASE(SOBER_KIND_MATERN52, 8)
                          LR_CASE(SOBER_KIND_MATERN52, 12) LR_CASE(SOBER_KIND_MATERN52, 16)
                          LR_CASE(SOBER_KIND_MATERN52, 20) LR_CASE(SOBER_KIND_MATERN52, 24)
                          LR_CASE(SOBER_KIND_MATERN52, 32) default: return SOBER_E_DIM; }
        case SOBER_KIND_TANIMOTO:
            switch (dt) { LR_CASE(SOBER_KIND_TANIMOTO, 1) LR_CASE(SOBER_KIND_TANIMOTO, 2)
                          LR_CASE(SOBER_KIND_TANIMOTO, 4) LR_CASE(SOBER_KIND_TANIMOTO, 8)
                          LR_CASE(SOBER_KIND_TANIMOTO, 16) LR_CASE(SOBER_KIND_TANIMOTO, 32)
                          default: return SOBER_E_DIM; }
        default: return SOBER_E_ARG;
    }
#undef LR_CASE
}
