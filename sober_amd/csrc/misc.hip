// Small kernels around the hot path: point preparation, materialised kernel matrices, posterior
// mean over a pool, partial-sum reduction, barycentres, the K7 weight/list update, weight scrubbing.
#include "common.hpp"
#include "internal.hpp"

namespace sober {

// ------------------------------------------------------------------ point preparation
__global__ void k_scale_points(const double* __restrict__ X, int64_t n, int d, int64_t ldx,
                               const double* __restrict__ ls, int ls_len, double* __restrict__ out,
                               int dt) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * dt) return;
    const int64_t i = t / dt;
    const int j = (int)(t % dt);
    double v = 0.0;
    if (j < d) v = X[i * ldx + j] / ls[ls_len == 1 ? 0 : j];
    out[t] = v;
}

// ... of two point sets into one table (rows = [Xa; Xb] / lengthscale: the Nystrom points and the observations of a plan)
__global__ void k_scale_points2(const double* __restrict__ Xa, int64_t na, int64_t lda, const double* __restrict__ Xb, int64_t nb,
                                int64_t ldb, int d, const double* __restrict__ ls, int ls_len, double* __restrict__ out, int dt) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (na + nb) * dt) return;
    const int64_t i = t / dt;
    const int j = (int)(t % dt);
    double v = 0.0;
    if (j < d) v = (i < na ? Xa[i * lda + j] : Xb[(i - na) * ldb + j]) / ls[ls_len == 1 ? 0 : j];
    out[t] = v;
}

// ... of the rows idx[0:n] only (the final direct level's <= 2 b points: the pool itself is never copied in scaled form
// on the matrix-core path -- 192 MB written and read back at 1M x 20 for the sake of 200 rows)
// (gsrc != NULL: gout[i] = gsrc[idx[i]] rides along -- the live positions' weights of SOBER/_rchq.py:84, one launch less)
__global__ void k_scale_points_idx(const double* __restrict__ X, const int32_t* __restrict__ idx, int64_t n, int d, int64_t ldx,
                                   const double* __restrict__ ls, int ls_len, double* __restrict__ out, int dt,
                                   const double* __restrict__ gsrc, double* __restrict__ gout) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * dt) return;
    const int64_t i = t / dt;
    const int j = (int)(t % dt);
    const int64_t c = idx[i];
    double v = 0.0;
    if (j < d) v = X[c * ldx + j] / ls[ls_len == 1 ? 0 : j];
    out[t] = v;
    if (gsrc != nullptr && j == 0) gout[i] = gsrc[c];
}

// one wave per row: ballot packs 64 bits at a time
__global__ void k_pack_bits(const double* __restrict__ X, int64_t n, int d, int64_t ldx,
                            uint64_t* __restrict__ words, int nwords, double* __restrict__ norms,
                            int32_t* __restrict__ bad) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n) return;
    int pop = 0;
    for (int w = 0; w < nwords; ++w) {
        const int j = w * 64 + lane;
        double v = (j < d) ? X[row * ldx + j] : 0.0;
        if (v != 0.0 && v != 1.0) atomicExch(bad, 1);
        const unsigned long long m = __ballot(v != 0.0);
        if (lane == 0) words[row * nwords + w] = m;
        pop += __popcll(m);
    }
    if (lane == 0) norms[row] = (double)pop;
}

// ------------------------------------------------------------------ materialised kernel matrix
template <int KIND>
__global__ void k_pairwise(const double* __restrict__ a, const double* __restrict__ an, int64_t m,
                           const double* __restrict__ b, const double* __restrict__ bn,
                           const int32_t* __restrict__ idx, int64_t n, int dt, double os,
                           double* __restrict__ out, int64_t ldo) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = blockIdx.y;
    if (j >= n) return;
    const int64_t c = idx ? (int64_t)idx[j] : j;
    double nx = 0.0, ny = 0.0;
    if constexpr (KIND == SOBER_KIND_TANIMOTO) { nx = an[i]; ny = bn[c]; }
    out[i * ldo + j] = kern_eval_rt<KIND>(a + i * dt, nx, b + c * dt, ny, dt, os);
}

// out[j] = c0 + sum_i k(a_i, b_j) v_i : lanes <-> pool points, a tiles broadcast from LDS
template <int KIND, int DT>
__global__ __launch_bounds__(256) void k_kernel_matvec(
    const double* __restrict__ a, const double* __restrict__ an, const double* __restrict__ v,
    int64_t m, const double* __restrict__ b, const double* __restrict__ bn, int64_t n, double os,
    double c0, double* __restrict__ out) {
    constexpr int TA = 64;
    __shared__ double s_a[TA][DT];
    __shared__ double s_v[TA];
    __shared__ double s_n[TA];
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double y[DT];
    double ny = 0.0;
    if (j < n) {
#pragma unroll
        for (int q = 0; q < DT; ++q) y[q] = b[j * DT + q];
        if constexpr (KIND == SOBER_KIND_TANIMOTO) ny = bn[j];
    } else {
#pragma unroll
        for (int q = 0; q < DT; ++q) y[q] = 0.0;
    }
    double acc = 0.0;
    for (int64_t i0 = 0; i0 < m; i0 += TA) {
        const int cnt = (int)min((int64_t)TA, m - i0);
        __syncthreads();
        for (int t = threadIdx.x; t < cnt * DT; t += blockDim.x) s_a[t / DT][t % DT] = a[i0 * DT + t];
        if (threadIdx.x < cnt) {
            s_v[threadIdx.x] = v[i0 + threadIdx.x];
            s_n[threadIdx.x] = (KIND == SOBER_KIND_TANIMOTO) ? an[i0 + threadIdx.x] : 0.0;
        }
        __syncthreads();
        for (int i = 0; i < cnt; ++i) {
            double xa[DT];
#pragma unroll
            for (int q = 0; q < DT; ++q) xa[q] = s_a[i][q];
            acc = fma(kern_eval<KIND, DT>(xa, s_n[i], y, ny, os), s_v[i], acc);
        }
    }
    if (j < n) out[j] = c0 + acc;
}

// ------------------------------------------------------------------ level plumbing
__global__ void k_sum_partials(const double* __restrict__ partG, const double* __restrict__ partTot,
                               int n_chunks, int n_rows, int ldg, int S,
                               const double* __restrict__ extraG, const double* __restrict__ extraTot,
                               int n_xchunks, int n_xcols,
                               double* __restrict__ G, int ldo, double* __restrict__ tot,
                               const int64_t* __restrict__ dR, int chunked) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = blockIdx.y;            // row == n_rows -> the tot vector
    if (s >= S) return;
    if (dR != nullptr) {                   // queued levels: the chunk counts follow from the exact level size
        const int64_t R = __hip_atomic_load(dR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (R <= S) return;
        const int64_t E = R / S, left = R - E * S;
        // (chunked: the element-chunk kernels -- Tanimoto --, else the matrix-core FP64 kernel's slots per tile)
        n_chunks = chunked ? level_chunks_tani_for(n_rows, (R + S - 1) / S, S) : level_parts_mfma_for(n_rows, (R + S - 1) / S, S);
        n_xchunks = left <= 0 ? 0 : (chunked ? level_chunks_tani_for(n_rows, (left + n_xcols - 1) / n_xcols, n_xcols)
                                             : level_parts_mfma_for(n_rows, (left + n_xcols - 1) / n_xcols, n_xcols));
        if (left <= 0) { extraG = nullptr; extraTot = nullptr; }
    }
    const bool fold = (extraG != nullptr) && (s == S - 1);
    if (row < n_rows) {
        double acc = 0.0;
        for (int c = 0; c < n_chunks; ++c) acc += partG[((size_t)c * n_rows + row) * ldg + s];
        if (fold) {
            double ex = 0.0;
            for (int c = 0; c < n_xchunks; ++c)
                for (int q = 0; q < n_xcols; ++q) ex += extraG[((size_t)c * n_rows + row) * n_xcols + q];
            acc += ex;
        }
        G[(size_t)row * ldo + s] = acc;
    } else if (tot != nullptr && partTot != nullptr) {
        double acc = 0.0;
        for (int c = 0; c < n_chunks; ++c) acc += partTot[(size_t)c * ldg + s];
        if (fold && extraTot != nullptr) {
            double ex = 0.0;
            for (int c = 0; c < n_xchunks; ++c)
                for (int q = 0; q < n_xcols; ++q) ex += extraTot[(size_t)c * n_xcols + q];
            acc += ex;
        }
        tot[s] = acc;
    }
}

__global__ void k_barycentres(const double* __restrict__ Xtr, int ldx, int n, int S,
                              const double* __restrict__ tot, double* __restrict__ X_tmp) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * S) return;
    const int s = t / n, i = t % n;
    double v = Xtr[(size_t)i * ldx + s];
    if (tot) v = v / tot[s];
    X_tmp[t] = v;
}

__global__ void k_level_update(const int32_t* __restrict__ idx_cur, int64_t pos0, int64_t count, int S,
                               int64_t E, const int32_t* __restrict__ keep_rank,
                               const double* __restrict__ w_star, const double* __restrict__ tot,
                               int n_keep, double* __restrict__ mu, int32_t* __restrict__ idx_new,
                               int64_t new_pos0) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const int64_t p = pos0 + t;               // global list position
    const int64_t ES = E * S;
    const int c = idx_cur[t];
    int s;
    int64_t dst;
    if (p < ES) {
        s = (int)(p % S);
        const int k = keep_rank[s];
        dst = (k >= 0) ? (p / S) * n_keep + k : -1;
    } else {                                   // leftovers ride on the last set (:208-218)
        s = S - 1;
        dst = (keep_rank[s] >= 0) ? E * n_keep + (p - ES) : -1;
    }
    if (dst >= 0) {
        const int k = keep_rank[s];
        mu[c] = (mu[c] * w_star[k]) / tot[s];   // multiply, then divide (:204-205, :212-213)
        idx_new[dst - new_pos0] = c;
    } else {
        mu[c] = 0.0;
    }
}

// K7 for a queued level: everything the host of sober_level_loop decides between two levels, decided by every
// workgroup for itself from device memory -- R = *dR_cur, E, the leftovers, n_keep = keep_rank[S], whether the last
// set survived -- and R_new handed to the next level through *dR_next.  Anything the host loop would stop at (no
// progress, a Caratheodory step that gave up, more survivors than the next launches were sized for, nothing left to
// halve) leaves mu and the list UNTOUCHED and stops the chain (*dR_next = -1): the host redoes that level.
__global__ void k_level_update_queued(const int32_t* __restrict__ idx_cur, int S, const int32_t* __restrict__ keep_rank,
                                      const double* __restrict__ w_star, const double* __restrict__ tot,
                                      double* __restrict__ mu, int32_t* __restrict__ idx_new,
                                      const int64_t* __restrict__ dR_cur, int64_t* __restrict__ dR_next,
                                      int64_t R_ub_next, int need_keep, double* __restrict__ cls_scale,
                                      int32_t* __restrict__ cls_sof, int64_t R_known) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // (R_known >= 0: the chain's first level, whose size the host knows -- no launch just to put it into dR[0])
    const int64_t R = R_known >= 0 ? R_known : __hip_atomic_load(dR_cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int64_t E = R > 0 ? R / S : 0, ES = E * S, r = R - ES;
    const int n_keep = keep_rank[S];
    const bool last_kept = keep_rank[S - 1] >= 0;
    const int64_t R_new = E * (int64_t)n_keep + (last_kept ? r : 0);
    // need_keep > 0: the next level's set sums are DERIVED from this level's class sums (level_class.hip), which holds for
    // exactly need_keep = S / 2 kept sets and no leftovers -- anything else stops the chain here, the synchronised loop
    // redoes this level and goes on with evaluated sums
    const bool stop = R <= S || n_keep <= 0 || n_keep > S || R_new >= R || R_new > R_ub_next ||
                      (need_keep > 0 && (n_keep != need_keep || r != 0));
    if (t == 0) *dR_next = stop ? -1 : R_new;
    if (stop) return;
    if (cls_scale != nullptr && t < S) {                        // rank k -> its set and the factor of its survivors' weights
        const int k = keep_rank[t];
        if (k >= 0) { cls_sof[k] = (int32_t)t; cls_scale[k] = w_star[k] / tot[t]; }
    }
    if (t >= R) return;
    const int c = idx_cur[t];
    int s;
    int64_t dst;
    if (t < ES) {
        s = (int)(t % S);
        const int k = keep_rank[s];
        dst = (k >= 0) ? (t / S) * n_keep + k : -1;
    } else {                                   // leftovers ride on the last set (:208-218)
        s = S - 1;
        dst = last_kept ? E * n_keep + (t - ES) : -1;
    }
    if (dst >= 0) {
        mu[c] = (mu[c] * w_star[keep_rank[s]]) / tot[s];   // multiply, then divide (:204-205, :212-213)
        idx_new[dst] = c;
    } else {
        mu[c] = 0.0;
    }
}

__global__ void k_scatter_weights(const int32_t* __restrict__ idx_cur, const int32_t* __restrict__ sel,
                                  const double* __restrict__ w, int n_sel, double* __restrict__ mu,
                                  int64_t* __restrict__ out_idx) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_sel) return;
    const int c = idx_cur[sel[k]];
    mu[c] = w[k];
    out_idx[k] = c;
}

// out[t] = src[idx[t]]  (the weights of the live positions, SOBER/_rchq.py:84)
__global__ void k_gather_f64(const double* __restrict__ src, const int32_t* __restrict__ idx, int64_t n,
                             double* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[t] = src[idx[t]];
}

// Final direct level write-back on the device (SOBER/_rchq.py:108-114; mu[:] = 0 is the caller's memset): position t
// survives iff keep_rank[t] >= 0; its rank k orders the result: mu[idx[t]] = w_star[k], out_idx[k] = idx[t] + row_offset,
// out_w[k] = w_star[k].
__global__ void k_final_scatter(const int32_t* __restrict__ idx, int n, const int32_t* __restrict__ keep_rank,
                                const double* __restrict__ w_star, int64_t row_offset, double* __restrict__ mu,
                                int64_t* __restrict__ out_idx, double* __restrict__ out_w) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int k = keep_rank[t];
    if (k < 0) return;
    const int c = idx[t];
    const double w = w_star[k];
    mu[c] = w;
    out_idx[k] = (int64_t)c + row_offset;
    out_w[k] = w;
}

// mu[:] = 0 (SOBER/_rchq.py:109) -- unless the Caratheodory step in front reported no result (*n_keep < 0): the caller
// then redoes the step on another route with the weights intact
__global__ void k_zero_unless_failed(double* __restrict__ mu, int64_t N, const int32_t* __restrict__ n_keep) {
    if (*n_keep < 0) return;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < N; t += (int64_t)gridDim.x * blockDim.x) mu[t] = 0.0;
}
__global__ void k_final_scatter_guarded(const int32_t* __restrict__ idx, int n, const int32_t* __restrict__ keep_rank,
                                        const double* __restrict__ w_star, const int32_t* __restrict__ n_keep,
                                        int64_t row_offset, double* __restrict__ mu, int64_t* __restrict__ out_idx,
                                        double* __restrict__ out_w) {
    if (*n_keep < 0) return;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int k = keep_rank[t];
    if (k < 0) return;
    const int c = idx[t];
    const double w = w_star[k];
    mu[c] = w;
    out_idx[k] = (int64_t)c + row_offset;
    out_w[k] = w;
}

// mu[:] = 0 and the write-back in ONE launch (round 6): every workgroup keeps the <= 512 kept (candidate, weight) pairs in
// LDS, in rank order -- ascending candidates, because the live list is (a list that is not falls back to a scan) -- and each
// element of mu looks itself up: a binary search of <= 9 steps instead of a second launch behind the zero fill.
__global__ __launch_bounds__(256) void k_final_commit(const int32_t* __restrict__ idx, int n, const int32_t* __restrict__ keep_rank,
                                                      const double* __restrict__ w_star, const int32_t* __restrict__ n_keep,
                                                      int64_t row_offset, double* __restrict__ mu, int64_t N,
                                                      int64_t* __restrict__ out_idx, double* __restrict__ out_w) {
    const int nk = *n_keep;
    if (nk < 0) return;                                      // (the step reported no result: the weights stay what they were)
    __shared__ int s_c[512];
    __shared__ double s_w[512];
    __shared__ int s_unsorted;
    if (threadIdx.x == 0) s_unsorted = 0;
    __syncthreads();
    for (int t = threadIdx.x; t < n; t += blockDim.x) {
        const int k = keep_rank[t];
        if (k >= 0 && k < 512) { s_c[k] = idx[t]; s_w[k] = w_star[k]; }
    }
    __syncthreads();
    for (int k = threadIdx.x + 1; k < nk; k += blockDim.x)
        if (s_c[k] <= s_c[k - 1]) s_unsorted = 1;
    __syncthreads();
    const bool sorted = s_unsorted == 0;
    if (blockIdx.x == 0)
        for (int k = threadIdx.x; k < nk; k += blockDim.x) { out_idx[k] = (int64_t)s_c[k] + row_offset; out_w[k] = s_w[k]; }
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < N; t += (int64_t)gridDim.x * blockDim.x) {
        double v = 0.0;
        if (sorted) {
            int lo = 0, hi = nk;                              // first k with s_c[k] >= t
            while (lo < hi) { const int mid = (lo + hi) >> 1; if ((int64_t)s_c[mid] < t) lo = mid + 1; else hi = mid; }
            if (lo < nk && (int64_t)s_c[lo] == t) v = s_w[lo];
        } else {
            for (int k = 0; k < nk; ++k) if ((int64_t)s_c[k] == t) v = s_w[k];     // (the last rank writing a candidate wins, like the scatter)
        }
        mu[t] = v;
    }
}

// X_for_obj of SOBER/_rchq.py:138-146 plus the leftover addition to the last set (:157-163) for the list positions
// [pos0, pos0 + count): out[s] = sum over the positions p = s (mod S) of obj[c] mu[c]  (c = idx[p - pos0]; leftovers p >= E S
// are counted in set p mod S AND in set S - 1: quirk Q1, like the kernel rows).  One workgroup per set, a thread's elements
// in element order, the 256 partial sums by a fixed tree: no atomics, run-to-run bit-equal.  (It was nine torch launches
// per level of the acquisition-guided branch.)
__global__ __launch_bounds__(256) void k_obj_set_sums(const double* __restrict__ obj, const double* __restrict__ mu,
                                                      const int32_t* __restrict__ idx, int64_t pos0, int64_t count, int S,
                                                      int64_t E, double* __restrict__ out) {
    __shared__ double s_p[256];
    const int s = blockIdx.x, tid = threadIdx.x;
    const int64_t ES = E * (int64_t)S, end = pos0 + count;
    double acc = 0.0;
    // first position >= pos0 that is congruent to s
    int64_t p = pos0 + ((s - pos0 % S) % S + S) % S;
    for (p += (int64_t)tid * S; p < end; p += (int64_t)256 * S) { const int c = idx[p - pos0]; acc += obj[c] * mu[c]; }
    if (s == S - 1)                                          // the leftovers' second placement
        for (int64_t q = max(pos0, ES) + tid; q < end; q += 256) { const int c = idx[q - pos0]; acc += obj[c] * mu[c]; }
    s_p[tid] = acc;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if (tid < h) s_p[tid] += s_p[tid + h];
        __syncthreads();
    }
    if (tid == 0) out[s] = s_p[0];
}

// ... for a QUEUED level (level_exec.cpp: the acquisition-guided branch's chain): the level's size read on the device (*dR, or
// R_known >= 0 for the chain's first level), and -- tot being complete by then -- the objective's barycentre column written where
// the Caratheodory step and the second elimination read it: xcol[s * ldx] (the last column of X_tmp) and ocol[s], both out / tot.
__global__ __launch_bounds__(256) void k_obj_set_sums_queued(const double* __restrict__ obj, const double* __restrict__ mu,
                                                             const int32_t* __restrict__ idx, int S, const int64_t* __restrict__ dR,
                                                             int64_t R_known, const double* __restrict__ tot,
                                                             double* __restrict__ xcol, int ldx, double* __restrict__ ocol) {
    __shared__ double s_p[256];
    const int s = blockIdx.x, tid = threadIdx.x;
    const int64_t R = R_known >= 0 ? R_known : __hip_atomic_load(dR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (R <= S) return;                                      // (the chain has stopped, or nothing left to halve)
    const int64_t ES = (R / S) * (int64_t)S;
    double acc = 0.0;
    for (int64_t p = s + (int64_t)tid * S; p < R; p += (int64_t)256 * S) { const int c = idx[p]; acc += obj[c] * mu[c]; }
    if (s == S - 1)                                          // the leftovers' second placement (Q1)
        for (int64_t q = ES + tid; q < R; q += 256) { const int c = idx[q]; acc += obj[c] * mu[c]; }
    s_p[tid] = acc;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if (tid < h) s_p[tid] += s_p[tid + h];
        __syncthreads();
    }
    if (tid == 0) {
        const double v = s_p[0] / tot[s];
        xcol[(size_t)s * ldx] = v;
        ocol[s] = v;
    }
}

// left block of the projection: P[r][c] = Ut[r][c] * mean[c]  (c < M; ldp = M + n_obs)
__global__ void k_projection_left(const double* __restrict__ Ut, int s, int M, const double* __restrict__ mean,
                                  double* __restrict__ P, int ldp) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= s * M) return;
    const int r = t / M, c = t - r * M;
    P[(size_t)r * ldp + c] = mean ? Ut[t] * mean[c] : Ut[t];
}

// Ut = Q^T and the left block of P from Q itself (one launch for k_barycentres(tot = NULL) + k_projection_left)
__global__ void k_transpose_projection(const double* __restrict__ Q, int s, int M, const double* __restrict__ mean,
                                       double* __restrict__ Ut, double* __restrict__ P, int ldp) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= s * M) return;
    const int r = t / M, c = t - r * M;
    const double v = Q[(size_t)c * s + r];
    Ut[t] = v;
    if (P != nullptr) P[(size_t)r * ldp + c] = mean ? v * mean[c] : v;
}

__global__ void k_i64_to_i32(const int64_t* __restrict__ in, int64_t n, int32_t* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[t] = (int32_t)in[t];
}

// ------------------------------------------------------------------ GP prediction over a pool
// var[j] = kxx_j - sum_i KX[i][j] * V[i][j] + noise   with V = W KX  (SOBER/_gp.py:212-238, exact GP);
// optionally pi[j] = Phi((mean[j] - eta) / sqrt(var[j]))  (SOBER/_pi.py:31-38), log variant adds FP32 eps.
__global__ void k_predict_finish(const double* __restrict__ KX, const double* __restrict__ V, int n_obs,
                                 int64_t N, int64_t ld, const double* __restrict__ mean, double kxx_const,
                                 const double* __restrict__ norms, double outputscale, double noise,
                                 double* __restrict__ var_out, double eta, double* __restrict__ lfi_out,
                                 int log_flag) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N) return;
    double q0 = 0.0, q1 = 0.0;
    int i = 0;
    for (; i + 1 < n_obs; i += 2) {
        q0 = fma(KX[(size_t)i * ld + j], V[(size_t)i * ld + j], q0);
        q1 = fma(KX[(size_t)(i + 1) * ld + j], V[(size_t)(i + 1) * ld + j], q1);
    }
    if (i < n_obs) q0 = fma(KX[(size_t)i * ld + j], V[(size_t)i * ld + j], q0);
    double kxx = kxx_const;
    if (norms != nullptr) {                                   // Tanimoto: k(x, x) = (|x|^2 + eps) / (eps + |x|^2)
        const double n2 = norms[j];
        kxx = (n2 + 1e-6) / (1e-6 + n2) * outputscale;
    }
    const double var = kxx - (q0 + q1) + noise;
    var_out[j] = var;
    if (lfi_out != nullptr) {
        const double z = (mean[j] - eta) / sqrt(var);
        double p = 0.5 * erfc(-z * 0.70710678118654752440);
        if (log_flag) p = log(p + 1.1920928955078125e-07);     // + torch.finfo().eps (FP32 eps, quirk Q5)
        lfi_out[j] = p;
    }
}

// ------------------------------------------------------------------ cleansing_weights
constexpr int CW_BLOCKS = 256, CW_THREADS = 256;

__global__ __launch_bounds__(CW_THREADS) void k_clean_partial(double* __restrict__ w, int64_t n,
                                                              double eps, double* __restrict__ part) {
    __shared__ double s[CW_THREADS];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * CW_THREADS + threadIdx.x; i < n;
         i += (int64_t)CW_BLOCKS * CW_THREADS) {
        double v = w[i];
        if (v < eps) v = 0.0;          // NaN compares false, -inf -> 0  (_weights.py:31)
        if (isinf(v)) v = eps;         // :32
        if (isnan(v)) v = eps;         // :33
        w[i] = v;
        acc += v;
    }
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int o = CW_THREADS / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = s[0];
}

// (every workgroup forms the total from the 256 partial sums itself, by the tree the single-workgroup launch of rounds 1-6 used -- the same bits --: the launch
//  that did only that is gone)
__global__ __launch_bounds__(CW_BLOCKS) void k_clean_scale(double* __restrict__ w, int64_t n, const double* __restrict__ part) {
    __shared__ double s[CW_BLOCKS];
    s[threadIdx.x] = part[threadIdx.x];
    __syncthreads();
    for (int o = CW_BLOCKS / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    const double tot = s[0];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    w[i] = (tot != 0.0) ? w[i] / tot : 1.0 / (double)n;     // :34-37
}

// KDE component draws (SOBER/_wkde.py:162-219): x = X_c + L eps per row, inside = 1 iff lo <= x <= hi
__global__ void k_wkde_draw(const double* __restrict__ eps, int64_t n, int d, const int32_t* __restrict__ comp,
                            const double* __restrict__ Xobs, int ldx, const double* __restrict__ L,
                            const double* __restrict__ lo, const double* __restrict__ hi,
                            double* __restrict__ x, int32_t* __restrict__ inside) {
    extern __shared__ double sL[];                                   // d x d, lower triangle used
    for (int t = threadIdx.x; t < d * d; t += blockDim.x) sL[t] = L[t];
    __syncthreads();
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const double* e = eps + r * d;
    const double* loc = Xobs + (size_t)comp[r] * ldx;
    bool ok = true;
    for (int k = 0; k < d; ++k) {
        double acc = 0.0;
        for (int j = 0; j <= k; ++j) acc = fma(sL[k * d + j], e[j], acc);
        const double v = loc[k] + acc;
        x[r * d + k] = v;
        if (lo) ok = ok && !(v < lo[k]) && !(v > hi[k]);
    }
    if (inside) inside[r] = ok ? 1 : 0;
}

}  // namespace sober

#include <cstdlib>
using namespace sober;

static inline unsigned nblk(int64_t n, int b) { return (unsigned)((n + b - 1) / b); }

extern "C" int sober_abi_version(void) { return SOBER_ABI_VERSION; }
// 1 = this library carries in-kernel stamps (a diagnostic build: never the one the product or the tests load)
extern "C" int sober_diag_build(void) {
#ifdef SOBER_DIAG_BUILD
    return 1;
#else
    return 0;
#endif
}

namespace sober {
static Switches read_switches() {
    Switches w;
    w.level_two_launches = getenv("SOBER_LEVEL_TWO_LAUNCHES") != nullptr;
    w.tani_no_queue = getenv("SOBER_TANI_NO_QUEUE") != nullptr;
    w.car_force_giveup = getenv("SOBER_CAR_FORCE_GIVEUP") != nullptr;
    w.car_unfused = getenv("SOBER_CAR_UNFUSED") != nullptr;
    w.level_no_classes = getenv("SOBER_LEVEL_NO_CLASSES") != nullptr;
    w.car_exact_ratio = getenv("SOBER_CAR_EXACT_RATIO") != nullptr;
    return w;
}
static Switches g_switches = read_switches();       // once, when the library is loaded
const Switches& switches() { return g_switches; }
}  // namespace sober
extern "C" int sober_reload_switches(void) { sober::g_switches = sober::read_switches(); return 0; }
namespace sober {
LaunchEvents& launch_events() {
    static thread_local LaunchEvents le;
    return le;
}
}  // namespace sober
extern "C" int sober_set_launch_events(void* start, void* stop) {
    if ((start == nullptr) != (stop == nullptr)) return SOBER_E_ARG;
    sober::launch_events() = sober::LaunchEvents{(hipEvent_t)start, (hipEvent_t)stop};
    return 0;
}

extern "C" int sober_padded_dim(int d) {
    if (d <= 0) return SOBER_E_ARG;
    const int dts[] = {4, 8, 12, 16, 20, 24, 32};
    for (int v : dts)
        if (d <= v) return v;
    return SOBER_E_DIM;
}

extern "C" int sober_bit_words(int d) {
    if (d <= 0) return SOBER_E_ARG;
    const int w = (d + 63) / 64;
    const int ws[] = {1, 2, 4, 8, 16, 32};
    for (int v : ws)
        if (w <= v) return v;
    return SOBER_E_DIM;
}

// Box-Muller exactly as ATen's CPU normal fill arranges it for a contiguous double tensor (groups of 16: element j < 8
// pairs with element j + 8; u1 = 1 - u[j], u2 = u[j + 8]; radius sqrt(-2 log u1), angle 2 pi u2; cos to j, sin to
// j + 8), the last 16 elements recomputed from 16 further uniforms when numel is not a multiple of 16.  u holds the
// uniforms of sober_mt19937_uniform53 (numel, + 16 for that tail).  The values agree with torch.randn's to the last
// ulp or two of the device's log / sincos -- the test matrix of a randomised range finder, nothing downstream
// resolves that.
__global__ void k_box_muller(const double* __restrict__ u, int64_t numel, double* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // one pair per thread
    const int64_t full = numel / 16;                                      // complete groups
    const bool tail = (numel % 16) != 0;
    const int64_t npairs = (full + (tail ? 1 : 0)) * 8;
    if (t >= npairs) return;
    const int64_t g = t >> 3;
    const int j = (int)(t & 7);
    // the tail group reads its own 16 uniforms and lands on the last 16 elements; a complete group that overlaps
    // them leaves those elements to it
    const bool is_tail = g == full;
    const int64_t ub = is_tail ? numel : g * 16, ob = is_tail ? numel - 16 : g * 16;
    const double u1 = 1.0 - u[ub + j], u2 = u[ub + j + 8];
    const double radius = sqrt(-2.0 * log(u1));
    const double theta = 6.283185307179586 * u2;
    double sn, cs;
    sincos(theta, &sn, &cs);
    const int64_t keep_below = tail ? numel - 16 : numel;
    if (is_tail || ob + j < keep_below) out[ob + j] = radius * cs;
    if (is_tail || ob + j + 8 < keep_below) out[ob + j + 8] = radius * sn;
}

extern "C" int sober_box_muller(const double* u, int64_t numel, double* out, void* stream) {
    if (!u || !out || numel < 16) return SOBER_E_ARG;
    const int64_t npairs = ((numel + 15) / 16) * 8;
    hipLaunchKernelGGL(k_box_muller, dim3(nblk(npairs, 256)), dim3(256), 0, (hipStream_t)stream, u, numel, out);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int sober_scale_points(const double* X, int64_t n, int d, int64_t ldx,
                                  const double* lengthscale, int ls_len, double* out, int dt,
                                  void* stream) {
    if (!X || !lengthscale || !out || n <= 0 || d <= 0 || dt < d || ldx < d) return SOBER_E_ARG;
    if (ls_len != 1 && ls_len != d) return SOBER_E_ARG;
    hipLaunchKernelGGL(k_scale_points, dim3(nblk(n * dt, 256)), dim3(256), 0, (hipStream_t)stream, X,
                       n, d, ldx, lengthscale, ls_len, out, dt);
    LAUNCH_CHECK();
    return 0;
}

int sober::scale_points2(const double* Xa, int64_t na, int64_t lda, const double* Xb, int64_t nb, int64_t ldb, int d,
                         const double* lengthscale, int ls_len, double* out, int dt, void* stream) {
    if (!Xa || !Xb || !lengthscale || !out || na <= 0 || nb <= 0 || d <= 0 || dt < d || lda < d || ldb < d) return SOBER_E_ARG;
    if (ls_len != 1 && ls_len != d) return SOBER_E_ARG;
    hipLaunchKernelGGL(k_scale_points2, dim3(nblk((na + nb) * dt, 256)), dim3(256), 0, (hipStream_t)stream, Xa, na, lda, Xb, nb,
                       ldb, d, lengthscale, ls_len, out, dt);
    LAUNCH_CHECK();
    return 0;
}

int sober::transpose_projection(const double* Q, int s, int M, const double* mean, double* Ut, double* P, int ldp, void* stream) {
    if (!Q || !Ut || s <= 0 || M <= 0 || (P && ldp < M)) return SOBER_E_ARG;
    hipLaunchKernelGGL(k_transpose_projection, dim3(nblk((int64_t)s * M, 256)), dim3(256), 0, (hipStream_t)stream, Q, s, M, mean,
                       Ut, P, ldp);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int sober_scale_points_idx(const double* X, const int32_t* idx, int64_t n, int d, int64_t ldx,
                                      const double* lengthscale, int ls_len, double* out, int dt, const double* gather_src,
                                      double* gather_out, void* stream) {
    if (!X || !idx || !lengthscale || !out || n <= 0 || d <= 0 || dt < d || ldx < d) return SOBER_E_ARG;
    if ((ls_len != 1 && ls_len != d) || (gather_src && !gather_out)) return SOBER_E_ARG;
    hipLaunchKernelGGL(k_scale_points_idx, dim3(nblk(n * dt, 256)), dim3(256), 0, (hipStream_t)stream, X, idx, n, d, ldx,
                       lengthscale, ls_len, out, dt, gather_src, gather_out);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int sober_pack_bits(const double* X, int64_t n, int d, int64_t ldx, uint64_t* words,
                               int nwords, double* norms, int32_t* bad_flag, void* stream) {
    if (!X || !words || !norms || !bad_flag || n <= 0 || d <= 0 || nwords * 64 < d || ldx < d)
        return SOBER_E_ARG;
    hipLaunchKernelGGL(k_pack_bits, dim3(nblk(n, 4)), dim3(256), 0, (hipStream_t)stream, X, n, d, ldx,
                       words, nwords, norms, bad_flag);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int sober_pairwise(int kind, const void* a, const double* a_norm, int64_t m, const void* b,
                              const double* b_norm, const int32_t* idx, int64_t n, int dt,
                              double outputscale, double* out, int64_t ldo, void* stream) {
    if (!a || !b || !out || m <= 0 || n <= 0 || dt <= 0 || ldo < n || m > 65535) return SOBER_E_ARG;
    if (kind == SOBER_KIND_TANIMOTO && (!a_norm || !b_norm)) return SOBER_E_ARG;
    dim3 grid(nblk(n, 256), (unsigned)m);
    hipStream_t st = (hipStream_t)stream;
    const double* A = (const double*)a;
    const double* B = (const double*)b;
    switch (kind) {
        case SOBER_KIND_RBF:
            hipLaunchKernelGGL(k_pairwise<SOBER_KIND_RBF>, grid, dim3(256), 0, st, A, a_norm, m, B,
                               b_norm, idx, n, dt, outputscale, out, ldo);
            break;
        case SOBER_KIND_MATERN52:
            hipLaunchKernelGGL(k_pairwise<SOBER_KIND_MATERN52>, grid, dim3(256), 0, st, A, a_norm, m, B,
                               b_norm, idx, n, dt, outputscale, out, ldo);
            break;
        case SOBER_KIND_TANIMOTO:
            hipLaunchKernelGGL(k_pairwise<SOBER_KIND_TANIMOTO>, grid, dim3(256), 0, st, A, a_norm, m, B,
                               b_norm, idx, n, dt, outputscale, out, ldo);
            break;
        default: return SOBER_E_ARG;
    }
    LAUNCH_CHECK();
    return 0;
}

template <int KIND, int DT>
static int launch_mv(const void* a, const double* an, const double* v, int64_t m, const void* b,
                     const double* bn, int64_t n, double os, double c0, double* out, hipStream_t st) {
    hipLaunchKernelGGL((k_kernel_matvec<KIND, DT>), dim3(nblk(n, 256)), dim3(256), 0, st,
                       (const double*)a, an, v, m, (const double*)b, bn, n, os, c0, out);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int sober_kernel_matvec(int kind, const void* a, const double* a_norm, const double* v,
                                   int64_t m, const void* b, const double* b_norm, int64_t n, int dt,
                                   double outputscale, double c0, double* out, void* stream) {
    if (!a || !v || !b || !out || m <= 0 || n <= 0) return SOBER_E_ARG;
    if (kind == SOBER_KIND_TANIMOTO && (!a_norm || !b_norm)) return SOBER_E_ARG;
    hipStream_t st = (hipStream_t)stream;
#define MV_CASE(K, D) \
    case D: return launch_mv<K, D>(a, a_norm, v, m, b, b_norm, n, outputscale, c0, out, st);
    switch (kind) {
        case SOBER_KIND_RBF:
            switch (dt) { MV_CASE(SOBER_KIND_RBF, 4) MV_CASE(SOBER_KIND_RBF, 8) MV_CASE(SOBER_KIND_RBF, 12)
                          MV_CASE(SOBER_KIND_RBF, 16) MV_CASE(SOBER_KIND_RBF, 20)
                          MV_CASE(SOBER_KIND_RBF, 24) MV_CASE(SOBER_KIND_RBF, 32)
                          default: return SOBER_E_DIM; }
        case SOBER_KIND_MATERN52:
            switch (dt) { MV_CASE(SOBER_KIND_MATERN52, 4) MV_CASE(SOBER_KIND_MATERN52, 8)
                          MV_CASE(SOBER_KIND_MATERN52, 12) MV_CASE(SOBER_KIND_MATERN52, 16)
                          MV_CASE(SOBER_KIND_MATERN52, 20) MV_CASE(SOBER_KIND_MATERN52, 24)
                          MV_CASE(SOBER_KIND_MATERN52, 32) default: return SOBER_E_DIM; }
        case SOBER_KIND_TANIMOTO:
            switch (dt) { MV_CASE(SOBER_KIND_TANIMOTO, 1) MV_CASE(SOBER_KIND_TANIMOTO, 2)
                          MV_CASE(SOBER_KIND_TANIMOTO, 4) MV_CASE(SOBER_KIND_TANIMOTO, 8)
                          MV_CASE(SOBER_KIND_TANIMOTO, 16) MV_CASE(SOBER_KIND_TANIMOTO, 32)
                          default: return SOBER_E_DIM; }
        default: return SOBER_E_ARG;
    }
#undef MV_CASE
}

extern "C" int sober_level_update_queued(const int32_t* idx_cur, int64_t R_ub, int S, const int32_t* keep_rank,
                                         const double* w_star, const double* tot, double* mu, int32_t* idx_new,
                                         const int64_t* dR_cur, int64_t* dR_next, int64_t R_ub_next, void* stream) {
    if (!idx_cur || !keep_rank || !w_star || !tot || !mu || !idx_new || !dR_cur || !dR_next || R_ub <= 0 || S <= 0)
        return SOBER_E_ARG;
    hipLaunchKernelGGL(k_level_update_queued, dim3(nblk(R_ub, 256)), dim3(256), 0, (hipStream_t)stream, idx_cur, S,
                       keep_rank, w_star, tot, mu, idx_new, dR_cur, dR_next, R_ub_next, 0, (double*)nullptr, (int32_t*)nullptr, (int64_t)-1);
    LAUNCH_CHECK();
    return 0;
}

// the general form: need_keep = 0 / cls_* = NULL for a level whose successor is evaluated; R_known >= 0: this level's size by
// value (the chain's first level) instead of *dR_cur
extern "C" int sober_level_update_queued_ex(const int32_t* idx_cur, int64_t R_ub, int S, const int32_t* keep_rank,
                                            const double* w_star, const double* tot, double* mu, int32_t* idx_new,
                                            const int64_t* dR_cur, int64_t* dR_next, int64_t R_ub_next, int need_keep,
                                            double* cls_scale, int32_t* cls_sof, int64_t R_known, void* stream) {
    if (!idx_cur || !keep_rank || !w_star || !tot || !mu || !idx_new || (!dR_cur && R_known < 0) || !dR_next || R_ub <= 0 ||
        S <= 0 || need_keep < 0 || need_keep > S || (need_keep > 0 && (!cls_scale || !cls_sof)))
        return SOBER_E_ARG;
    hipLaunchKernelGGL(k_level_update_queued, dim3(nblk(R_ub, 256)), dim3(256), 0, (hipStream_t)stream, idx_cur, S,
                       keep_rank, w_star, tot, mu, idx_new, dR_cur, dR_next, R_ub_next, need_keep,
                       need_keep > 0 ? cls_scale : nullptr, need_keep > 0 ? cls_sof : nullptr, R_known);
    LAUNCH_CHECK();
    return 0;
}

// ... of a level whose successor's set sums are derived from class sums (level_class.hip): the chain stops unless exactly
// need_keep sets survived and the level had no leftovers; cls_sof[k] / cls_scale[k] = the set of rank k and w*_k / tot_k
extern "C" int sober_level_update_queued_cls(const int32_t* idx_cur, int64_t R_ub, int S, const int32_t* keep_rank,
                                             const double* w_star, const double* tot, double* mu, int32_t* idx_new,
                                             const int64_t* dR_cur, int64_t* dR_next, int64_t R_ub_next, int need_keep,
                                             double* cls_scale, int32_t* cls_sof, void* stream) {
    if (!idx_cur || !keep_rank || !w_star || !tot || !mu || !idx_new || !dR_cur || !dR_next || R_ub < S || S <= 0 ||
        need_keep <= 0 || need_keep > S || !cls_scale || !cls_sof)
        return SOBER_E_ARG;
    hipLaunchKernelGGL(k_level_update_queued, dim3(nblk(R_ub, 256)), dim3(256), 0, (hipStream_t)stream, idx_cur, S,
                       keep_rank, w_star, tot, mu, idx_new, dR_cur, dR_next, R_ub_next, need_keep, cls_scale, cls_sof, (int64_t)-1);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int sober_sum_partials(const double* partG, const double* partTot, int n_chunks,
                                  int n_rows, int ldg, int S, const double* extraG,
                                  const double* extraTot, int n_xchunks, int n_xcols, double* G,
                                  int ldo, double* tot, void* stream) {
    if (!partG || !G || n_chunks <= 0 || n_rows <= 0 || S <= 0 || ldg < S || ldo < S) return SOBER_E_ARG;
    if (extraG && (n_xchunks <= 0 || n_xcols <= 0)) return SOBER_E_ARG;
    dim3 grid(nblk(S, 64), (unsigned)(n_rows + 1));
    hipLaunchKernelGGL(k_sum_partials, grid, dim3(64), 0, (hipStream_t)stream, partG, partTot, n_chunks,
                       n_rows, ldg, S, extraG, extraTot, n_xchunks, n_xcols, G, ldo, tot, (const int64_t*)nullptr, 0);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int sober_sum_partials_queued(const double* partG, const double* partTot, int n_rows, int ldg, int S,
                                         const double* extraG, const double* extraTot, int n_xcols, double* G,
                                         int ldo, double* tot, const int64_t* dR, int chunked, void* stream) {
    if (!partG || !partTot || !extraG || !extraTot || !G || !tot || !dR || n_rows <= 0 || S <= 0 || ldg < S || ldo < S ||
        n_xcols <= 0)
        return SOBER_E_ARG;
    dim3 grid(nblk(S, 64), (unsigned)(n_rows + 1));
    hipLaunchKernelGGL(k_sum_partials, grid, dim3(64), 0, (hipStream_t)stream, partG, partTot, 0, n_rows, ldg, S,
                       extraG, extraTot, 0, n_xcols, G, ldo, tot, dR, chunked);
    LAUNCH_CHECK();
    return 0;
}


extern "C" int sober_barycentres(const double* Xtr, int ldx, int n, int S, const double* tot,
                                 double* X_tmp, void* stream) {
    if (!Xtr || !X_tmp || n <= 0 || S <= 0 || ldx < S) return SOBER_E_ARG;
    hipLaunchKernelGGL(k_barycentres, dim3(nblk((int64_t)n * S, 256)), dim3(256), 0,
                       (hipStream_t)stream, Xtr, ldx, n, S, tot, X_tmp);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int sober_level_update(const int32_t* idx_cur, int64_t pos0, int64_t count, int S, int64_t E,
                                  const int32_t* keep_rank, const double* w_star, const double* tot,
                                  int n_keep, double* mu, int32_t* idx_new, int64_t new_pos0,
                                  void* stream) {
    if (!idx_cur || !keep_rank || !w_star || !tot || !mu || !idx_new) return SOBER_E_ARG;
    if (pos0 < 0 || count <= 0 || S <= 0 || E <= 0 || n_keep < 0 || n_keep > S || new_pos0 < 0)
        return SOBER_E_ARG;
    hipLaunchKernelGGL(k_level_update, dim3(nblk(count, 256)), dim3(256), 0, (hipStream_t)stream,
                       idx_cur, pos0, count, S, E, keep_rank, w_star, tot, n_keep, mu, idx_new, new_pos0);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int sober_scatter_weights(const int32_t* idx_cur, const int32_t* sel, const double* w,
                                     int n_sel, double* mu, int64_t* out_idx, void* stream) {
    if (!idx_cur || !sel || !w || !mu || !out_idx || n_sel <= 0) return SOBER_E_ARG;
    hipLaunchKernelGGL(k_scatter_weights, dim3(nblk(n_sel, 256)), dim3(256), 0, (hipStream_t)stream,
                       idx_cur, sel, w, n_sel, mu, out_idx);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int sober_gather_f64(const double* src, const int32_t* idx, int64_t n, double* out, void* stream) {
    if (!src || !idx || !out || n <= 0) return SOBER_E_ARG;
    hipLaunchKernelGGL(k_gather_f64, dim3(nblk(n, 256)), dim3(256), 0, (hipStream_t)stream, src, idx, n, out);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int sober_final_scatter(const int32_t* idx, int n, const int32_t* keep_rank, const double* w_star,
                                   int64_t row_offset, double* mu, int64_t* out_idx, double* out_w, void* stream) {
    if (!idx || !keep_rank || !w_star || !mu || !out_idx || !out_w || n <= 0) return SOBER_E_ARG;
    hipLaunchKernelGGL(k_final_scatter, dim3(nblk(n, 256)), dim3(256), 0, (hipStream_t)stream, idx, n, keep_rank, w_star,
                       row_offset, mu, out_idx, out_w);
    LAUNCH_CHECK();
    return 0;
}

__global__ void k_set_i64(int64_t* dst, int64_t v) { *dst = v; }

// rows of X (N x n, ld) and entries of ocol whose rank is 0 .. n1-1 go to that rank's place (Xp: n1 x n, objp: n1)
__global__ void k_rank_scatter(const double* __restrict__ X, int ldx, int N, int n, const double* __restrict__ ocol,
                               const int32_t* __restrict__ rank, int n1, double* __restrict__ Xp, double* __restrict__ objp) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)N * (n + 1)) return;
    const int s = (int)(t / (n + 1)), j = (int)(t - (int64_t)s * (n + 1));
    const int r = rank[s];
    if (r < 0 || r >= n1) return;
    if (j < n) Xp[(size_t)r * n + j] = X[(size_t)s * ldx + j];
    else objp[r] = ocol[s];
}

// the survivors of a Caratheodory step in rank order: Xp[rank[s]] = X[s, 0:n], objp[rank[s]] = ocol[s] for the sets s whose
// rank is 0 .. n1-1 (the acquisition-guided branch's second step, SOBER/_rchq.py:87-91 / :177-181, without a host decision)
extern "C" int sober_rank_scatter(const double* X, int ldx, int N, int n, const double* ocol, const int32_t* rank, int n1,
                                  double* Xp, double* objp, void* stream) {
    if (!X || !ocol || !rank || !Xp || !objp || N <= 0 || n <= 0 || n1 <= 0 || ldx < n) return SOBER_E_ARG;
    hipLaunchKernelGGL(k_rank_scatter, dim3(nblk((int64_t)N * (n + 1), 256)), dim3(256), 0, (hipStream_t)stream, X, ldx, N, n, ocol,
                       rank, n1, Xp, objp);
    LAUNCH_CHECK();
    return 0;
}

// *dst = v on the stream, as a one-thread kernel: a hipMemcpyAsync of 8 host bytes goes through the copy path and
// holds the stream for ~20 us in front of the first queued level
extern "C" int sober_set_i64(int64_t* dst, int64_t v, void* stream) {
    if (!dst) return SOBER_E_ARG;
    hipLaunchKernelGGL(k_set_i64, dim3(1), dim3(1), 0, (hipStream_t)stream, dst, v);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int sober_final_commit(const int32_t* idx, int n, const int32_t* keep_rank, const double* w_star,
                                  const int32_t* n_keep, int64_t row_offset, double* mu, int64_t N, int64_t* out_idx,
                                  double* out_w, void* stream) {
    if (!idx || !keep_rank || !w_star || !n_keep || !mu || !out_idx || !out_w || n <= 0 || N <= 0) return SOBER_E_ARG;
    int64_t nb = nblk(N, 256);
    if (nb > 4096) nb = 4096;
    if (n <= 512) {
        hipLaunchKernelGGL(k_final_commit, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, idx, n, keep_rank, w_star, n_keep,
                           row_offset, mu, N, out_idx, out_w);
        LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(k_zero_unless_failed, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, mu, N, n_keep);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_final_scatter_guarded, dim3(nblk(n, 256)), dim3(256), 0, (hipStream_t)stream, idx, n, keep_rank,
                       w_star, n_keep, row_offset, mu, out_idx, out_w);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int sober_obj_set_sums(const double* obj, const double* mu, const int32_t* idx, int64_t pos0, int64_t count, int S,
                                  int64_t E, double* out, void* stream) {
    if (!obj || !mu || !idx || !out || pos0 < 0 || count <= 0 || S <= 0 || E < 0) return SOBER_E_ARG;
    hipLaunchKernelGGL(k_obj_set_sums, dim3((unsigned)S), dim3(256), 0, (hipStream_t)stream, obj, mu, idx, pos0, count, S, E, out);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int sober_obj_set_sums_queued(const double* obj, const double* mu, const int32_t* idx, int S, const int64_t* dR,
                                        int64_t R_known, const double* tot, double* xcol, int ldx, double* ocol, void* stream) {
    if (!obj || !mu || !idx || !tot || !xcol || !ocol || S <= 0 || ldx <= 0 || (!dR && R_known < 0)) return SOBER_E_ARG;
    hipLaunchKernelGGL(k_obj_set_sums_queued, dim3((unsigned)S), dim3(256), 0, (hipStream_t)stream, obj, mu, idx, S, dR, R_known,
                       tot, xcol, ldx, ocol);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int sober_projection(const double* Ut, int s, int M, const double* mean, const double* T, int n_obs, double* P,
                                void* stream) {
    if (!Ut || !P || s <= 0 || M <= 0 || (T && n_obs <= 0)) return SOBER_E_ARG;
    const int ldp = M + (T ? n_obs : 0);
    hipLaunchKernelGGL(k_projection_left, dim3(nblk((int64_t)s * M, 256)), dim3(256), 0, (hipStream_t)stream, Ut, s, M,
                       mean, P, ldp);
    LAUNCH_CHECK();
    if (!T) return 0;
    return sober_dgemm(0, 0, s, n_obs, M, -1.0, P, ldp, T, n_obs, 0.0, P + M, ldp, stream);
}

extern "C" int sober_i64_to_i32(const int64_t* in, int64_t n, int32_t* out, void* stream) {
    if (!in || !out || n <= 0) return SOBER_E_ARG;
    hipLaunchKernelGGL(k_i64_to_i32, dim3(nblk(n, 256)), dim3(256), 0, (hipStream_t)stream, in, n, out);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int sober_predict_finish(const double* KX, const double* V, int n_obs, int64_t N, int64_t ld,
                                    const double* mean, double kxx_const, const double* norms,
                                    double outputscale, double noise, double* var_out, double eta,
                                    double* lfi_out, int log_flag, void* stream) {
    if (!KX || !V || !var_out || n_obs <= 0 || N <= 0 || ld < N) return SOBER_E_ARG;
    if (lfi_out && !mean) return SOBER_E_ARG;
    hipLaunchKernelGGL(k_predict_finish, dim3(nblk(N, 256)), dim3(256), 0, (hipStream_t)stream, KX, V, n_obs, N, ld,
                       mean, kxx_const, norms, outputscale, noise, var_out, eta, lfi_out, log_flag);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int64_t sober_reduce_ws_bytes(int64_t n) { (void)n; return (CW_BLOCKS + 1) * sizeof(double); }

extern "C" int sober_cleansing_weights(double* w, int64_t n, double eps, void* ws, int64_t ws_bytes,
                                       void* stream) {
    if (!w || !ws || n <= 0) return SOBER_E_ARG;
    if (ws_bytes < sober_reduce_ws_bytes(n)) return SOBER_E_WS;
    hipStream_t st = (hipStream_t)stream;
    double* part = (double*)ws;
    hipLaunchKernelGGL(k_clean_partial, dim3(CW_BLOCKS), dim3(CW_THREADS), 0, st, w, n, eps, part);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_clean_scale, dim3(nblk(n, CW_BLOCKS)), dim3(CW_BLOCKS), 0, st, w, n, part);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int sober_wkde_draw(const double* eps, int64_t n, int d, const int32_t* comp, const double* Xobs,
                               int ldx, const double* L, const double* lo, const double* hi, double* x,
                               int32_t* inside, void* stream) {
    if (!eps || !comp || !Xobs || !L || !x || n <= 0 || d <= 0 || d > 64 || ldx < d) return SOBER_E_ARG;
    if ((lo == nullptr) != (hi == nullptr) || (lo && !inside)) return SOBER_E_ARG;
    hipLaunchKernelGGL(k_wkde_draw, dim3(nblk(n, 256)), dim3(256), (size_t)d * d * sizeof(double),
                       (hipStream_t)stream, eps, n, d, comp, Xobs, ldx, L, lo, hi, x, inside);
    LAUNCH_CHECK();
    return 0;
}
