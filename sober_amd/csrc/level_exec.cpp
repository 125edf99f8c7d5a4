// Level executor: the launch sequence of one level of the halving loop behind one C call each side of the
// (optional) all-reduce.  Pure host code: it only sequences the library's own entry points on the caller's
// stream, so a level costs two calls from the host language instead of ten.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#include "../../include/sober_hip.h"
#include "switches.hpp"

extern "C" int sober_level_job_size(void) { return (int)sizeof(sober_level_job); }
extern "C" int sober_obj_job_size(void) { return (int)sizeof(sober_obj_job); }

#define LX_TRY(call)                 \
    do {                             \
        const int rc_ = (call);      \
        if (rc_ != 0) return rc_;    \
    } while (0)

// The event pair of a level-kernel launch: the matrix-core variants carry it in the dispatch (the kernel's own begin /
// end timestamps: sober_set_launch_events), the others get a record on the stream either side of the launch.
static inline bool lx_timed_in_dispatch(const sober_level_job* j) {
    return j->variant == SOBER_LEVEL_MFMA || j->variant == SOBER_LEVEL_TANI;
}
#define LX_EVENTS_BEFORE(A, B)                                                                   \
    if ((A) && (B)) {                                                                            \
        if (lx_timed_in_dispatch(j)) { LX_TRY(sober_set_launch_events((A), (B))); }              \
        else { const hipError_t e_ = hipEventRecord((hipEvent_t)(A), (hipStream_t)stream); if (e_ != hipSuccess) return (int)e_; } \
    }
#define LX_EVENTS_AFTER(A, B)                                                                    \
    if ((A) && (B) && !lx_timed_in_dispatch(j)) {                                                \
        const hipError_t e_ = hipEventRecord((hipEvent_t)(B), (hipStream_t)stream);              \
        if (e_ != hipSuccess) return (int)e_;                                                    \
    }

static int lx_reduce(const sober_level_job* j, const int32_t* idx, int64_t pos0, int64_t count, int S, int n_chunks,
                     double* partG, double* partTot, int64_t tot_limit, void* stream) {
    switch (j->variant) {
        case SOBER_LEVEL_MFMA:
            return sober_level_reduce_mfma(j->kind, (const double*)j->rows, j->n_rows, (const double*)j->cand, j->dim,
                                           idx, pos0, count, S, j->mu, j->wmul, j->outputscale, n_chunks, partG, S, 0,
                                           partTot, tot_limit, stream);
        case SOBER_LEVEL_VALU:
            return sober_level_reduce(j->kind, j->rows, j->rows_norm, j->n_rows, j->cand, j->cand_norm, j->dim, idx,
                                      pos0, count, S, j->mu, j->wmul, j->outputscale, n_chunks, partG, S, 0, partTot,
                                      tot_limit, stream);
        case SOBER_LEVEL_TANI:
            return sober_level_reduce_tani(j->rows, j->rows_norm, j->n_rows, j->cand, j->cand_norm, j->dim, idx, pos0, count,
                                           S, j->mu, j->wmul, j->outputscale, n_chunks, partG, S, 0, partTot, tot_limit,
                                           stream);
        case SOBER_LEVEL_GATHER:
            return sober_level_gather((const double*)j->cand, j->n_rows, j->kmat_ld, idx, pos0, count, S, j->mu,
                                      j->wmul, n_chunks, partG, S, 0, partTot, tot_limit, stream);
        default:
            return SOBER_E_ARG;
    }
}

// ---- levels derived from class sums (csrc/level_class.hip) ------------------------------------------------------------
// D for a pool of R live positions: no leftovers at levels 0 .. D (R % (2^D S) == 0), at least two elements per set left
// at level D (so that level D is one of the queued ones), 2^D x the class launch's slots within the partial-sum buffers.
extern "C" int sober_level_class_depth(int variant, int n_rows, int64_t R, int S) {
    if (n_rows <= 0 || R <= 0 || S <= 0 || (S & 1) || sober::switches().level_no_classes) return 0;
    if (variant != SOBER_LEVEL_MFMA && variant != SOBER_LEVEL_TANI) return 0;
    int D = 0;
    while (D < SOBER_CLASS_MAX_DEPTH) {
        const int64_t SC = (int64_t)S << (D + 1);
        if (R % SC != 0 || R / SC < 2) break;
        int slots;
        if (variant == SOBER_LEVEL_MFMA) {
            const int wpt = sober_level_class_wpt(n_rows, R / SC, (int)SC);
            slots = wpt > 0 ? sober_level_class_slots(wpt) : 0;
        } else {
            // the fingerprint kernel runs one workgroup per compute unit in (ideally) one round: a class launch with clearly
            // fewer workgroups than the level's own launch costs level 0 more than the derived level saves
            slots = sober_level_chunks_tani(n_rows, 0, R, (int)SC);
            const int64_t rb = (n_rows + 255) / 256;
            const int64_t wg_c = (int64_t)slots * ((SC + 15) / 16 + 1) * rb;
            const int64_t wg_0 = (int64_t)sober_level_chunks_tani(n_rows, 0, R, S) * ((S + 15) / 16 + 1) * rb;
            if (slots > 0 && wg_c * 100 < wg_0 * 85) break;
        }
        if (slots <= 0 || slots * (1 << (D + 1)) > SOBER_LEVEL_MAX_CHUNKS) break;
        ++D;
    }
    return D;
}

// level 0 of such a pool: the level kernel over 2^D S sets, then class sums + their fold (G, tot: what every level has)
static int lx_class_first(const sober_level_job* j, void* stream) {
    const int S = j->S, D = j->class_depth, CL = 1 << D, SC = S * CL;
    if ((j->variant != SOBER_LEVEL_MFMA && j->variant != SOBER_LEVEL_TANI) || D < 1 || D > SOBER_CLASS_MAX_DEPTH ||
        j->pos0 != 0 || j->count % SC != 0 || j->count / SC < 2 || !j->Gc[0] || !j->totc[0])
        return SOBER_E_ARG;
    int nch;
    if (j->variant == SOBER_LEVEL_MFMA) {
        const int wpt = sober_level_class_wpt(j->n_rows, j->count / SC, SC);
        if (wpt <= 0) return wpt < 0 ? wpt : SOBER_E_ARG;
        nch = sober_level_class_slots(wpt);
        if (nch <= 0 || nch * CL > SOBER_LEVEL_MAX_CHUNKS) return SOBER_E_WS;
        LX_EVENTS_BEFORE(j->ev[0], j->ev[1])
        LX_TRY(sober_level_reduce_mfma_wpt(j->kind, (const double*)j->rows, j->n_rows, (const double*)j->cand, j->dim, j->idx,
                                           j->count, SC, j->mu, j->wmul, j->outputscale, wpt, nch, j->partG, SC, j->partTot,
                                           stream));
    } else {
        nch = sober_level_chunks_tani(j->n_rows, 0, j->count, SC);
        if (nch <= 0 || nch * CL > SOBER_LEVEL_MAX_CHUNKS) return nch <= 0 ? nch : SOBER_E_WS;
        LX_EVENTS_BEFORE(j->ev[0], j->ev[1])
        LX_TRY(sober_level_reduce_tani(j->rows, j->rows_norm, j->n_rows, j->cand, j->cand_norm, j->dim, j->idx, 0, j->count, SC,
                                       j->mu, j->wmul, j->outputscale, nch, j->partG, SC, 0, j->partTot, j->count, stream));
    }
    return sober_class_sum(j->partG, j->partTot, nch, j->n_rows, S, CL, j->Gc[0], j->totc[0], j->G, j->tot, stream);
}

extern "C" int sober_level_moments(const sober_level_job* j, void* stream) {
    if (!j || !j->G || !j->tot || j->n_rows <= 0 || j->S <= 0 || j->n <= 0 || j->phase < 0 || j->phase > 2)
        return SOBER_E_ARG;
    const int S = j->S;
    if (j->phase == 2) {
        if (!j->P || !j->Xtr) return SOBER_E_ARG;
        return sober_dgemm(0, 0, j->n, S, j->n_rows, 1.0, j->P, j->n_rows, j->G, S, 0.0, j->Xtr, S, stream);
    }
    if (!j->cand || !j->idx || !j->mu || !j->partG || !j->partTot || (j->phase == 0 && (!j->P || !j->Xtr)))
        return SOBER_E_ARG;
    if (j->count <= 0 || j->pos0 < 0 || j->E < 0) return SOBER_E_ARG;
    if (j->phase == 1 && j->class_depth > 0) return lx_class_first(j, stream);
    const int64_t ES = j->E * S;
    const bool mfma = j->variant == SOBER_LEVEL_MFMA;
    const bool tani_k = j->variant == SOBER_LEVEL_TANI;
    const int n_chunks = mfma ? sober_level_parts_mfma(j->n_rows, j->pos0, j->count, S)
                              : (tani_k ? sober_level_chunks_tani(j->n_rows, j->pos0, j->count, S)
                                        : sober_level_chunks(j->n_rows, j->pos0, j->count, S));
    if (n_chunks <= 0 || n_chunks > SOBER_LEVEL_MAX_CHUNKS) return n_chunks <= 0 ? n_chunks : SOBER_E_WS;
    // first placement: every live position, set = p mod S (leftovers land in sets 0..r-1, quirk Q1); tot over p < ES
    LX_EVENTS_BEFORE(j->ev[0], j->ev[1])
    LX_TRY(lx_reduce(j, j->idx, j->pos0, j->count, S, n_chunks, j->partG, j->partTot, ES, stream));
    LX_EVENTS_AFTER(j->ev[0], j->ev[1])
    // second placement of the leftovers (SOBER/_rchq.py:153-164): the same kernel over the leftover positions
    // alone, spread over XS pseudo-sets that sum_partials folds into set S-1
    const int64_t lo = j->pos0 > ES ? j->pos0 : ES;
    const int64_t n_left = j->pos0 + j->count - lo;
    int n_xchunks = 0;
    if (n_left > 0) {
        if (!j->extraG || !j->extraTot) return SOBER_E_ARG;
        n_xchunks = mfma ? sober_level_parts_mfma(j->n_rows, 0, n_left, SOBER_LEVEL_XS)
                         : (tani_k ? sober_level_chunks_tani(j->n_rows, 0, n_left, SOBER_LEVEL_XS)
                                   : sober_level_chunks(j->n_rows, 0, n_left, SOBER_LEVEL_XS));
        if (n_xchunks <= 0 || n_xchunks > SOBER_LEVEL_MAX_CHUNKS) return n_xchunks <= 0 ? n_xchunks : SOBER_E_WS;
        LX_EVENTS_BEFORE(j->ev[2], j->ev[3])
        LX_TRY(lx_reduce(j, j->idx + (lo - j->pos0), 0, n_left, SOBER_LEVEL_XS, n_xchunks, j->extraG, j->extraTot,
                         n_left, stream));
        LX_EVENTS_AFTER(j->ev[2], j->ev[3])
    }
    LX_TRY(sober_sum_partials(j->partG, j->partTot, n_chunks, j->n_rows, S, S, n_left > 0 ? j->extraG : nullptr,
                              n_left > 0 ? j->extraTot : nullptr, n_xchunks, SOBER_LEVEL_XS, j->G, S, j->tot, stream));
    if (j->phase == 1) return 0;
    return sober_dgemm(0, 0, j->n, S, j->n_rows, 1.0, j->P, j->n_rows, j->G, S, 0.0, j->Xtr, S, stream);
}

extern "C" int sober_level_car(const sober_level_job* j, void* stream) {
    if (!j || !j->Xtr || !j->tot || !j->X_tmp || !j->keep_rank || !j->w_star || !j->mu_out || !j->car_ws || !j->h_flags)
        return SOBER_E_ARG;
    const int S = j->S, n = j->n;
    if (!sober_car_supported(S, n + 1)) return SOBER_E_DIM;
    LX_TRY(sober_barycentres(j->Xtr, S, n, S, j->tot, j->X_tmp, stream));
    LX_TRY(sober_car_device_ex(j->X_tmp, n, S, n + 1, j->tot, j->keep_rank, j->w_star, j->keep_rank + S, j->mu_out,
                               nullptr, j->car_ws, j->car_ws_bytes, j->car_mode, stream));
    const hipError_t e = hipMemcpyAsync(j->h_flags, j->keep_rank, sizeof(int32_t) * (size_t)(S + 1),
                                        hipMemcpyDeviceToHost, (hipStream_t)stream);
    return e == hipSuccess ? 0 : (int)e;
}

// A level whose Caratheodory step came back with n_keep = -1 (its launches gave up waiting for partner workgroups):
// the barycentres and masses are still in place, so the step alone is redone with the launches that depend on nobody
// (SOBER_CAR_SAFE: the single-workgroup kernels at the one-CU sizes, csrc/car_big.hip's launch per dependency beyond) and the
// job stays in that mode.  -> 0 with a fresh verdict in h_flags, or
// SOBER_E_EXCHANGE when that mode does not cover the size (the caller's host route is next).
extern "C" int sober_level_car_retry(sober_level_job* j, void* stream) {
    if (!j || !j->X_tmp || !j->tot || !j->keep_rank || !j->w_star || !j->mu_out || !j->car_ws || !j->h_flags) return SOBER_E_ARG;
    const int S = j->S, n = j->n;
    if (j->car_mode == SOBER_CAR_SAFE || !sober_car_safe_supported(S, n + 1)) return SOBER_E_EXCHANGE;
    j->car_mode = SOBER_CAR_SAFE;
    LX_TRY(sober_car_device_ex(j->X_tmp, n, S, n + 1, j->tot, j->keep_rank, j->w_star, j->keep_rank + S, j->mu_out,
                               nullptr, j->car_ws, j->car_ws_bytes, SOBER_CAR_SAFE, stream));
    hipError_t e = hipMemcpyAsync(j->h_flags, j->keep_rank, sizeof(int32_t) * (size_t)(S + 1), hipMemcpyDeviceToHost,
                                  (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    return j->h_flags[S] < 0 ? SOBER_E_EXCHANGE : 0;
}

// Partial sums a queued launch of the matrix-core level kernel is sized for: an upper bound of the count the kernel
// finds from the exact number of live positions (the count itself is not monotone in that number).
static int lx_chunks(int n_rows, int64_t e_total, int S) { return sober_level_parts_mfma_cap(n_rows, e_total, S); }

// Phase A of sober_level_loop: every level that certainly exists, enqueued back to back with device-resident sizes.
// -> *done = levels completed, *R_out = live positions after them (their list: idx_a if *done is even).
// (oj != NULL: the acquisition-guided branch -- the objective as one more function of the step and the second elimination
//  behind it, SOBER/_rchq.py:138-150, :173-196 -- with every verdict read on the device; the regular outcome leaves b sets too)
static int lx_loop_queued(sober_level_job* j, const sober_obj_job* oj, int64_t R0, int32_t* idx_a, int32_t* idx_b,
                          int first_sums_ready, void** events, int max_levels, int64_t* level_R, int* done, int64_t* R_out,
                          void* stream) {
    const int S = j->S, n = j->n, b = n + 1;
    *done = 0;
    *R_out = R0;
    int64_t Rlo[SOBER_LEVEL_QUEUE + 1], Rub[SOBER_LEVEL_QUEUE + 1];
    Rlo[0] = Rub[0] = R0;
    int L = 0;
    while (L < SOBER_LEVEL_QUEUE && L < max_levels && Rlo[L] > S) {
        Rlo[L + 1] = (Rlo[L] / S) * b;                                  // no leftovers kept, n_keep = b
        // the largest E with all leftovers kept; a level whose size is known exactly has exactly R mod S leftovers
        Rub[L + 1] = (Rub[L] / S) * b + (Rlo[L] == Rub[L] ? Rub[L] % S : S - 1);
        ++L;
    }
    if (L < 2) return 0;                                                // nothing to gain: the synchronised loop
    if (!sober_car_supported(S, oj ? b + 1 : b)) return SOBER_E_DIM;
    if (oj && (!sober_null_vector_supported(n) || b + 1 > 512)) return SOBER_E_DIM;
    // levels 1 .. D gathered and scaled from level 0's class sums (the caller's phase-1 call left them: first_sums_ready)
    int D = 0;
    if (first_sums_ready && j->class_depth > 0 && !oj) {     // (the objective's row has no class sums: every level evaluated)
        D = j->class_depth;
        if ((j->variant != SOBER_LEVEL_MFMA && j->variant != SOBER_LEVEL_TANI) || D > SOBER_CLASS_MAX_DEPTH || D >= L ||
            R0 % ((int64_t)S << D) != 0 ||
            !j->Gc[0] || !j->Gc[1] || !j->totc[0] || !j->totc[1] || !j->cls_scale || !j->cls_sof)
            return SOBER_E_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    // (dR[0]: only the first level's own set sums read it -- when the caller has them already, every kernel of level 0 takes
    //  R0 by value and the launch that would put it there is not made)
    if (!first_sums_ready) LX_TRY(sober_set_i64(j->dR, R0, stream));
    hipError_t e = hipSuccess;
    for (int l = 0; l < L; ++l) {
        int32_t* cur = (l & 1) ? idx_b : idx_a;
        int32_t* nxt = (l & 1) ? idx_a : idx_b;
        for (int k = 0; k < 4; ++k) j->ev[k] = events ? events[4 * l + k] : nullptr;
        if (l >= 1 && l <= D) {
            // (level l holds 2^(D - l) classes; its predecessor's update left scale / sof and stopped the chain unless b sets survived)
            LX_TRY(sober_class_derive_queued(j->Gc[(l - 1) & 1], j->totc[(l - 1) & 1], j->n_rows, S, 1 << (D - l + 1),
                                             j->cls_scale, j->cls_sof, j->Gc[l & 1], j->totc[l & 1], j->G, j->tot, j->dR + l,
                                             stream));
        } else if (!(l == 0 && first_sums_ready)) {
            const bool tani = j->variant == SOBER_LEVEL_TANI;
            const int64_t e_ub = (Rub[l] + S - 1) / S, ex_ub = (S - 1 + SOBER_LEVEL_XS - 1) / SOBER_LEVEL_XS;
            const int nch = tani ? sober_level_chunks_tani_cap(j->n_rows, e_ub, S) : lx_chunks(j->n_rows, e_ub, S);
            const int nxch = tani ? sober_level_chunks_tani_cap(j->n_rows, ex_ub, SOBER_LEVEL_XS) : lx_chunks(j->n_rows, ex_ub, SOBER_LEVEL_XS);
            if (nch <= 0 || nch > SOBER_LEVEL_MAX_CHUNKS || nxch <= 0 || nxch > SOBER_LEVEL_MAX_CHUNKS) return SOBER_E_WS;
            LX_EVENTS_BEFORE(j->ev[0], j->ev[1])
            if (j->ev[0] && j->ev[1]) j->ev_used[0] |= 1ull << l;
            // (a level whose size is known exactly and leaves no leftover carries no leftover workgroups: the sums
            //  kernel finds the same zero from dR); otherwise both placements travel in one launch (matrix-core FP64
            //  kernel) or in two queued ones (Tanimoto)
            const bool no_left = Rlo[l] == Rub[l] && Rub[l] % S == 0;
            if (tani && !no_left && !sober::switches().level_two_launches) {
                LX_TRY(sober_level_reduce_tani_queued_pair(j->rows, j->rows_norm, j->n_rows, j->cand, j->cand_norm, j->dim, cur,
                                                           Rub[l], S, SOBER_LEVEL_XS, j->mu, j->wmul, j->outputscale, nch,
                                                           j->partG, S, j->partTot, nxch, j->extraG, j->extraTot, j->dR + l,
                                                           stream));
            } else if (tani) {
                LX_TRY(sober_level_reduce_tani_queued(j->rows, j->rows_norm, j->n_rows, j->cand, j->cand_norm, j->dim, cur,
                                                      Rub[l], S, S, 0, j->mu, j->wmul, j->outputscale, nch, j->partG, S,
                                                      j->partTot, j->dR + l, stream));
                if (!no_left) {
                    LX_EVENTS_BEFORE(j->ev[2], j->ev[3])
                    if (j->ev[2] && j->ev[3]) j->ev_used[1] |= 1ull << l;
                    LX_TRY(sober_level_reduce_tani_queued(j->rows, j->rows_norm, j->n_rows, j->cand, j->cand_norm, j->dim, cur,
                                                          S - 1, SOBER_LEVEL_XS, S, 1, j->mu, j->wmul, j->outputscale, nxch,
                                                          j->extraG, SOBER_LEVEL_XS, j->extraTot, j->dR + l, stream));
                }
            } else if (no_left || sober::switches().level_two_launches) {
                LX_TRY(sober_level_reduce_mfma_queued(j->kind, (const double*)j->rows, j->n_rows, (const double*)j->cand,
                                                      j->dim, cur, Rub[l], S, S, 0, j->mu, j->wmul, j->outputscale, nch,
                                                      j->partG, S, j->partTot, j->dR + l, stream));
                if (!no_left) {
                    LX_EVENTS_BEFORE(j->ev[2], j->ev[3])
                    if (j->ev[2] && j->ev[3]) j->ev_used[1] |= 1ull << l;
                    LX_TRY(sober_level_reduce_mfma_queued(j->kind, (const double*)j->rows, j->n_rows, (const double*)j->cand,
                                                          j->dim, cur, S - 1, SOBER_LEVEL_XS, S, 1, j->mu, j->wmul,
                                                          j->outputscale, nxch, j->extraG, SOBER_LEVEL_XS, j->extraTot,
                                                          j->dR + l, stream));
                }
            } else {
                LX_TRY(sober_level_reduce_mfma_queued_pair(j->kind, (const double*)j->rows, j->n_rows, (const double*)j->cand,
                                                           j->dim, cur, Rub[l], S, SOBER_LEVEL_XS, j->mu, j->wmul,
                                                           j->outputscale, nch, j->partG, S, j->partTot, nxch, j->extraG,
                                                           j->extraTot, j->dR + l, stream));
            }
            LX_TRY(sober_sum_partials_queued(j->partG, j->partTot, j->n_rows, S, S, j->extraG, j->extraTot,
                                             SOBER_LEVEL_XS, j->G, S, j->tot, j->dR + l, tani ? 1 : 0, stream));
        }
        // projection and barycentres in one launch (X_tmp = (P G)^T / tot: bit-identical to the two steps)
        if (oj) {
            // ... into n + 1 columns, the last one the objective's barycentres (:138-150); the step with n + 2 functions (:173);
            // the direction of the second elimination from the survivors' own matrix and the elimination (:177-196).  A first
            // step that did not leave n + 2 sets, a rank-deficient survivors' matrix or a give-up end as n_keep <= 0 in
            // keep_rank[S]: the update below then stops the chain and the caller's synchronised route takes that level
            LX_TRY(sober_dgemm_coldiv_t(n, S, j->n_rows, j->P, j->n_rows, j->G, S, j->tot, oj->X_tmp, n + 1, stream));
            LX_TRY(sober_obj_set_sums_queued(oj->obj, j->mu, cur, S, j->dR + l, (l == 0 && first_sums_ready) ? R0 : (int64_t)-1,
                                             j->tot, oj->X_tmp + n, n + 1, oj->ocol, stream));
            LX_TRY(sober_car_device_ex(oj->X_tmp, n + 1, S, n + 2, j->tot, oj->kr1, oj->w1, oj->nk1, j->mu_out, nullptr,
                                       j->car_ws, j->car_ws_bytes, j->car_mode, stream));
            LX_TRY(sober_null_vector(oj->X_tmp, n + 1, S, n, oj->kr1, oj->nk1, n + 2, oj->null_row, oj->status, stream));
            LX_TRY(sober_second_elimination_rows(oj->null_row, oj->ocol, oj->w1, oj->kr1, oj->nk1, oj->status, n + 2, S,
                                                 j->keep_rank, j->w_star, j->keep_rank + S, stream));
        } else {
            LX_TRY(sober_dgemm_coldiv_t(n, S, j->n_rows, j->P, j->n_rows, j->G, S, j->tot, j->X_tmp, n, stream));
            LX_TRY(sober_car_device_ex(j->X_tmp, n, S, n + 1, j->tot, j->keep_rank, j->w_star, j->keep_rank + S, j->mu_out,
                                       nullptr, j->car_ws, j->car_ws_bytes, j->car_mode, stream));
        }
        LX_TRY(sober_level_update_queued_ex(cur, Rub[l], S, j->keep_rank, j->w_star, j->tot, j->mu, nxt, j->dR + l,
                                            j->dR + l + 1, Rub[l + 1], l < D ? b : 0, j->cls_scale, j->cls_sof,
                                            (l == 0 && first_sums_ready) ? R0 : (int64_t)-1, stream));
    }
    for (int k = 0; k < 4; ++k) j->ev[k] = nullptr;
    e = hipMemcpyAsync(j->h_dR, j->dR, sizeof(int64_t) * (size_t)(L + 1), hipMemcpyDeviceToHost, st);
    if (e != hipSuccess) return (int)e;
    e = hipStreamSynchronize(st);                                       // the one synchronisation of the chain
    if (e != hipSuccess) return (int)e;
    j->h_dR[0] = R0;
    int d = 0;
    while (d < L && j->h_dR[d + 1] >= 0) { level_R[d] = j->h_dR[d]; ++d; }
    *done = d;
    *R_out = j->h_dR[d];
    return 0;
}

// The queued chain of the acquisition-guided branch by itself: every level that certainly exists, enqueued back to back, ONE
// synchronisation.  *n_levels levels are complete (level_R[l] = their sizes), *R_final positions are live (their list: idx_b when
// *in_b); the level the chain stopped at -- if any: an irregular outcome, a give-up, the leftover-dependent last levels -- and the
// final direct level are the caller's (its synchronised route knows the reference's answers for the irregular cases).
extern "C" int sober_level_loop_obj(sober_level_job* j, const sober_obj_job* oj, int64_t R, int32_t* idx_a, int32_t* idx_b,
                                    int first_sums_ready, int max_levels, int64_t* level_R, int32_t* n_levels, int64_t* R_final,
                                    int32_t* in_b, void* stream) {
    if (!j || !oj || !idx_a || !idx_b || !level_R || !n_levels || !R_final || !in_b || !j->mu || R <= 0 || !j->dR || !j->h_dR)
        return SOBER_E_ARG;
    if (!oj->obj || !oj->X_tmp || !oj->ocol || !oj->kr1 || !oj->w1 || !oj->nk1 || !oj->null_row || !oj->status) return SOBER_E_ARG;
    if (j->variant != SOBER_LEVEL_MFMA && j->variant != SOBER_LEVEL_TANI) return SOBER_E_ARG;
    int done = 0;
    int64_t R_after = R;
    j->ev_used[0] = j->ev_used[1] = 0;
    LX_TRY(lx_loop_queued(j, oj, R, idx_a, idx_b, first_sums_ready, nullptr, max_levels, level_R, &done, &R_after, stream));
    *n_levels = done;
    *R_final = R_after;
    *in_b = (done & 1) ? 1 : 0;
    return 0;
}

extern "C" int sober_level_loop(sober_level_job* j, int64_t R, int32_t* idx_a, int32_t* idx_b, int first_sums_ready,
                                void** events, int max_levels, int64_t* level_R, int32_t* n_levels, int64_t* R_final,
                                int32_t* in_b, void* stream) {
    if (!j || !idx_a || !idx_b || !level_R || !n_levels || !R_final || !in_b || !j->h_flags || !j->mu || R <= 0)
        return SOBER_E_ARG;
    const int S = j->S;
    int32_t *cur = idx_a, *nxt = idx_b;
    int levels = 0;
    j->ev_used[0] = j->ev_used[1] = 0;
    if (j->dR && j->h_dR && (j->variant == SOBER_LEVEL_MFMA || (j->variant == SOBER_LEVEL_TANI && !sober::switches().tani_no_queue))) {
        int done = 0;
        int64_t R_after = R;
        LX_TRY(lx_loop_queued(j, nullptr, R, idx_a, idx_b, first_sums_ready, events, max_levels, level_R, &done, &R_after, stream));
        if (done > 0) {
            levels = done;
            R = R_after;
            first_sums_ready = 0;
            if (done & 1) { cur = idx_b; nxt = idx_a; }
        }
    }
    while (R > S) {
        if (levels >= max_levels) return SOBER_E_WS;
        const int64_t E = R / S, r = R - E * S;
        j->idx = cur; j->pos0 = 0; j->count = R; j->E = E;
        j->phase = (levels == 0 && first_sums_ready) ? 2 : 0;
        for (int k = 0; k < 4; ++k) j->ev[k] = (events && levels < max_levels) ? events[4 * levels + k] : nullptr;
        if (j->phase != 2 && levels < 64) {
            if (j->ev[0] && j->ev[1]) j->ev_used[0] |= 1ull << levels;
            if (j->ev[2] && j->ev[3] && r > 0) j->ev_used[1] |= 1ull << levels;
        }
        LX_TRY(sober_level_moments(j, stream));
        LX_TRY(sober_level_car(j, stream));
        const hipError_t e = hipStreamSynchronize((hipStream_t)stream);      // the host decides the next level's size
        if (e != hipSuccess) return (int)e;
        if (j->h_flags[S] < 0) {                                            // the Caratheodory launches gave up
            const int rc = sober_level_car_retry(j, stream);
            if (rc != 0) {                                                  // (state handed back: this level is still due)
                *n_levels = levels; *R_final = R; *in_b = (cur == idx_b) ? 1 : 0;
                for (int k = 0; k < 4; ++k) j->ev[k] = nullptr;
                j->phase = 0;
                return rc;
            }
        }
        const int n_keep = j->h_flags[S];
        const bool last_kept = j->h_flags[S - 1] >= 0;
        const int64_t R_new = E * n_keep + (last_kept ? r : 0);             // :198-221
        level_R[levels++] = R;
        if (R_new >= R) { *n_levels = levels; return SOBER_E_NOPROGRESS; }
        LX_TRY(sober_level_update(cur, 0, R, S, E, j->keep_rank, j->w_star, j->tot, n_keep, j->mu, nxt, 0, stream));
        int32_t* t = cur; cur = nxt; nxt = t;
        R = R_new;
    }
    for (int k = 0; k < 4; ++k) j->ev[k] = nullptr;
    j->phase = 0;
    *n_levels = levels;
    *R_final = R;
    *in_b = (cur == idx_b) ? 1 : 0;
    return 0;
}

extern "C" int sober_level_final_job(const sober_level_job* j, const sober_final_job* f, const int32_t* idx, int R,
                                     void* stream) {
    if (!j || !f || !f->rows_sc || (!f->cand_sc && (!f->cand_raw || !f->ls || !f->sc_buf)) || !idx || !f->K || !f->mu_live ||
        !f->out_idx || !f->out_w || !j->P || !j->Xtr || !j->X_tmp || !j->keep_rank || !j->w_star || !j->mu_out || !j->car_ws ||
        !j->h_flags || !j->mu)
        return SOBER_E_ARG;
    const int S = j->S, n = j->n;
    if (R <= n + 1 || R > S || f->N <= 0) return SOBER_E_ARG;
    if (!sober_car_supported(R, n + 1)) return SOBER_E_DIM;
    if (f->cand_sc) {
        LX_TRY(sober_pairwise(j->kind, f->rows_sc, f->rows_norm, j->n_rows, f->cand_sc, f->cand_norm, idx, R, f->dt,
                              j->outputscale, f->K, R, stream));                          // kernel(pt_nys, samp[idx])  (:78)
    } else {                                                                              // (the R rows scaled here: same quotients)
        LX_TRY(sober_scale_points_idx(f->cand_raw, idx, R, f->d_raw, f->ld_raw, f->ls, f->ls_len, f->sc_buf, f->dt, j->mu,
                                      f->mu_live, stream));                               // (+ mu[idx], :84)
        LX_TRY(sober_pairwise(j->kind, f->rows_sc, f->rows_norm, j->n_rows, f->sc_buf, nullptr, nullptr, R, f->dt,
                              j->outputscale, f->K, R, stream));
    }
    LX_TRY(sober_dgemm_coldiv_t(n, R, j->n_rows, j->P, j->n_rows, f->K, R, nullptr, j->X_tmp, n, stream));   // (U K)^T, no division
    if (f->cand_sc) LX_TRY(sober_gather_f64(j->mu, idx, R, f->mu_live, stream));       // :84
    LX_TRY(sober_car_device_ex(j->X_tmp, n, R, n + 1, f->mu_live, j->keep_rank, j->w_star, j->keep_rank + S, j->mu_out,
                               nullptr, j->car_ws, j->car_ws_bytes, j->car_mode, stream));   // :85
    // mu[:] = 0 (:109) and the write-back -- both skipped on the device when the step reported no result
    LX_TRY(sober_final_commit(idx, R, j->keep_rank, j->w_star, j->keep_rank + S, f->row_offset, j->mu, f->N, f->out_idx,
                              f->out_w, stream));
    hipError_t e = hipMemcpyAsync(j->h_flags + S, j->keep_rank + S, sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream);
    return e == hipSuccess ? 0 : (int)e;
}

extern "C" int sober_level_final(const sober_level_job* j, const void* rows_sc, const double* rows_norm,
                                 const void* cand_sc, const double* cand_norm, int dt, const int32_t* idx, int R,
                                 int64_t N, int64_t row_offset, double* K, double* mu_live, int64_t* out_idx,
                                 double* out_w, void* stream) {
    if (!cand_sc) return SOBER_E_ARG;
    sober_final_job f = {};
    f.rows_sc = rows_sc; f.rows_norm = rows_norm; f.cand_sc = cand_sc; f.cand_norm = cand_norm; f.dt = dt;
    f.N = N; f.row_offset = row_offset; f.K = K; f.mu_live = mu_live; f.out_idx = out_idx; f.out_w = out_w;
    return sober_level_final_job(j, &f, idx, R, stream);
}

extern "C" int sober_final_job_size(void) { return (int)sizeof(sober_final_job); }

// The loop and the final direct level behind one call: the host language is not visited between the loop's
// synchronisation (which tells R) and the final level's launches.
extern "C" int sober_level_loop_final(sober_level_job* j, sober_final_job* f, int64_t R, int32_t* idx_a, int32_t* idx_b,
                                      int first_sums_ready, void** events, int max_levels, int64_t* level_R,
                                      int32_t* n_levels, int64_t* R_final, int32_t* in_b, void* stream) {
    if (f) f->done = 0;
    const int rc = sober_level_loop(j, R, idx_a, idx_b, first_sums_ready, events, max_levels, level_R, n_levels, R_final,
                                    in_b, stream);
    if (rc != 0 || !f) return rc;
    const int S = j->S, n = j->n;
    const int64_t Rf = *R_final;
    if (Rf <= n + 1 || Rf > S) return 0;
    if (j->car_mode != SOBER_CAR_DEFAULT && j->car_mode != SOBER_CAR_SAFE) return 0;
    if (!sober_car_supported((int)Rf, n + 1)) return 0;
    if (j->car_mode == SOBER_CAR_SAFE && !sober_car_safe_supported((int)Rf, n + 1)) return 0;
    LX_TRY(sober_level_final_job(j, f, *in_b ? idx_b : idx_a, (int)Rf, stream));
    const hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    f->done = 1;
    return 0;
}

// Two events recorded back to back on the stream: what an empty ev[0]/ev[1] bracket of sober_level_moments
// measures (the calibration of the caller's kernel timing).
extern "C" int sober_record_event_pair(void* ev0, void* ev1, void* stream) {
    if (!ev0 || !ev1) return SOBER_E_ARG;
    hipError_t e = hipEventRecord((hipEvent_t)ev0, (hipStream_t)stream);
    if (e == hipSuccess) e = hipEventRecord((hipEvent_t)ev1, (hipStream_t)stream);
    return e == hipSuccess ? 0 : (int)e;
}

// ---- the level loop of a ROW-SHARDED pool, native: no host language between the levels ------------
// Rank `rank` of `world` owns the list positions [bounds[rank], bounds[rank + 1]) (contiguous ranges, kept in closed
// form from the replicated verdicts).  Per level: the local set sums and masses (zeros for a rank without live
// positions), ONE all-reduce of the flat (n S + S) buffer on the stream (`allreduce`: sober_rccl_allreduce_f64 in
// production), the replicated Caratheodory step, one synchronisation for its verdict, the local weight update with the
// closed-form compaction.  Runs while the global R exceeds max(S, R_stop): the caller finishes the rest replicated
// after gathering the live rows once (the per-level all-reduce costs more than the sharding saves on short lists).
static int64_t lx_survivors_before(int64_t p, int S, int64_t E, const int32_t* kept_prefix, int n_keep, bool last_kept) {
    const int64_t ES = E * S;
    if (p <= ES) return (p / S) * n_keep + kept_prefix[p % S];
    return E * (int64_t)n_keep + (last_kept ? (p - ES) : 0);
}

extern "C" int sober_level_loop_sharded(sober_level_job* j, int rank, int world, int64_t* bounds, int32_t* idx_a,
                                        int32_t* idx_b, int first_sums_ready, sober_allreduce_fn allreduce,
                                        void* comm, int64_t R_stop, int max_levels, int64_t* level_R,
                                        int32_t* n_levels, int32_t* in_b, void* stream) {
    if (!j || !bounds || !idx_a || !idx_b || !allreduce || !level_R || !n_levels || !in_b || !j->h_flags || !j->mu ||
        world <= 0 || rank < 0 || rank >= world || world > 1024)
        return SOBER_E_ARG;
    const int S = j->S, n = j->n;
    if (j->Xtr + (size_t)n * S != j->tot) return SOBER_E_ARG;        // the flat all-reduce message: Xtr then tot
    if (!sober_car_supported(S, n + 1)) return SOBER_E_DIM;
    hipStream_t st = (hipStream_t)stream;
    int32_t *cur = idx_a, *nxt = idx_b;
    int levels = 0;
    int32_t kept_prefix[1024 + 1];
    if (S > 1024) return SOBER_E_DIM;
    while (bounds[world] > S && bounds[world] > R_stop) {
        if (levels >= max_levels) return SOBER_E_WS;
        const int64_t R = bounds[world], E = R / S, r = R - E * S;
        const int64_t pos0 = bounds[rank], count = bounds[rank + 1] - bounds[rank];
        if (levels == 0 && first_sums_ready) {
            // (the set sums of the first level are in G / tot already; only the projection is due)
            j->phase = 2;
            LX_TRY(sober_level_moments(j, stream));
        } else if (count > 0) {
            j->idx = cur; j->pos0 = pos0; j->count = count; j->E = E; j->phase = 0;
            LX_TRY(sober_level_moments(j, stream));
        } else {
            hipError_t e = hipMemsetAsync(j->Xtr, 0, sizeof(double) * ((size_t)n * S + S), st);
            if (e != hipSuccess) return (int)e;
        }
        if (world > 1 || comm) LX_TRY(allreduce(comm, j->Xtr, (int64_t)n * S + S, stream));
        LX_TRY(sober_level_car(j, stream));
        hipError_t e = hipStreamSynchronize(st);
        if (e != hipSuccess) return (int)e;
        if ((void*)allreduce == (void*)&sober_peer_allreduce_f64 && comm) {
            // the direct-peer all-reduce reports a rank that never arrived here, after the synchronisation (the list and
            // the weights are still those of this level)
            const int prc = sober_peer_status(comm, nullptr, 0, stream);
            if (prc != 0) { *n_levels = levels; *in_b = (cur == idx_b) ? 1 : 0; j->phase = 0; return prc; }
        }
        if (j->h_flags[S] < 0) {                                            // (a rank-local event: the redone step gives
            const int rc = sober_level_car_retry(j, stream);                    //  this rank the other ranks' verdict; beyond
            if (rc != 0) { *n_levels = levels; *in_b = (cur == idx_b) ? 1 : 0; j->phase = 0; return rc; }   //  the safe sizes the run ends here)
        }
        const int n_keep = j->h_flags[S];
        const bool last_kept = j->h_flags[S - 1] >= 0;
        const int64_t R_new = E * n_keep + (last_kept ? r : 0);
        level_R[levels++] = R;
        if (R_new >= R) { *n_levels = levels; return SOBER_E_NOPROGRESS; }
        kept_prefix[0] = 0;
        for (int s = 0; s < S; ++s) kept_prefix[s + 1] = kept_prefix[s] + (j->h_flags[s] >= 0 ? 1 : 0);
        int64_t nb_lo = lx_survivors_before(bounds[rank], S, E, kept_prefix, n_keep, last_kept);
        if (count > 0)
            LX_TRY(sober_level_update(cur, pos0, count, S, E, j->keep_rank, j->w_star, j->tot, n_keep, j->mu, nxt,
                                      nb_lo, stream));
        for (int q = 0; q <= world; ++q) bounds[q] = lx_survivors_before(bounds[q], S, E, kept_prefix, n_keep, last_kept);
        int32_t* t = cur; cur = nxt; nxt = t;
    }
    j->phase = 0;
    *n_levels = levels;
    *in_b = (cur == idx_b) ? 1 : 0;
    return 0;
}
