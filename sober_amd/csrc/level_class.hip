// Levels whose set sums are already held (round 6).
//
// Survivors of a level are compacted element-major (SOBER/_rchq.py:198-221): with n_keep = b kept sets, no leftovers and
// S = 2b, the survivor of element e in the kept set of rank k moves to list position e b + k -- element e div 2, set
// (e mod 2) b + k of the next level -- and its weight becomes mu w*_k / tot_k (:204-205).  The next level's set sums
// (:116-126) are therefore
//     G'[row, (e mod 2) b + k] = (w*_k / tot_k)  sum over the elements e of that parity of  k(row, x_{e, s_k}) mu_{e, s_k},
// sums of terms level l has already evaluated, split by the parity of e.  Carrying the split D bits deep (classes
// c = e mod 2^D at level 0) gives levels 1 .. D by a gather and a scale, no kernel evaluation: 1/2 + 1/4 + 1/8 of level
// 0's work at D = 3.  Class sums ARE set sums with 2^D S sets (position p = e S + s = (e div 2^D) 2^D S + (c S + s)), so
// level 0 runs the level kernel unchanged over S' = 2^D S sets (level_reduce_mfma.hip: sober_level_reduce_mfma_wpt);
// this file holds what follows it:
//   k_class_sum     the class launch's partial slots -> class sums Gc (n_rows x CL S), class masses totc, and their
//                   fold over the classes: the level's own G (n_rows x S) and tot -- fixed order, no atomics
//   k_class_derive  level l + 1 from level l: Gc'[row, c' S + par b + k] = scale_k Gc[row, (2 c' + par) S + s_k]
//                   (scale_k, s_k: written by the previous level's k_level_update_queued), folded into G and tot
// The reference multiplies every survivor's weight and sums afresh; here the factor multiplies the sum: the same number
// to a rounding of each (the re-association DESIGN.md section 2 already makes for the posterior correction).
#include "common.hpp"

namespace sober {

__global__ void k_class_sum(const double* __restrict__ partG, const double* __restrict__ partTot, int n_chunks, int n_rows,
                            int S, int CL, double* __restrict__ Gc, double* __restrict__ totc, double* __restrict__ G,
                            double* __restrict__ tot) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = blockIdx.y;            // row == n_rows -> the masses
    if (s >= S) return;
    const int SC = S * CL;
    double sum = 0.0;
    for (int c = 0; c < CL; ++c) {
        double acc = 0.0;
        if (row < n_rows) {
            for (int k = 0; k < n_chunks; ++k) acc += partG[((size_t)k * n_rows + row) * SC + c * S + s];
            Gc[(size_t)row * SC + c * S + s] = acc;
        } else {
            for (int k = 0; k < n_chunks; ++k) acc += partTot[(size_t)k * SC + c * S + s];
            totc[c * S + s] = acc;
        }
        sum += acc;
    }
    if (row < n_rows) G[(size_t)row * S + s] = sum;
    else tot[s] = sum;
}

__global__ void k_class_derive(const double* __restrict__ Gc, const double* __restrict__ totc, int n_rows, int S, int CL,
                               const double* __restrict__ scale, const int32_t* __restrict__ sof,
                               double* __restrict__ Gn, double* __restrict__ totn, double* __restrict__ G,
                               double* __restrict__ tot, const int64_t* __restrict__ dR) {
    const int sp = blockIdx.x * blockDim.x + threadIdx.x;      // set of the NEW level: par b + k
    const int row = blockIdx.y;
    if (sp >= S) return;
    const int64_t R = __hip_atomic_load(dR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (R <= S) return;                                         // (the chain stopped, or the loop is over: nobody reads G)
    const int b = S >> 1, par = sp >= b ? 1 : 0, k = sp - par * b;
    const int sk = sof[k];
    const double sc = scale[k];
    const int CN = CL >> 1, SC = S * CL, SN = S * CN;
    double sum = 0.0;
    for (int c = 0; c < CN; ++c) {
        double v;
        if (row < n_rows) {
            v = sc * Gc[(size_t)row * SC + (2 * c + par) * S + sk];
            if (CN > 1) Gn[(size_t)row * SN + c * S + sp] = v;
        } else {
            v = sc * totc[(2 * c + par) * S + sk];
            if (CN > 1) totn[c * S + sp] = v;
        }
        sum += v;
    }
    if (row < n_rows) G[(size_t)row * S + sp] = sum;
    else tot[sp] = sum;
}

}  // namespace sober

using namespace sober;

extern "C" int sober_class_sum(const double* partG, const double* partTot, int n_chunks, int n_rows, int S, int CL,
                               double* Gc, double* totc, double* G, double* tot, void* stream) {
    if (!partG || !partTot || !Gc || !totc || !G || !tot || n_chunks <= 0 || n_rows <= 0 || S <= 0 || CL < 2 || (CL & (CL - 1)))
        return SOBER_E_ARG;
    dim3 grid((unsigned)((S + 63) / 64), (unsigned)(n_rows + 1));
    hipLaunchKernelGGL(k_class_sum, grid, dim3(64), 0, (hipStream_t)stream, partG, partTot, n_chunks, n_rows, S, CL, Gc, totc,
                       G, tot);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int sober_class_derive_queued(const double* Gc, const double* totc, int n_rows, int S, int CL, const double* scale,
                                         const int32_t* sof, double* Gn, double* totn, double* G, double* tot,
                                         const int64_t* dR, void* stream) {
    if (!Gc || !totc || !scale || !sof || !Gn || !totn || !G || !tot || !dR || n_rows <= 0 || S <= 0 || (S & 1) || CL < 2 ||
        (CL & (CL - 1)))
        return SOBER_E_ARG;
    dim3 grid((unsigned)((S + 63) / 64), (unsigned)(n_rows + 1));
    hipLaunchKernelGGL(k_class_derive, grid, dim3(64), 0, (hipStream_t)stream, Gc, totc, n_rows, S, CL, scale, sof, Gn, totn,
                       G, tot, dR);
    LAUNCH_CHECK();
    return 0;
}
