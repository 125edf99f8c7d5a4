// Shared device helpers for libsober_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/sober_hip.h"

#define SOBER_MAX_DT 32          // largest padded dimension with a register-tiled instantiation
#define SOBER_WAVE 64

#define HIP_TRY(expr)                                   \
    do {                                                \
        hipError_t _e = (expr);                         \
        if (_e != hipSuccess) return (int)_e;           \
    } while (0)

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
