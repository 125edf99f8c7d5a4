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

namespace sober {

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
