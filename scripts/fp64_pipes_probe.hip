// Probe: do v_mfma_f64_16x16x4 and VALU v_fma_f64 run concurrently on gfx950, and what do the
// "other" FP64 ops (rndne, ldexp, cvt, mul, add) cost?  One workgroup per CU, W waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double double4_t __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, int iters, double a, double b) {
    double4_t c0 = {0, 0, 0, 0}, c1 = {1, 1, 1, 1};
    double x0 = a, x1 = b, x2 = a + 1, x3 = b + 1, x4 = a + 2, x5 = b + 2, x6 = a + 3, x7 = b + 3;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0 || MODE == 2) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, c1, 0, 0, 0);
        }
        if (MODE == 1 || MODE == 2) {   // 32 independent-ish FMAs = 128 issue cycles ~ 2 MFMAs
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                x0 = fma(x0, a, b); x1 = fma(x1, a, b); x2 = fma(x2, a, b); x3 = fma(x3, a, b);
                x4 = fma(x4, a, b); x5 = fma(x5, a, b); x6 = fma(x6, a, b); x7 = fma(x7, a, b);
            }
        }
        if (MODE == 3) {   // rndne
#pragma unroll
            for (int u = 0; u < 4; ++u) { x0 = __builtin_rint(x0 * a); x1 = __builtin_rint(x1 * a); x2 = __builtin_rint(x2 * a); x3 = __builtin_rint(x3 * a);
                                          x4 = __builtin_rint(x4 * a); x5 = __builtin_rint(x5 * a); x6 = __builtin_rint(x6 * a); x7 = __builtin_rint(x7 * a); }
        }
        if (MODE == 4) {   // ldexp
#pragma unroll
            for (int u = 0; u < 4; ++u) { x0 = ldexp(x0, i & 1); x1 = ldexp(x1, i & 1); x2 = ldexp(x2, i & 1); x3 = ldexp(x3, i & 1);
                                          x4 = ldexp(x4, i & 1); x5 = ldexp(x5, i & 1); x6 = ldexp(x6, i & 1); x7 = ldexp(x7, i & 1); }
        }
        if (MODE == 5) {   // add
#pragma unroll
            for (int u = 0; u < 4; ++u) { x0 += a; x1 += a; x2 += a; x3 += a; x4 += a; x5 += a; x6 += a; x7 += a; }
        }
        if (MODE == 6) {   // cvt f64->i32->f64
#pragma unroll
            for (int u = 0; u < 4; ++u) { x0 = (double)(int)x0 + a; x1 = (double)(int)x1 + a; x2 = (double)(int)x2 + a; x3 = (double)(int)x3 + a;
                                          x4 = (double)(int)x4 + a; x5 = (double)(int)x5 + a; x6 = (double)(int)x6 + a; x7 = (double)(int)x7 + a; }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c0[1] + c1[2] + c1[3] + x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
template <int MODE> float run(double* d, int wg_per_cu, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * wg_per_cu), dim3(256), 0, 0, d, iters, 1.0000001, 0.5);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * wg_per_cu), dim3(256), 0, 0, d, iters, 1.0000001, 0.5);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    double* d; hipMalloc(&d, 256 * 8 * 256 * 8);
    const int iters = 20000;
    const char* names[] = {"mfma only (2/iter)", "fma only (32/iter)", "mfma + fma", "mul+rndne (32 each/iter)", "ldexp (32/iter)", "add (32/iter)", "cvt+cvt+add (32 each)"};
    for (int w = 1; w <= 2; ++w) {
        float t[7] = {run<0>(d, w, iters), run<1>(d, w, iters), run<2>(d, w, iters), run<3>(d, w, iters), run<4>(d, w, iters), run<5>(d, w, iters), run<6>(d, w, iters)};
        for (int m = 0; m < 7; ++m) {
            // per SIMD: w waves; cycles per iteration per wave-group at 2.4 GHz
            printf("waves/SIMD=%d  %-26s %.3f ms  => %.1f cycles per iter per SIMD (@2.4GHz)\n", w, names[m], t[m], t[m] * 1e-3 * 2.4e9 / iters);
        }
    }
    return 0;
}
