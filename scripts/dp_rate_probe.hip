// What the FP64 units actually sustain chip-wide (wall clock), for the level kernel's roofline:
//   fma   : v_fma_f64 only, 16 independent chains per lane
//   mfma  : v_mfma_f64_16x16x4 only, 4 independent accumulator tiles
//   mix   : 12 MFMAs + 160 FMAs per iteration (the level kernel's DP mix at d = 10), same wave
//   int   : mix + 104 integer VALU ops per iteration (its non-DP vector instructions)
// at 1, 2 and 3 waves per SIMD on all 256 CUs.  Build: hipcc --offload-arch=gfx950 -O3 -o build/probes/dp_rate_probe scripts/dp_rate_probe.hip   (make -f scripts/probes.mk)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, int iters, double seed) {
    double x[16];
    for (int j = 0; j < 16; ++j) x[j] = seed + threadIdx.x * 1e-3 + j;
    const double y = 1.0 + seed * 1e-9, z = seed * 1e-12;
    d4 c[4];
    for (int t = 0; t < 4; ++t) c[t] = (d4){0, 0, 0, 0};
    int n[8];
    for (int j = 0; j < 8; ++j) n[j] = threadIdx.x + j;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 2 || MODE == 3) {
#pragma unroll
            for (int r = 0; r < 10; ++r)
#pragma unroll
                for (int j = 0; j < 16; ++j) x[j] = __builtin_fma(x[j], y, z);
        }
        if (MODE >= 1 && MODE <= 3) {
#pragma unroll
            for (int ks = 0; ks < 3; ++ks)
#pragma unroll
                for (int t = 0; t < 4; ++t) c[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[t], y, c[t], 0, 0, 0);
        }
        if (MODE == 4 || MODE == 5) {          // 12 x (MFMA, then 8 int / 8 DP vector instructions), order pinned
#pragma unroll
            for (int ks = 0; ks < 3; ++ks)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    c[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[t], y, c[t], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (MODE == 4) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) n[j] = __builtin_amdgcn_alignbyte(n[j], n[(j + 1) & 7], 1);
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) x[8 + j] = __builtin_fma(x[8 + j], y, z);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
        if (MODE == 6 || MODE == 7) {          // the same instructions, MFMAs first
#pragma unroll
            for (int ks = 0; ks < 3; ++ks)
#pragma unroll
                for (int t = 0; t < 4; ++t) c[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[t], y, c[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 12; ++r) {
                if (MODE == 6) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) n[j] = __builtin_amdgcn_alignbyte(n[j], n[(j + 1) & 7], 1);
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[8 + j] = __builtin_fma(x[8 + j], y, z);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 3) {
#pragma unroll
            for (int r = 0; r < 13; ++r)
#pragma unroll
                for (int j = 0; j < 8; ++j) n[j] = (n[j] << 1) ^ (n[j] + r);
        }
    }
    double s = 0;
    for (int j = 0; j < 16; ++j) s += x[j];
    for (int t = 0; t < 4; ++t) s += c[t][0] + c[t][1] + c[t][2] + c[t][3];
    for (int j = 0; j < 8; ++j) s += n[j];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
static void run(const char* name, double flop_per_iter_per_wave) {
    double* out; hipMalloc(&out, sizeof(double) * 256 * 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    for (int wps : {1, 2, 3}) {
        const int grid = 256 * wps;
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, iters, 0.5);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, iters, 0.5);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double waves = grid * 4.0;
        printf("%-5s %d waves/SIMD: %7.3f ms  %6.1f TFLOP/s  (%.0f ns per iteration per SIMD)\n", name, wps, ms,
               waves * iters * flop_per_iter_per_wave / ms / 1e9, ms * 1e6 / iters / wps);
    }
    hipFree(out);
}
int main() {
    run<0>("fma", 160.0 * 64 * 2);
    run<1>("mfma", 12.0 * 16 * 16 * 4 * 2);
    run<2>("mix", 160.0 * 64 * 2 + 12.0 * 16 * 16 * 4 * 2);
    run<3>("int", 160.0 * 64 * 2 + 12.0 * 16 * 16 * 4 * 2);
    run<4>("m/i", 12.0 * 16 * 16 * 4 * 2);                       // 12 MFMA interleaved with 96 int instructions
    run<6>("m+i", 12.0 * 16 * 16 * 4 * 2);                       // ... MFMAs first, then the 96
    run<5>("m/f", 96.0 * 64 * 2 + 12.0 * 16 * 16 * 4 * 2);       // 12 MFMA interleaved with 96 FMAs
    run<7>("m+f", 96.0 * 64 * 2 + 12.0 * 16 * 16 * 4 * 2);
    return 0;
}
