// s_barrier release latency seen by the LAST arriver: 15 (or 7, 3) waves wait, one wave arrives after a delay.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double* out, long long* cyc, int iters, int late_wave, int delay) {
    __shared__ double sh[64];
    const int wave = threadIdx.x >> 6;
    double x = 1.0 + threadIdx.x * 1e-6;
    long long acc = 0;
    for (int it = 0; it < iters; ++it) {
        if (wave == late_wave) { for (int d = 0; d < delay; ++d) x = fma(x, 1.0000001, 1e-9); }
        if (threadIdx.x == late_wave * 64) sh[0] = x;                 // an LDS write to publish, like pscal
        long long t0 = __builtin_amdgcn_s_memtime();
        __syncthreads();
        long long t1 = __builtin_amdgcn_s_memtime();
        if (wave == late_wave) acc += t1 - t0;
        x += sh[0] * 1e-9;
    }
    if (threadIdx.x == late_wave * 64) cyc[0] = acc;
    out[threadIdx.x] = x;
}
int main() {
    double* out; long long* cyc; hipMalloc(&out, 1024 * 8); hipMalloc(&cyc, 8);
    const int iters = 2000;
    for (int th : {256, 512, 1024})
        for (int delay : {0, 50, 200}) {
            for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(k, dim3(1), dim3(th), 0, 0, out, cyc, iters, 1, delay); hipDeviceSynchronize(); }
            long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
            printf("threads %4d, late wave delayed by %3d dependent FMAs: barrier costs the late wave %.0f ticks\n", th, delay, (double)h / iters);
        }
    return 0;
}
