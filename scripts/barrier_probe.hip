// How much does one "reduce across the workgroup" step cost as a function of workgroup size?
// Pattern per iteration (what k_car / k_chol do per serial step): wave DPP sum -> lane 0 writes LDS ->
// barrier -> every thread reads all wave partials.  Prints cycles (s_memtime, 100 MHz ticks -> ns) per step.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ double wsum(double v) {
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <int MODE>
__global__ void k(double* out, long long* ticks, int iters) {
    __shared__ double part[2][16];
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6, lane = threadIdx.x & 63;
    double x = 1.0 + threadIdx.x * 1e-6;
    long long t0 = __builtin_readcyclecounter();
    long long s0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        double s = (MODE == 0) ? x : wsum(x);
        if (MODE != 0) {
            if (lane == 0) part[it & 1][w] = s;
            __syncthreads();
            double tot = 0.0;
            for (int i = 0; i < nw; ++i) tot += part[it & 1][i];
            x = x * 0.5 + tot * 1e-9;
        } else {
            __syncthreads();
            x = x * 0.5 + s * 1e-9;
        }
    }
    long long s1 = wall_clock64();
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { ticks[0] = s1 - s0; ticks[1] = t1 - t0; }
    out[threadIdx.x] = x;
}
int main() {
    double* out; long long* ticks;
    hipMalloc(&out, 1024 * 8); hipMalloc(&ticks, 16);
    const int iters = 20000;
    int rate = 0; hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
    printf("wall clock rate kHz %d\n", rate);
    for (int mode = 0; mode < 2; ++mode)
        for (int th : {64, 128, 256, 512, 1024}) {
            long long h[2];
            for (int rep = 0; rep < 2; ++rep) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(th), 0, 0, out, ticks, iters);
                else hipLaunchKernelGGL(k<1>, dim3(1), dim3(th), 0, 0, out, ticks, iters);
                hipDeviceSynchronize();
            }
            hipMemcpy(h, ticks, 16, hipMemcpyDeviceToHost);
            printf("mode %d threads %4d: %.1f ns/step  (%.0f shader cycles/step)\n", mode, th,
                   (double)h[0] / rate * 1e6 / iters, (double)h[1] / iters);
        }
    return 0;
}
