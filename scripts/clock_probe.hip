// Is the shader clock what s_memtime counts?  v_add_f64 issue interval (16 independent chains, one wave per
// workgroup) measured in s_memtime ticks and in wall-clock ns, for a lightly and a fully loaded GPU and for
// short and long kernels (DPM ramp).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double* out, long long* cyc, int iters, double seed) {
    double x[16];
    for (int j = 0; j < 16; ++j) x[j] = seed + threadIdx.x * 1e-3 + j;
    const double y = 1.0 + seed * 1e-9;
    long long t0 = __builtin_amdgcn_s_memtime();
    long long w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) x[j] = x[j] + y;
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    long long w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
    double s = 0; for (int j = 0; j < 16; ++j) s += x[j];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
int main() {
    double* out; long long* cyc; hipMalloc(&out, 8 * 64 * 4096); hipMalloc(&cyc, 16);
    for (int iters : {2000, 200000, 2000000})
        for (int grid : {1, 256, 2048}) {
            for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(k, dim3(grid), dim3(64), 0, 0, out, cyc, iters, 0.5); hipDeviceSynchronize(); }
            long long h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
            printf("iters %8d grid %5d: %6.2f memtime ticks/op, %6.3f ns/op  (kernel %.3f ms)\n", iters, grid,
                   (double)h[0] / iters / 16, (double)h[1] * 10.0 / iters / 16, h[1] * 1e-5);
        }
    return 0;
}
