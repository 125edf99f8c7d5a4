// Dependent-chain latencies of the primitives k_car's serial steps are made of (one wave, gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL>
__device__ __forceinline__ double dpp(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row16_sum(double v) {
    v += dpp<0x128>(v); v += dpp<0x124>(v); v += dpp<0x122>(v); v += dpp<0x121>(v);
    return v;
}
__device__ __forceinline__ double rdlane(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double wave_sum(double v) {
    v = row16_sum(v);
    return ((rdlane(v, 0) + rdlane(v, 16)) + rdlane(v, 32)) + rdlane(v, 48);
}
template <int MODE>
__global__ void k(double* out, long long* cyc, int iters, double seed) {
    __shared__ double sh[256];
    double x = seed + threadIdx.x * 1e-3, y = 1.0 + seed;
    sh[threadIdx.x] = x;
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) x = fma(x, y, 1e-9);                               // dependent FMA
        if (MODE == 1) x = 1.0 / (x + 2.0);                               // IEEE division
        if (MODE == 2) x = sqrt(x + 2.0);                                 // IEEE sqrt
        if (MODE == 3) x = row16_sum(x) * 0.0625;                         // 16-lane DPP sum
        if (MODE == 4) x = wave_sum(x) * (1.0 / 64);                      // + readlanes
        if (MODE == 5) { sh[threadIdx.x] = x; __syncthreads(); x = sh[(threadIdx.x + 1) & 63] + 1e-9; }   // LDS round trip
        if (MODE == 6) { double b = -copysign(sqrt(fma(x, x, y)), x); double tau = (b - x) / b; double sc = 1.0 / (x - b); x = tau + sc * 1e-3 + 1.0; }  // larfg
        if (MODE == 7) x = __shfl(x, (threadIdx.x + 1) & 63, 64) + 1e-9; // ds_bpermute
        if (MODE == 8) { double r = __builtin_amdgcn_rcp(x + 2.0); x = r; } // raw v_rcp_f64
        if (MODE == 9) { x = x * y + 1e-9; __builtin_amdgcn_s_barrier(); } // barrier alone (1 wave)
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
    out[threadIdx.x] = x;
}
int main() {
    double* out; long long* cyc; hipMalloc(&out, 2048); hipMalloc(&cyc, 8);
    const int iters = 4000;
    const char* names[] = {"dependent v_fma_f64", "IEEE f64 division", "IEEE f64 sqrt", "row16_sum (4 DPP stages)", "wave_sum (DPP + 4 readlane)",
                           "LDS write+barrier+read (1 wave)", "larfg (sqrt + 2 div)", "__shfl (ds_bpermute)", "v_rcp_f64", "fma + s_barrier"};
#define RUN(M) { for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(k<M>, dim3(1), dim3(64), 0, 0, out, cyc, iters, 0.5); hipDeviceSynchronize(); } \
    long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("%-36s %7.1f cycles\n", names[M], (double)h / iters); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9)
    return 0;
}
