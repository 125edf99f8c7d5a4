// Issue rates (cycles per wave instruction, one wave on a SIMD) of the instruction kinds k_car is made of.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(double* out, long long* cyc, int iters, double seed) {
    __shared__ double sh[1024];
    double x[16];
    for (int j = 0; j < 16; ++j) x[j] = seed + threadIdx.x * 1e-3 + j;
    const double y = 1.0 + seed * 1e-9;
    for (int j = threadIdx.x; j < 1024; j += 64) sh[j] = j;
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (MODE == 0) x[j] = fma(x[j], y, 1e-9);                       // 16 independent DP FMA
            if (MODE == 1) x[j] = x[j] * y;                                  // DP mul
            if (MODE == 2) x[j] = x[j] + y;                                  // DP add
            if (MODE == 3) x[j] = (x[j] > 3.0) ? x[(j + 1) & 15] : x[j];     // cmp + 2 cndmask
            if (MODE == 4) x[j] += sh[(threadIdx.x + 16 * j + it) & 1023];   // LDS read b64 + add
            if (MODE == 5) { int lo = __double2loint(x[j]); lo = __builtin_amdgcn_update_dpp(0, lo, 0x121, 0xf, 0xf, false); x[j] = __hiloint2double(__double2hiint(x[j]), lo); }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
    double s = 0; for (int j = 0; j < 16; ++j) s += x[j];
    out[threadIdx.x] = s;
}
int main() {
    double* out; long long* cyc; hipMalloc(&out, 2048); hipMalloc(&cyc, 8);
    const int iters = 2000;
    const char* names[] = {"v_fma_f64 (16 independent)", "v_mul_f64", "v_add_f64", "v_cmp + 2 v_cndmask", "ds_read_b64 + v_add_f64", "v_mov_b32_dpp"};
#define RUN(M) { for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(k<M>, dim3(1), dim3(64), 0, 0, out, cyc, iters, 0.5); hipDeviceSynchronize(); } \
    long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("%-32s %6.2f cycles per op-group\n", names[M], (double)h / iters / 16); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5)
    return 0;
}
