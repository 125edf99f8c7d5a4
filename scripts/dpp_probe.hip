// Probe: semantics of the DPP helpers used by k_car (run on the GPU box).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
template <int CTRL>
__device__ __forceinline__ double dpp(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__global__ void k(double* out) {
    int lane = threadIdx.x;
    double v = (double)lane * 1.000000001234567 + 1e-13;
    out[lane] = dpp<0x128>(v);
    out[64 + lane] = dpp<0x124>(v);
    out[128 + lane] = dpp<0x122>(v);
    out[192 + lane] = dpp<0x121>(v);
    double s = v;
    s += dpp<0x128>(s); s += dpp<0x124>(s); s += dpp<0x122>(s); s += dpp<0x121>(s);
    out[256 + lane] = s;
    // divergent rows: only rows 1 and 3 active
    double t = -1;
    if ((lane >> 4) & 1) { t = v; t += dpp<0x128>(t); t += dpp<0x124>(t); t += dpp<0x122>(t); t += dpp<0x121>(t); }
    out[320 + lane] = t;
}
int main() {
    double* d; hipMalloc(&d, 384 * 8);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    double h[384]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < 6; ++b) { printf("blk %d:", b); for (int i = 0; i < 64; ++i) printf(" %.17g", h[b * 64 + i]); printf("\n"); }
    return 0;
}
