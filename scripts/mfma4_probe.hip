// Lane layout of v_mfma_f64_4x4x4_4b_f64 on gfx950, found by experiment: a = [lane == la], b = [lane == lb], c = 0 for all
// 64 x 64 pairs; prints which output lanes are non-zero, then checks the two-instruction 16-lane all-reduce built on it.
//   make -f scripts/probes.mk build/probes/mfma4_probe && build/probes/mfma4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
__global__ void k_map(unsigned long long* out) {
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
            const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            const unsigned long long m = __ballot(d != 0.0);
            if (lane == 0) out[la * 64 + lb] = m;
        }
}
// all-reduce over each 16-lane row: stage 1 sums over one index, stage 2 over the other
__global__ void k_sum(const double* x, double* out) {
    const int lane = threadIdx.x;
    const double v = x[lane];
    const double s1 = __builtin_amdgcn_mfma_f64_4x4x4f64(v, 1.0, 0.0, 0, 0, 0);
    const double s2 = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, s1, 0.0, 0, 0, 0);
    out[lane] = s2;
    out[64 + lane] = s1;
}
int main() {
    unsigned long long* d; hipMalloc(&d, 4096 * 8);
    hipLaunchKernelGGL(k_map, dim3(1), dim3(64), 0, 0, d);
    static unsigned long long h[4096];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    // decode: for A lane la and B lane lb that meet, the output lanes
    int a_blk[64], a_i[64], a_k[64];
    for (int la = 0; la < 64; ++la) {
        printf("A lane %2d meets B lanes:", la);
        for (int lb = 0; lb < 64; ++lb) if (h[la * 64 + lb]) printf(" %d->out%llx", lb, h[la * 64 + lb]);
        printf("\n");
        if (la == 7) { printf("...\n"); }
        if (la >= 7 && la < 60) { /* keep the print short */ }
    }
    double hx[64], *dx, *dout, ho[128];
    for (int i = 0; i < 64; ++i) hx[i] = std::ldexp(1.0, i % 16) + i / 16 * 1e-3;
    hipMalloc(&dx, 512); hipMalloc(&dout, 1024);
    hipMemcpy(dx, hx, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_sum, dim3(1), dim3(64), 0, 0, dx, dout);
    hipMemcpy(ho, dout, 1024, hipMemcpyDeviceToHost);
    for (int r = 0; r < 4; ++r) {
        double want = 0; for (int i = 0; i < 16; ++i) want += hx[16 * r + i];
        printf("row %d want %.6f:", r, want);
        for (int i = 0; i < 16; ++i) printf(" %.3f", ho[16 * r + i]);
        printf("\n   stage 1:");
        for (int i = 0; i < 16; ++i) printf(" %.3f", ho[64 + 16 * r + i]);
        printf("\n");
    }
    return 0;
}
