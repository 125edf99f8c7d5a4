// popcount(x & y) of 0/1 vectors on the FP4 matrix cores of gfx950: v_mfma_scale_f32_16x16x128_f8f6f4 with both operands in
// E2M1 (1.0 = 0x2, 0.0 = 0x0), unit scales (E8M0 127) -- exact in FP32 up to 2^24.  Checks, with asymmetric random bits:
// the result against the host's popcount, the C/D map (col = lane & 15, row = 4 (lane >> 4) + reg), and that ANY common
// bit -> (lane group, nibble) map of A and B serves a dot product.  Then the instruction's rate against the INT8 form
// (v_mfma_i32_16x16x64_i8) on one wave per SIMD of every CU.
//   make -f scripts/probes.mk && gpurun -- build/probes/fp4_popcount_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned expand8_f4(unsigned x) {          // 8 bits -> 8 nibbles of 0x2 / 0x0 (bit k of the low
    const unsigned lo = ((x & 0xFu) * 0x00408102u) & 0x02020202u;     //  four -> nibble 2k, of the high four -> nibble 2k + 1)
    const unsigned hi = (((x >> 4) & 0xFu) * 0x00408102u) & 0x02020202u;
    return lo | (hi << 4);
}
__device__ __forceinline__ v4i expand32_f4(unsigned x) {
    v4i r;
    r.x = (int)expand8_f4(x & 0xFFu); r.y = (int)expand8_f4((x >> 8) & 0xFFu);
    r.z = (int)expand8_f4((x >> 16) & 0xFFu); r.w = (int)expand8_f4(x >> 24);
    return r;
}
// rows[16][4] / cols[16][4]: 128-bit vectors as four 32-bit words; lane (i = l & 15, g = l >> 4) takes word g
__global__ void k_check(const unsigned* rows, const unsigned* cols, float* out) {
    const int l = threadIdx.x, i = l & 15, g = l >> 4;
    const v4i a4 = expand32_f4(rows[i * 4 + g]), b4 = expand32_f4(cols[i * 4 + g]);
    v8i A = {a4.x, a4.y, a4.z, a4.w, 0, 0, 0, 0}, B = {b4.x, b4.y, b4.z, b4.w, 0, 0, 0, 0};
    v4f acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, acc, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    for (int v = 0; v < 4; ++v) out[(4 * g + v) * 16 + i] = acc[v];      // [row][col]
}
template <bool F4>
__global__ void k_rate(float* out, int iters) {
    v8i A = {0x22222222, 0x02020202, 0x20202020, 0x22002200, 0, 0, 0, 0}, B = A;
    v4i a4 = {0x01010101, 0x01000100, 0x00010001, 0x01010000}, b4 = a4;
    v4f acc0 = {0, 0, 0, 0}, acc1 = acc0;
    v4i c0 = {0, 0, 0, 0}, c1 = c0;
    for (int it = 0; it < iters; ++it) {
        if (F4) {
            acc0 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, acc0, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            acc1 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, acc1, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        } else {
            c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a4, b4, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a4, b4, c1, 0, 0, 0);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = F4 ? acc0[0] + acc1[1] : (float)(c0[0] + c1[1]);
}
int main() {
    std::vector<unsigned> rows(64), cols(64);
    srand(7);
    for (auto& v : rows) v = (unsigned)rand() * 2654435761u;
    for (auto& v : cols) v = ((unsigned)rand() * 40503u) ^ ((unsigned)rand() << 16);
    unsigned *dr, *dc; float* dout;
    hipMalloc(&dr, 256); hipMalloc(&dc, 256); hipMalloc(&dout, 1024 * 256 * 4 * 4);
    hipMemcpy(dr, rows.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dc, cols.data(), 256, hipMemcpyHostToDevice);
    k_check<<<1, 64>>>(dr, dc, dout);
    float out[256];
    hipMemcpy(out, dout, sizeof(out), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int r = 0; r < 16; ++r)
        for (int c = 0; c < 16; ++c) {
            int want = 0;
            for (int g = 0; g < 4; ++g) want += __builtin_popcount(rows[r * 4 + g] & cols[c * 4 + g]);
            if (out[r * 16 + c] != (float)want) { if (bad < 5) printf("mismatch r %d c %d got %g want %d\n", r, c, out[r * 16 + c], want); ++bad; }
        }
    printf("fp4 popcount check: %d mismatches of 256\n", bad);
    for (int f4 = 0; f4 < 2; ++f4) {
        const int iters = 20000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (f4) k_rate<true><<<1024, 256>>>(dout, iters); else k_rate<false><<<1024, 256>>>(dout, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double macs = (double)iters * 2 * 16 * 16 * (f4 ? 128 : 64) * 1024 * 4;
        printf("%s: %.3f ms, %.0f T bit-pair-ops/s (2 per multiply-add)\n", f4 ? "fp4 16x16x128" : "int8 16x16x64", ms, 2 * macs / (ms * 1e-3) / 1e12);
    }
    return bad != 0;
}
