// Probe: cost of one all-to-all exchange of 256 doubles per workgroup among G workgroups that sit on ONE XCD
// (elected by XCC id, as csrc/car_mc.hip does), in the forms the multi-CU Caratheodory kernel could use.
//   hipcc -O3 --offload-arch=gfx950 scripts/xcd_exchange_probe.hip -o build/probes/xcd_exchange_probe   (make -f scripts/probes.mk)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
#define LDSBAR() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)

__device__ int elect(unsigned* cnt, int n, bool same_xcd) {
    unsigned* win = cnt + 8;
    const unsigned xcc = same_xcd ? (__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u) : 0u;
    if (!same_xcd) {       // spread: one worker per XCD round robin = take blocks whose id is below n
        return (int)blockIdx.x < n ? (int)blockIdx.x : -1;
    }
    const unsigned t = __hip_atomic_fetch_add(cnt + xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t >= (unsigned)n) return -1;
    if (t == (unsigned)n - 1u) { unsigned e = 0u; __hip_atomic_compare_exchange_strong(win, &e, xcc + 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    unsigned w; unsigned spins = 0;
    while ((w = __hip_atomic_load(win, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u) { if (++spins > (1u << 24)) return -1; }
    return (w == xcc + 1u) ? (int)t : -1;
}

// MODE 0: tagged 16-byte granules, every thread publishes its row and polls the G partials of its row
// MODE 1: same with waves 0-1 publishing two rows, waves 2-3 polling two rows
// MODE 2: plain 8-byte doubles + one flag per workgroup and round (store, vmcnt(0), barrier, flag; poll flags, then load)
// STORE_AUX / LOAD_AUX: 0 plain, 16 sc1
template <int MODE, int G, int STORE_AUX>
__global__ __launch_bounds__(256) void k_probe(void* comm, unsigned cbytes, unsigned* el, int rounds, int same_xcd, double* out,
                                               unsigned long long* cyc) {
    __shared__ int lcu;
    __shared__ double lsum[256];
    const int tid = threadIdx.x;
    if (tid == 0) lcu = elect(el, G, same_xcd != 0);
    __syncthreads();
    const int cu = lcu;
    if (cu < 0) return;
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(comm, 0, (int)cbytes, 0x00020000);
    double acc = 1.0 + tid * 1e-3 + cu;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rounds; ++r) {
        const unsigned tag = (unsigned)r + 1u, par = (unsigned)r & 1u;
        lsum[tid] = acc;
        LDSBAR();
        if (MODE == 0 || MODE == 1) {
            const bool pub = MODE == 0 || tid < 128, pol = MODE == 0 || tid >= 128;
            const int nrow = MODE == 0 ? 1 : 2;
            const int base = MODE == 0 ? tid : (tid & 127);
            if (pub) for (int h = 0; h < nrow; ++h) {
                const int row = base + 128 * h;
                const double v = lsum[row];
                u32x4 g; g.x = (unsigned)__double2loint(v); g.y = tag; g.z = (unsigned)__double2hiint(v); g.w = tag;
                __builtin_amdgcn_raw_buffer_store_b128(g, rs, ((par * 16u + (unsigned)cu) * 256u + (unsigned)row) * 16u, 0, STORE_AUX);
            }
            if (pol) {
                double s[2] = {0.0, 0.0};
                for (;;) {
                    u32x4 gq[2][G];
                    bool ok = true;
#pragma unroll
                    for (int h = 0; h < 2; ++h) if (h < nrow)
#pragma unroll
                        for (int g = 0; g < G; ++g)
                            gq[h][g] = __builtin_amdgcn_raw_buffer_load_b128(rs, ((par * 16u + (unsigned)g) * 256u + (unsigned)(base + 128 * h)) * 16u, 0, 16);
#pragma unroll
                    for (int h = 0; h < 2; ++h) if (h < nrow) {
                        s[h] = 0.0;
#pragma unroll
                        for (int g = 0; g < G; ++g) { ok &= (gq[h][g].y == tag) & (gq[h][g].w == tag); s[h] += __hiloint2double((int)gq[h][g].z, (int)gq[h][g].x); }
                    }
                    if (__all(ok)) break;
                    asm volatile("" ::: "memory");
                }
                for (int h = 0; h < nrow; ++h) lsum[base + 128 * h] = s[h];
            }
        } else {
            // data [par][cu][256] doubles at offset 65536; flags [par][16] dwords at offset 0
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, lsum[tid]) , rs, 65536u + ((par * 16u + (unsigned)cu) * 256u + (unsigned)tid) * 8u, 0, STORE_AUX);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            LDSBAR();
            if (tid == 0) __builtin_amdgcn_raw_buffer_store_b32(tag, rs, (par * 16u + (unsigned)cu) * 4u, 0, STORE_AUX);
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int g = 0; g < G; ++g) ok &= (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs, (par * 16u + (unsigned)g) * 4u, 0, 16) == tag;
                if (__all(ok)) break;
                asm volatile("" ::: "memory");
            }
            double s = 0.0;
            u32x2 d[G];
#pragma unroll
            for (int g = 0; g < G; ++g) d[g] = __builtin_amdgcn_raw_buffer_load_b64(rs, 65536u + ((par * 16u + (unsigned)g) * 256u + (unsigned)tid) * 8u, 0, 16);
#pragma unroll
            for (int g = 0; g < G; ++g) s += __builtin_bit_cast(double, d[g]);
            lsum[tid] = s;
        }
        LDSBAR();
        acc = lsum[tid] * 0.125 + lsum[(tid + 1) & 255] * 1e-3;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) cyc[cu] = t1 - t0;
    if (cu == 0) out[tid] = acc;
}

template <int MODE, int G, int SA>
static void run(const char* name, int same_xcd, int rounds) {
    void* comm; unsigned* el; double* out; unsigned long long* cyc;
    const unsigned cbytes = 1u << 18;
    hipMalloc(&comm, cbytes); hipMalloc(&el, 64); hipMalloc(&out, 256 * 8); hipMalloc(&cyc, 16 * 8);
    float best = 1e9f;
    unsigned long long hc[16];
    for (int rep = 0; rep < 4; ++rep) {
        hipMemset(comm, 0, cbytes); hipMemset(el, 0, 64); hipMemset(cyc, 0, 128);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_probe<MODE, G, SA>), dim3(same_xcd ? 128 : G), dim3(256), 0, 0, comm, cbytes, el, rounds, same_xcd, out, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
        hipMemcpy(hc, cyc, 128, hipMemcpyDeviceToHost);
    }
    printf("%-44s G %2d same_xcd %d: %.3f us per round (event), %.0f cycles per round (s_memtime, worker 0)\n", name, G, same_xcd,
           best * 1e3 / rounds, (double)hc[0] / rounds);
    hipFree(comm); hipFree(el); hipFree(out); hipFree(cyc);
}

int main() {
    const int R = 2000;
    run<0, 4, 0>("tagged granules, plain store", 1, R);
    run<0, 9, 0>("tagged granules, plain store", 1, R);
    run<0, 16, 0>("tagged granules, plain store", 1, R);
    run<0, 9, 16>("tagged granules, sc1 store", 1, R);
    run<0, 8, 16>("tagged granules, sc1 store", 0, R);
    run<1, 9, 0>("tagged granules, split roles, plain store", 1, R);
    run<1, 9, 16>("tagged granules, split roles, sc1 store", 1, R);
    run<2, 4, 0>("doubles + flag, plain store", 1, R);
    run<2, 9, 0>("doubles + flag, plain store", 1, R);
    run<2, 16, 0>("doubles + flag, plain store", 1, R);
    run<2, 9, 16>("doubles + flag, sc1 store", 1, R);
    run<2, 8, 16>("doubles + flag, sc1 store", 0, R);
    return 0;
}
