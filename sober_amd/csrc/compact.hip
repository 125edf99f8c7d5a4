// The live list of a step: idx_story = arange(N)[mu != 0] (SOBER/_rchq.py:63-65) as an ordered stream compaction that
// leaves its count ON THE DEVICE -- torch.nonzero has to synchronise to size its output, and that synchronisation sat
// behind the whole Nystrom chain; with the count copied to pinned memory asynchronously the first level is enqueued
// right behind that chain (sober_amd/_engine.py: the list is requested before the chain is enqueued).
//
// ONE launch (round 6; rocprim::select was a fill and three kernels, ~25 us of a serial stream): a workgroup takes the next
// tile of 2048 weights in TICKET order, counts its non-zeros by wave ballots, publishes the count, finds the number of live
// positions in front of it by a decoupled look-back over its predecessors' status words (Merrill & Garland 2016: every
// predecessor holds an earlier ticket, so it is resident and publishes without waiting for anybody -- no deadlock whatever
// the dispatch order), and writes its positions in ascending order.  The workgroup that finishes last zeroes the status
// words again: the workspace must be zero before its FIRST use and every call leaves it so (include/sober_hip.h).
#include "common.hpp"

namespace sober {

constexpr int NZ_T = 256, NZ_PER = 8, NZ_TILE = NZ_T * NZ_PER;
constexpr unsigned long long NZ_LOCAL = 1ull << 32, NZ_INCL = 2ull << 32;     // status = flag << 32 | count (< 2^31)

__device__ __forceinline__ int nz_wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ws: [0] ticket, [1] finished workgroups, [2 + tile] status
__global__ __launch_bounds__(NZ_T) void k_nonzero(const double* __restrict__ mu, int64_t N, int32_t* __restrict__ out,
                                                  int64_t* __restrict__ count_out, unsigned long long* __restrict__ ws, int nb) {
    __shared__ int s_tile, s_total, s_prefix, s_last;
    __shared__ int s_cnt[NZ_PER * 4];
    if (threadIdx.x == 0) s_tile = (int)atomicAdd((unsigned*)ws, 1u);
    __syncthreads();
    const int tile = s_tile;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t base = (int64_t)tile * NZ_TILE;
    unsigned long long m[NZ_PER];
#pragma unroll
    for (int k = 0; k < NZ_PER; ++k) {
        const int64_t i = base + k * NZ_T + threadIdx.x;
        const bool nz = i < N && mu[i] != 0.0;                                  // (NaN counts as live, like torch)
        m[k] = __ballot(nz);
        if (lane == 0) s_cnt[k * 4 + w] = __popcll(m[k]);
    }
    __syncthreads();
    unsigned long long* status = ws + 2;
    if (w == 0) {
        const int v = lane < NZ_PER * 4 ? s_cnt[lane] : 0;
        int incl = v;
#pragma unroll
        for (int o = 1; o < NZ_PER * 4; o <<= 1) {
            const int u = __shfl_up(incl, o);
            if (lane >= o) incl += u;
        }
        if (lane < NZ_PER * 4) s_cnt[lane] = incl - v;
        const int total = __shfl(incl, NZ_PER * 4 - 1);
        int run = 0;
        if (tile == 0) {
            if (lane == 0) __hip_atomic_store(status, NZ_INCL | (unsigned long long)total, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (lane == 0)
                __hip_atomic_store(status + tile, NZ_LOCAL | (unsigned long long)total, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            for (int p = tile - 1; p >= 0; p -= 64) {
                const int q = p - lane;
                unsigned long long st = 0;
                if (q >= 0) {
                    do st = __hip_atomic_load(status + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); while ((st >> 32) == 0);
                }
                const unsigned long long bi = __ballot(q >= 0 && (st >> 32) == 2);
                if (bi != 0) {                                                   // the nearest predecessor that knows its prefix
                    const int first = __ffsll((long long)bi) - 1;
                    run += nz_wave_sum(lane <= first ? (int)(unsigned)st : 0);
                    break;
                }
                run += nz_wave_sum(q >= 0 ? (int)(unsigned)st : 0);
            }
            if (lane == 0)
                __hip_atomic_store(status + tile, NZ_INCL | (unsigned long long)(run + total), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) { s_prefix = run; s_total = total; }
    }
    __syncthreads();
    const int prefix = s_prefix;
#pragma unroll
    for (int k = 0; k < NZ_PER; ++k) {
        if ((m[k] >> lane) & 1ull)
            out[prefix + s_cnt[k * 4 + w] + __popcll(m[k] & ((1ull << lane) - 1ull))] = (int32_t)(base + k * NZ_T + threadIdx.x);
    }
    if (threadIdx.x == 0) {
        if (tile == nb - 1) *count_out = (int64_t)prefix + s_total;
        // (every status word this workgroup needed has been read: the last one to say so puts the workspace back)
        s_last = atomicAdd((unsigned*)(ws + 1), 1u) == (unsigned)(nb - 1);
    }
    __syncthreads();
    if (s_last)
        for (int t = threadIdx.x; t < nb + 2; t += NZ_T) ws[t] = 0ull;
}

}  // namespace sober

extern "C" int64_t sober_nonzero_ws_bytes(int64_t N) {
    if (N <= 0 || N > 0x7fffffffLL) return SOBER_E_ARG;
    const int64_t nb = (N + sober::NZ_TILE - 1) / sober::NZ_TILE;
    return ((2 + nb) * 8 + 255) / 256 * 256;
}

// idx_out[0 .. *count_out) = ascending positions of the non-zero entries of mu[0 .. N); N < 2^31.  ws: zero before its first
// use (the call leaves it zero); one call at a time per workspace.
extern "C" int sober_nonzero_i32(const double* mu, int64_t N, int32_t* idx_out, int64_t* count_out, void* ws,
                                 int64_t ws_bytes, void* stream) {
    if (!mu || !idx_out || !count_out || !ws || N <= 0 || N > 0x7fffffffLL) return SOBER_E_ARG;
    if (ws_bytes < sober_nonzero_ws_bytes(N)) return SOBER_E_WS;
    const int nb = (int)((N + sober::NZ_TILE - 1) / sober::NZ_TILE);
    hipLaunchKernelGGL(sober::k_nonzero, dim3((unsigned)nb), dim3(sober::NZ_T), 0, (hipStream_t)stream, mu, N, idx_out, count_out,
                       (unsigned long long*)ws, nb);
    LAUNCH_CHECK();
    return 0;
}
