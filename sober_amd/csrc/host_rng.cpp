// The uniform draws behind torch.randn(M, q, dtype=float64) on the CPU generator, taken straight from the generator's
// serialised state -- the host half of the range finder's random test matrix (SOBER/_rchq.py:37 -> torch.svd_lowrank ->
// torch.randn).  torch's own randn spends ~70 % of its 0.6 ms (M = 500, q = 99) in libm's log / cos / sin, with the
// Cholesky probes of the jitter ladder waiting behind it; here the host only steps the Mersenne twister (the part
// that is sequential by construction) and the Box-Muller transform runs on the device (misc.hip, k_box_muller).
//
// What is reproduced (ATen's CPU path for a contiguous double tensor of >= 16 elements): numel uniforms
//   u = (random64() & (2^53 - 1)) * 2^-53,  random64() = (engine() << 32) | engine()
// in element order, then -- when numel is not a multiple of 16 -- 16 more for the last 16 elements, which are
// recomputed.  The engine is MT19937 with the standard tempering; its state arrives and leaves in the layout of
// torch.get_rng_state() for the CPU generator: {u64 seed; i32 left; i32 seeded; u64 next; u64 state[624]; ...}.
// sober_amd/_rng.py checks this against torch.rand / torch.get_rng_state once per process and falls back to
// torch.randn if the installed torch does anything else.
#include <cstdint>
#include <cstring>
#include <vector>
#include "../../include/sober_hip.h"

namespace {
constexpr int MT_N = 624, MT_M = 397;
constexpr uint32_t MATRIX_A = 0x9908b0dfu, UMASK = 0x80000000u, LMASK = 0x7fffffffu;
constexpr int64_t OFF_LEFT = 8, OFF_SEEDED = 12, OFF_NEXT = 16, OFF_STATE = 24, STATE_BYTES_MIN = 24 + 8 * MT_N;

static inline uint32_t twist(uint32_t u, uint32_t v) { return (((u & UMASK) | (v & LMASK)) >> 1) ^ ((v & 1u) ? MATRIX_A : 0u); }

// The generator in BLOCKS, written as plain loops the compiler vectorises (-march=x86-64-v3, like car_host.cpp): the next
// 624 words -- word i needs words i, i + 1 and i + 397 (mod 624), i.e. old values for i < 227 and values written at least
// 227 iterations earlier after that: chunks of 227 carry no dependency --, the tempering of a run of words, the pairing
// into 53-bit uniforms.  Same words, same order, same state as the word-by-word engine (2-3 x faster: the host's
// stepping of the twister is what the GPU waits for between the Cholesky probes and the range finder).
void refill(uint32_t* __restrict__ s) {
    for (int i = 0; i < MT_N - MT_M; ++i) s[i] = s[i + MT_M] ^ twist(s[i], s[i + 1]);
    for (int base = MT_N - MT_M; base < MT_N - 1; base += MT_N - MT_M) {
        const int end = base + (MT_N - MT_M) < MT_N - 1 ? base + (MT_N - MT_M) : MT_N - 1;
        for (int i = base; i < end; ++i) s[i] = s[i + MT_M - MT_N] ^ twist(s[i], s[i + 1]);
    }
    s[MT_N - 1] = s[MT_M - 1] ^ twist(s[MT_N - 1], s[0]);
}
void temper(const uint32_t* __restrict__ s, uint32_t* __restrict__ y, int n) {
    for (int i = 0; i < n; ++i) {
        uint32_t v = s[i];
        v ^= (v >> 11);
        v ^= (v << 7) & 0x9d2c5680u;
        v ^= (v << 15) & 0xefc60000u;
        v ^= (v >> 18);
        y[i] = v;
    }
}
void to_uniform53(const uint32_t* __restrict__ w, double* __restrict__ out, int64_t n) {
    constexpr double DIV = 1.0 / (double)(1ull << 53);
    for (int64_t i = 0; i < n; ++i) {
        const uint64_t v = (((uint64_t)w[2 * i] << 32) | w[2 * i + 1]) & ((1ull << 53) - 1ull);
        out[i] = (double)(int64_t)v * DIV;
    }
}
}  // namespace

extern "C" int sober_mt19937_uniform53(uint8_t* state, int64_t state_bytes, int64_t numel, double* out) {
    if (state == nullptr || out == nullptr || state_bytes < STATE_BYTES_MIN || numel < 16) return SOBER_E_ARG;
    int32_t left, seeded;
    uint64_t next;
    std::memcpy(&left, state + OFF_LEFT, 4);
    std::memcpy(&seeded, state + OFF_SEEDED, 4);
    std::memcpy(&next, state + OFF_NEXT, 8);
    if (!seeded || left < 1 || left > MT_N || next > (uint64_t)MT_N) return SOBER_E_ARG;
    uint32_t s[MT_N];
    for (int i = 0; i < MT_N; ++i) {
        uint64_t w;
        std::memcpy(&w, state + OFF_STATE + 8 * i, 8);
        s[i] = (uint32_t)w;
    }
    const int64_t total = numel + ((numel % 16) ? 16 : 0);
    // the engine's draw is { if (--left == 0) refill; word = s[next++] }: left - 1 words remain in the current block; a
    // refill is entered as left = 625, next = 0, so that taking k words is always left -= k, next += k
    static thread_local std::vector<uint32_t> words;
    words.resize((size_t)(2 * total));
    uint32_t* dst = words.data();
    for (int64_t need = 2 * total; need > 0;) {
        if (left == 1) { refill(s); left = MT_N + 1; next = 0; }
        const int k = (int)((int64_t)(left - 1) < need ? (left - 1) : need);
        temper(s + next, dst, k);
        dst += k; next += (uint64_t)k; left -= k; need -= k;
    }
    to_uniform53(words.data(), out, total);
    std::memcpy(state + OFF_LEFT, &left, 4);
    std::memcpy(state + OFF_NEXT, &next, 8);
    for (int i = 0; i < MT_N; ++i) {
        const uint64_t w = s[i];
        std::memcpy(state + OFF_STATE + 8 * i, &w, 8);
    }
    return 0;
}
