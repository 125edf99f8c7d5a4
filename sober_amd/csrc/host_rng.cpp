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
#include "../../include/sober_hip.h"

namespace {
constexpr int MT_N = 624, MT_M = 397;
constexpr uint32_t MATRIX_A = 0x9908b0dfu, UMASK = 0x80000000u, LMASK = 0x7fffffffu;
constexpr int64_t OFF_LEFT = 8, OFF_SEEDED = 12, OFF_NEXT = 16, OFF_STATE = 24, STATE_BYTES_MIN = 24 + 8 * MT_N;

struct Engine {
    uint32_t s[MT_N];
    int left;
    uint64_t next;
    static inline uint32_t twist(uint32_t u, uint32_t v) { return (((u & UMASK) | (v & LMASK)) >> 1) ^ ((v & 1u) ? MATRIX_A : 0u); }
    void next_state() {
        uint32_t* p = s;
        left = MT_N;
        next = 0;
        for (int j = MT_N - MT_M + 1; --j; p++) *p = p[MT_M] ^ twist(p[0], p[1]);
        for (int j = MT_M; --j; p++) *p = p[MT_M - MT_N] ^ twist(p[0], p[1]);
        *p = p[MT_M - MT_N] ^ twist(p[0], s[0]);
    }
    inline uint32_t draw() {
        if (--left == 0) next_state();
        uint32_t y = s[next++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }
};
}  // namespace

extern "C" int sober_mt19937_uniform53(uint8_t* state, int64_t state_bytes, int64_t numel, double* out) {
    if (state == nullptr || out == nullptr || state_bytes < STATE_BYTES_MIN || numel < 16) return SOBER_E_ARG;
    Engine e;
    int32_t left, seeded;
    std::memcpy(&left, state + OFF_LEFT, 4);
    std::memcpy(&seeded, state + OFF_SEEDED, 4);
    std::memcpy(&e.next, state + OFF_NEXT, 8);
    if (!seeded || left < 1 || left > MT_N || e.next > (uint64_t)MT_N) return SOBER_E_ARG;
    e.left = left;
    for (int i = 0; i < MT_N; ++i) {
        uint64_t w;
        std::memcpy(&w, state + OFF_STATE + 8 * i, 8);
        e.s[i] = (uint32_t)w;
    }
    const int64_t total = numel + ((numel % 16) ? 16 : 0);
    constexpr double DIV = 1.0 / (double)(1ull << 53);
    for (int64_t i = 0; i < total; ++i) {
        const uint64_t hi = e.draw(), lo = e.draw();
        out[i] = (double)(((hi << 32) | lo) & ((1ull << 53) - 1ull)) * DIV;
    }
    left = e.left;
    std::memcpy(state + OFF_LEFT, &left, 4);
    std::memcpy(state + OFF_NEXT, &e.next, 8);
    for (int i = 0; i < MT_N; ++i) {
        const uint64_t w = e.s[i];
        std::memcpy(state + OFF_STATE + 8 * i, &w, 8);
    }
    return 0;
}
