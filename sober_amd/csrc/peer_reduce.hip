// One-shot direct-peer all-reduce for the sharded level loop: per level the ranks exchange n S + S
// doubles (158 KB at batch 100) -- latency-bound, so every rank READS its peers' contributions straight out of their
// memory (xGMI is point to point: seven links per GPU, no switch) and sums them IN RANK ORDER: one kernel per rank, no
// ring steps, and the sum is the same bits on every rank whatever the topology (ncclAllReduce, the portable route of
// csrc/rccl_link.cpp, does not promise an order).
//
// Every rank owns an exchange region in its own memory
//     [ flag u32 @0 | arrivals u32 @64 | error u32 @128 | 256: slot 0 (n_max doubles) | slot 1 (n_max doubles) ]
// and maps the regions of its peers (hipIpc handles across processes, plain pointers inside one process).  Call number
// e ("epoch", the same on every rank: they make the same calls in the same order) of sober_peer_allreduce_f64 is ONE launch:
//   1. every workgroup copies its share of the message into slot e & 1 of MY region, makes it visible system-wide and
//      counts itself in; the workgroup that completes the count raises MY flag to e (system-scope release);
//   2. every workgroup waits until the flag of every peer has reached e (system-scope acquire, bounded spins), then
//      sums slot e & 1 of ranks 0, 1, .., W-1 -- its own region included, in that order -- into its share of the message.
// Two slots are enough: a rank can only finish call e + 1 once every peer has raised e + 1, i.e. has left call e, so
// nobody still reads slot e & 1 when call e + 2 overwrites it.  A rank that never shows up ends the wait after
// spin_limit polls (~ 30 s: an uneven shard, a collector pause or a page-in on a peer are waited for like RCCL would):
// the error word takes the number of the failing call and sober_peer_status reports SOBER_E_EXCHANGE after the next
// synchronisation -- never a hang.  That error is FATAL for the sharded run on this rank (level_exec.cpp returns it,
// _native.level_loop_sharded raises): a time-out is decided per rank, so there is no group-wide state left to continue
// from; what can be continued is decided before the first call, by the self-check at set-up (PeerComm.self_check), after
// which the group takes the RCCL route together.  sober_peer_status can hand back the rank's own message of the failing
// call (its slot is still intact) for callers that want to report or re-send it.
#include "common.hpp"
#include <cstring>

namespace sober {

constexpr int PEER_MAX_WORLD = 64;
constexpr size_t PEER_HDR = 256;
constexpr size_t PEER_OFF_ARRIVALS = 64, PEER_OFF_ERR = 128;

struct PeerComm {
    int rank = 0, world = 0;
    int64_t n_max = 0;
    unsigned char* mine = nullptr;                    // my region
    unsigned char* peer[PEER_MAX_WORLD] = {};         // everybody's region as I see it (peer[rank] = mine)
    bool opened[PEER_MAX_WORLD] = {};                 // mapped through an IPC handle (to be closed)
    unsigned char** d_peers = nullptr;                // the table above, on the device
    unsigned epoch = 0;                               // calls made so far
    unsigned arrivals = 0;                            // workgroups launched so far (mod 2^32)
    unsigned spin_limit = 1u << 27;                   // polls before giving up = the wall-time budget of a call in 0.22 us units (~ 30 s; the set-up's self-check uses short waits)
    unsigned* h_err = nullptr;                        // the error word: pinned host memory the kernel writes (no copy to read it)
    bool connected = false;
};

__global__ __launch_bounds__(256) void k_peer_allreduce(unsigned char* __restrict__ mine,
                                                        unsigned char* const* __restrict__ peers, int W,
                                                        double* __restrict__ buf, int64_t n, int64_t n_max,
                                                        unsigned epoch, unsigned arrivals_target, unsigned spin_limit,
                                                        unsigned* __restrict__ err_host) {
    const size_t slot_off = PEER_HDR + (size_t)(epoch & 1u) * (size_t)n_max * sizeof(double);
    double* my_slot = (double*)(mine + slot_off);
    const int64_t t0 = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
    // A communicator whose error word is set (an earlier call timed out) neither writes nor posts any more: the wait was what
    // kept the two-slot double buffer safe (a peer had to post epoch e + 1 before slot e & 1 was reused), so a failed rank
    // that went on copying could overwrite call e's slot with call e + 2's data under a slower, healthy peer still summing
    // call e.  That peer now meets a missing epoch instead and fails on its own bounded wait -- loudly, not with wrong sums.
    if (__hip_atomic_load((unsigned*)(mine + PEER_OFF_ERR), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) return;
    for (int64_t i = t0; i < n; i += stride) my_slot[i] = buf[i];
    __threadfence_system();                                            // my share is out before I count myself in
    __syncthreads();
    __shared__ int s_ok;
    if (threadIdx.x == 0) {
        unsigned* arr = (unsigned*)(mine + PEER_OFF_ARRIVALS);
        const unsigned before = __hip_atomic_fetch_add(arr, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (before + 1u == arrivals_target)                            // the whole message is in my slot
            __hip_atomic_store((unsigned*)mine, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        // A rank whose error word is already set (an earlier call of this communicator timed out) does not wait again:
        // every all-reduce queued behind the failed one -- one per level -- returns at once instead of spinning W x 30 s more.
        // The wait itself is bounded in WALL time (s_memrealtime, 100 MHz: spin_limit polls of ~0.22 us each, for all
        // peers together) as well as in polls, and looks at the error word every 64 polls.
        unsigned* errw = (unsigned*)(mine + PEER_OFF_ERR);
        int ok = __hip_atomic_load(errw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0u ? 1 : 0;
        const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
        const unsigned long long budget = (unsigned long long)spin_limit * 22ull;
        for (int r = 0; r < W && ok; ++r) {
            const unsigned* pf = (const unsigned*)peers[r];
            unsigned spins = 0;
            while ((int)(__hip_atomic_load(pf, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - epoch) < 0) {
                if (++spins > spin_limit) { ok = 0; break; }
                if ((spins & 63u) == 0u &&
                    (__builtin_amdgcn_s_memrealtime() - t_start > budget ||
                     __hip_atomic_load(errw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u)) { ok = 0; break; }
                __builtin_amdgcn_s_sleep(8);
            }
        }
        if (!ok) {
            const bool first = __hip_atomic_exchange(errw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0u;
            // (the FIRST failing call's number: later calls queued behind it must not move the slot sober_peer_status
            //  restores from -- they leave above without waiting, and only the workgroup that SET the error word writes
            //  the host's copy: a plain store, no PCIe atomic needed; workgroups of the same call write the same value)
            if (first || *(volatile unsigned*)err_host == 0u) {
                unsigned expect = 0u;
                if (!__hip_atomic_compare_exchange_strong(err_host, &expect, epoch | 0x80000000u, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                          __HIP_MEMORY_SCOPE_SYSTEM) && first && expect == 0u)
                    *(volatile unsigned*)err_host = epoch | 0x80000000u;
            }
        }
        s_ok = ok;
    }
    __syncthreads();
    if (!s_ok) return;                                                 // (the message stays what it was for this workgroup)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");                      // every wave sees what thread 0's acquire saw
    for (int64_t i = t0; i < n; i += stride) {
        double acc = 0.0;
        for (int r = 0; r < W; ++r) acc += ((const double*)(peers[r] + slot_off))[i];       // rank order: the same bits everywhere
        buf[i] = acc;
    }
}

}  // namespace sober

using sober::PeerComm;

extern "C" int64_t sober_peer_region_bytes(int64_t n_max) {
    if (n_max <= 0) return SOBER_E_ARG;
    return (int64_t)(sober::PEER_HDR + 2 * (size_t)n_max * sizeof(double));
}

// -> *comm, and (handle64 != NULL) the 64-byte IPC handle of my region for the other PROCESSES
extern "C" int sober_peer_create(int rank, int world, int64_t n_max, void** comm, char* handle64) {
    if (!comm || world <= 0 || world > sober::PEER_MAX_WORLD || rank < 0 || rank >= world || n_max <= 0) return SOBER_E_ARG;
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
    PeerComm* c = new PeerComm();
    c->rank = rank; c->world = world; c->n_max = n_max;
    const size_t bytes = (size_t)sober_peer_region_bytes(n_max);
    hipError_t e = hipExtMallocWithFlags((void**)&c->mine, bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) { (void)hipGetLastError(); e = hipMalloc((void**)&c->mine, bytes); }
    if (e == hipSuccess) e = hipMemset(c->mine, 0, bytes);
    if (e == hipSuccess) e = hipMalloc((void**)&c->d_peers, sizeof(unsigned char*) * (size_t)world);
    if (e == hipSuccess) e = hipHostMalloc((void**)&c->h_err, sizeof(unsigned), hipHostMallocMapped);
    if (e == hipSuccess) *c->h_err = 0u;
    if (e == hipSuccess && handle64) {
        hipIpcMemHandle_t h;
        e = hipIpcGetMemHandle(&h, c->mine);
        if (e == hipSuccess) std::memcpy(handle64, &h, 64);
    }
    if (e != hipSuccess) {
        if (c->mine) (void)hipFree(c->mine);
        if (c->d_peers) (void)hipFree(c->d_peers);
        if (c->h_err) (void)hipHostFree(c->h_err);
        delete c;
        return (int)e;
    }
    c->peer[rank] = c->mine;
    *comm = c;
    return 0;
}

static int peer_finish_connect(PeerComm* c) {
    HIP_TRY(hipMemcpy(c->d_peers, c->peer, sizeof(unsigned char*) * (size_t)c->world, hipMemcpyHostToDevice));
    c->connected = true;
    return 0;
}

// handles: world x 64 bytes (rank order; mine is ignored): the other ranks are other processes
extern "C" int sober_peer_connect(void* comm, const char* handles) {
    PeerComm* c = (PeerComm*)comm;
    if (!c || !handles) return SOBER_E_ARG;
    for (int r = 0; r < c->world; ++r) {
        if (r == c->rank) continue;
        hipIpcMemHandle_t h;
        std::memcpy(&h, handles + 64 * (size_t)r, 64);
        void* p = nullptr;
        HIP_TRY(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
        c->peer[r] = (unsigned char*)p;
        c->opened[r] = true;
    }
    return peer_finish_connect(c);
}

// regions: world device pointers (rank order): ranks that live in ONE process (several GPUs under one host thread
// each, or the single-GPU tests); peer access between the devices is the caller's to enable
extern "C" int sober_peer_connect_ptrs(void* comm, void* const* regions) {
    PeerComm* c = (PeerComm*)comm;
    if (!c || !regions) return SOBER_E_ARG;
    for (int r = 0; r < c->world; ++r)
        if (r != c->rank) {
            if (!regions[r]) return SOBER_E_ARG;
            c->peer[r] = (unsigned char*)regions[r];
        }
    return peer_finish_connect(c);
}

extern "C" int64_t sober_peer_region(void* comm) { return comm ? (int64_t)(intptr_t)((PeerComm*)comm)->mine : 0; }

extern "C" int sober_peer_set_spin_limit(void* comm, unsigned spin_limit) {
    if (!comm || spin_limit == 0) return SOBER_E_ARG;
    ((PeerComm*)comm)->spin_limit = spin_limit;
    return 0;
}

// sober_allreduce_fn: buf[0:n] <- sum over the ranks, in rank order, on the stream
extern "C" int sober_peer_allreduce_f64(void* comm, double* buf, int64_t n, void* stream) {
    PeerComm* c = (PeerComm*)comm;
    if (!c || !c->connected || !buf || n <= 0 || n > c->n_max) return SOBER_E_ARG;
    int64_t wgs = (n + 1023) / 1024;                                   // four doubles per thread; a few dozen pollers at most
    if (wgs > 64) wgs = 64;
    c->epoch += 1u;
    c->arrivals += (unsigned)wgs;
    hipLaunchKernelGGL(sober::k_peer_allreduce, dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)stream, c->mine,
                       (unsigned char* const*)c->d_peers, c->world, buf, n, c->n_max, c->epoch, c->arrivals, c->spin_limit,
                       c->h_err);
    LAUNCH_CHECK();
    return 0;
}

extern "C" int64_t sober_peer_allreduce_ptr(void) { return (int64_t)(intptr_t)&sober_peer_allreduce_f64; }

// after a synchronisation of the stream: 0 = every call so far met all its peers; SOBER_E_EXCHANGE = a wait ran out
// (the rank's own contribution of the failed call is still in its slot: *restore != NULL gets it back, n doubles);
// the error is reported once
extern "C" int sober_peer_status(void* comm, double* restore, int64_t n, void* stream) {
    PeerComm* c = (PeerComm*)comm;
    if (!c) return SOBER_E_ARG;
    const unsigned ew = *(volatile unsigned*)c->h_err;
    if (ew == 0u) return 0;
    *(volatile unsigned*)c->h_err = 0u;                                // (reported once; the flags themselves stay consistent)
    // ... and the region's own error word, which makes calls queued behind a failed one return without waiting
    HIP_TRY(hipMemsetAsync(c->mine + sober::PEER_OFF_ERR, 0, sizeof(unsigned), (hipStream_t)stream));
    if (restore && n > 0 && n <= c->n_max) {
        const double* slot = (const double*)(c->mine + sober::PEER_HDR) + (size_t)(ew & 1u) * (size_t)c->n_max;   // (the failing call's slot)
        HIP_TRY(hipMemcpyAsync(restore, slot, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    }
    return SOBER_E_EXCHANGE;
}

extern "C" int sober_peer_destroy(void* comm) {
    PeerComm* c = (PeerComm*)comm;
    if (!c) return SOBER_E_ARG;
    for (int r = 0; r < c->world; ++r)
        if (c->opened[r] && c->peer[r]) (void)hipIpcCloseMemHandle(c->peer[r]);
    if (c->d_peers) (void)hipFree(c->d_peers);
    if (c->mine) (void)hipFree(c->mine);
    if (c->h_err) (void)hipHostFree(c->h_err);
    delete c;
    return 0;
}
