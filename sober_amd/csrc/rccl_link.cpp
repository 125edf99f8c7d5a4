// RCCL for the sharded level loop, bound at run time: libsober_hip.so does not link librccl (the process already
// carries torch's copy; a second one in the same address space is asking for trouble), it dlopens the library the
// host side names -- torch's own `lib/librccl.so` -- and keeps five entry points.  No rccl.h needed: the handful of
// types involved are restated here (rccl.h: ncclUniqueId is 128 opaque bytes passed by value, ncclFloat64 = 8,
// ncclSum = 0).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstring>

#include "../../include/sober_hip.h"

namespace {
struct UniqueId { char internal[128]; };
typedef int (*fn_get_id)(UniqueId*);
typedef int (*fn_init)(void**, int, UniqueId, int);
typedef int (*fn_destroy)(void*);
typedef int (*fn_allreduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
void* g_lib = nullptr;
fn_get_id p_get_id = nullptr;
fn_init p_init = nullptr;
fn_destroy p_destroy = nullptr;
fn_allreduce p_allreduce = nullptr;
}  // namespace

extern "C" int sober_rccl_load(const char* path) {
    if (g_lib) return 0;
    void* h = dlopen(path && path[0] ? path : "librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) return SOBER_E_ARG;
    p_get_id = (fn_get_id)dlsym(h, "ncclGetUniqueId");
    p_init = (fn_init)dlsym(h, "ncclCommInitRank");
    p_destroy = (fn_destroy)dlsym(h, "ncclCommDestroy");
    p_allreduce = (fn_allreduce)dlsym(h, "ncclAllReduce");
    if (!p_get_id || !p_init || !p_destroy || !p_allreduce) { dlclose(h); return SOBER_E_ARG; }
    g_lib = h;
    return 0;
}

extern "C" int sober_rccl_unique_id(char* out128) {
    if (!g_lib || !out128) return SOBER_E_ARG;
    UniqueId id;
    const int rc = p_get_id(&id);
    if (rc != 0) return 1000 + rc;
    std::memcpy(out128, id.internal, 128);
    return 0;
}

extern "C" int sober_rccl_comm_init(const char* id128, int rank, int world, void** comm) {
    if (!g_lib || !id128 || !comm || world <= 0 || rank < 0 || rank >= world) return SOBER_E_ARG;
    UniqueId id;
    std::memcpy(id.internal, id128, 128);
    const int rc = p_init(comm, world, id, rank);
    return rc == 0 ? 0 : 1000 + rc;
}

extern "C" int sober_rccl_comm_destroy(void* comm) {
    if (!g_lib || !comm) return SOBER_E_ARG;
    const int rc = p_destroy(comm);
    return rc == 0 ? 0 : 1000 + rc;
}

// in-place sum of n doubles over the communicator, enqueued on `stream` (the sober_allreduce_fn of the level loop)
extern "C" int sober_rccl_allreduce_f64(void* comm, double* buf, int64_t n, void* stream) {
    if (!g_lib || !comm || !buf || n <= 0) return SOBER_E_ARG;
    const int rc = p_allreduce(buf, buf, (size_t)n, /*ncclFloat64*/ 8, /*ncclSum*/ 0, comm, (hipStream_t)stream);
    return rc == 0 ? 0 : 1000 + rc;
}

extern "C" int64_t sober_rccl_allreduce_ptr(void) { return (int64_t)(intptr_t)&sober_rccl_allreduce_f64; }
