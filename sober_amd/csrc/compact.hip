// The live list of a step: idx_story = arange(N)[mu != 0] (SOBER/_rchq.py:63-65) as an ordered stream compaction that
// leaves its count ON THE DEVICE -- torch.nonzero has to synchronise to size its output, and that synchronisation sat
// behind the whole Nystrom chain; with the count copied to pinned memory asynchronously the first level is enqueued
// right behind that chain (sober_amd/_engine.py: the list is requested before the chain is enqueued).
#include "common.hpp"
#include <cstring>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

namespace sober {
struct NonZero {
    __host__ __device__ bool operator()(double v) const { return v != 0.0; }     // (NaN counts as live, like torch)
};
}  // namespace sober

extern "C" int64_t sober_nonzero_ws_bytes(int64_t N) {
    if (N <= 0) return SOBER_E_ARG;
    size_t bytes = 0;
    rocprim::counting_iterator<int32_t> ids(0);
    auto flags = rocprim::make_transform_iterator((const double*)nullptr, sober::NonZero());
    const hipError_t e = rocprim::select(nullptr, bytes, ids, flags, (int32_t*)nullptr, (int64_t*)nullptr, (size_t)N);
    return e == hipSuccess ? (int64_t)((bytes + 255) / 256 * 256) : -(int64_t)e;
}

// idx_out[0 .. *count_out) = ascending positions of the non-zero entries of mu[0 .. N); N < 2^31
extern "C" int sober_nonzero_i32(const double* mu, int64_t N, int32_t* idx_out, int64_t* count_out, void* ws,
                                 int64_t ws_bytes, void* stream) {
    if (!mu || !idx_out || !count_out || !ws || N <= 0 || N > 0x7fffffffLL) return SOBER_E_ARG;
    size_t bytes = (size_t)ws_bytes;
    rocprim::counting_iterator<int32_t> ids(0);
    auto flags = rocprim::make_transform_iterator(mu, sober::NonZero());
    size_t need = 0;
    HIP_TRY(rocprim::select(nullptr, need, ids, flags, idx_out, count_out, (size_t)N, (hipStream_t)stream));
    if (need > bytes) return SOBER_E_WS;
    HIP_TRY(rocprim::select(ws, bytes, ids, flags, idx_out, count_out, (size_t)N, (hipStream_t)stream));
    return 0;
}
