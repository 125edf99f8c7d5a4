// Helpers shared between the translation units of libsober_hip.so that are NOT part of the C-ABI (include/sober_hip.h):
// launches the executors (nystrom_exec.cpp) fold together.  C++ linkage, never bound by a host language.
#pragma once
#include <cstdint>

namespace sober {

// rows = [Xa; Xb] / lengthscale into one table (sober_scale_points twice, one launch)
int scale_points2(const double* Xa, int64_t na, int64_t lda, const double* Xb, int64_t nb, int64_t ldb, int d,
                  const double* lengthscale, int ls_len, double* out, int dt, void* stream);

// Ut = Q^T (s x M) and, with P != NULL, P[:, :M] = Ut diag(mean) (mean == NULL: Ut) in one launch
// (sober_barycentres(tot = NULL) followed by sober_projection's left block: same values)
int transpose_projection(const double* Q, int s, int M, const double* mean, double* Ut, double* P, int ldp, void* stream);

// the Nystrom job's flag block zeroed AND the multi-workgroup probe's exchange area / "no verdict" infos set up: one launch
// for a memset and sober_cholesky_probe_mc's own initialisation (probe_mc(..., init = false) then skips that)
int64_t probe_mc_flag_bytes(int n_shifts);
int nystrom_flags_init(void* flags_block, int64_t flags_bytes, void* probe_ws, int64_t ws_flag_bytes, int32_t* info,
                       int n_info, void* stream);
int cholesky_probe_mc(const double* src, int n, int ld_src, const double* shifts, int n_shifts, double* work, int32_t* info,
                      double* min_pivot, void* ws, int64_t ws_bytes, bool init, void* stream);

// the verdicts of two Cholesky factorisations in one slot: info <- the first failure, piv <- min, ratio <- min (ratio may be NULL)
int orth_merge(int32_t* info, double* piv, double* ratio, const int32_t* info2, const double* piv2, const double* ratio2, void* stream);

}  // namespace sober
