// Host-side pivot loop of Tchernychova_Lyons_CAR (SOBER/_rchq.py:237-266).
// Compiled without FMA contraction so every step rounds exactly like the reference's tensor
// expressions (alpha * Phi[:,0] materialised, then subtracted; outer product, then divided).
#include <stdint.h>
#include "../../include/sober_hip.h"

#pragma clang fp contract(off)

extern "C" int sober_car_pivot_host(double* h_Phi, int N, int m, double* h_mu) {
    if (!h_Phi || !h_mu || N <= 0 || m <= 0) return SOBER_E_ARG;
    int col0 = 0;          // Phi = Phi[:, 1:] is a moving first column
    int done = 0;
    for (int step = 0; step < m; ++step, ++col0) {
        // plis = Phi[:,0] > 0 ; idx = first argmin of mu/Phi[:,0] over plis  (:239-247)
        int piv = -1;
        double best = 0.0;
        for (int r = 0; r < N; ++r) {
            const double ph = h_Phi[(int64_t)r * m + col0];
            if (ph > 0.0) {
                const double a = h_mu[r] / ph;
                // torch.argmin: first minimum; a NaN counts as smaller than everything
                if (piv < 0 || (best == best && (a < best || a != a))) { piv = r; best = a; }
            }
        }
        if (piv < 0) break;                                   // Q6 (:241-242)
        const double alpha = best;
        for (int r = 0; r < N; ++r) {                         // mu[:] = mu - alpha*Phi[:,0] (:253)
            const double prod = alpha * h_Phi[(int64_t)r * m + col0];
            h_mu[r] = h_mu[r] - prod;
        }
        h_mu[piv] = 0.0;                                      // :254
        const double pp = h_Phi[(int64_t)piv * m + col0];
        const double* prow = h_Phi + (int64_t)piv * m;
        for (int r = 0; r < N; ++r) {                         // rank-1 elimination (:260-265)
            if (r == piv) continue;
            double* row = h_Phi + (int64_t)r * m;
            const double pr = row[col0];
            for (int c = col0 + 1; c < m; ++c) {
                const double prod = prow[c] * pr;             // (Phi[idx] (x) Phi_tmp).T
                const double q = prod / pp;
                row[c] = row[c] - q;
            }
        }
        double* zr = h_Phi + (int64_t)piv * m;                // Phi[idx, :] = 0 (:266)
        for (int c = col0 + 1; c < m; ++c) zr[c] = 0.0;
        ++done;
    }
    return done;
}

// Same pivot loop with the elimination factored as q_c = Phi[idx,c] / Phi[idx,0] (one division per column
// instead of one per element, as the device kernel k_car does).  Differs from the reference's tensor
// expression only in the last bits; used for the large-batch sizes (N > 200) where the N*m^2/2 divisions of
// the bit-exact form dominate a step (9 ms at batch 200).
extern "C" int sober_car_pivot_host_fast(double* h_Phi, int N, int m, double* h_mu) {
    if (!h_Phi || !h_mu || N <= 0 || m <= 0) return SOBER_E_ARG;
    int done = 0;
    double* qv = new double[m];
    for (int col0 = 0; col0 < m; ++col0) {
        int piv = -1;
        double best = 0.0;
        for (int r = 0; r < N; ++r) {
            const double ph = h_Phi[(int64_t)r * m + col0];
            if (ph > 0.0) {
                const double a = h_mu[r] / ph;
                if (piv < 0 || (best == best && (a < best || a != a))) { piv = r; best = a; }
            }
        }
        if (piv < 0) break;
        const double alpha = best;
        for (int r = 0; r < N; ++r) {
            const double prod = alpha * h_Phi[(int64_t)r * m + col0];
            h_mu[r] = h_mu[r] - prod;
        }
        h_mu[piv] = 0.0;
        const double rpp = 1.0 / h_Phi[(int64_t)piv * m + col0];
        const double* prow = h_Phi + (int64_t)piv * m;
        for (int c = col0 + 1; c < m; ++c) qv[c] = prow[c] * rpp;
        for (int r = 0; r < N; ++r) {
            if (r == piv) continue;
            double* row = h_Phi + (int64_t)r * m;
            const double pr = row[col0];
            if (pr == 0.0) continue;                          // rows cancelled earlier stay zero
            for (int c = col0 + 1; c < m; ++c) row[c] -= qv[c] * pr;
        }
        double* zr = h_Phi + (int64_t)piv * m;
        for (int c = col0 + 1; c < m; ++c) zr[c] = 0.0;
        ++done;
    }
    delete[] qv;
    return done;
}
