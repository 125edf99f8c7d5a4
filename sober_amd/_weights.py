"""`WeightsStabiliser` with the reference's interface (SOBER/_weights.py:4-126); the bulk work
(weight scrubbing, the KMeans Nystrom subsample) runs in HIP kernels."""
import warnings

import torch

from . import _native as nat


class WeightsStabiliser:
    def __init__(self, eps=torch.finfo().eps, thresh=5):
        """eps defaults to torch.finfo().eps = the FP32 epsilon even in FP64 (quirk Q5,
        SOBER/_weights.py:7)."""
        self.eps_weights = eps
        self.thresh = thresh
        # True: the random draws of weighted_resampling come from the CPU generator on a host copy of the weights,
        # i.e. the stream the reference consumes when it runs on the CPU (parity tests; one small D2H copy)
        self.reference_stream = False

    def cleansing_weights(self, weights):
        """SOBER/_weights.py:21-38, in place on the device."""
        if not weights.is_cuda:
            raise nat.SoberHipError("cleansing_weights: weights must live on the HIP device")
        if weights.dtype != torch.float64 or not weights.is_contiguous():
            w = weights.detach().to(torch.float64).contiguous()
            nat.cleansing_weights(w, self.eps_weights)
            weights.copy_(w.to(weights.dtype))
            return weights.detach()
        nat.cleansing_weights(weights, self.eps_weights)
        return weights.detach()

    def check_weights(self, weights):
        """SOBER/_weights.py:40-55: False for an all-zero vector or one with fewer than `thresh` (5) DISTINCT values."""
        # `len(weights.unique()) < thresh` sorts the whole pool to ask for five distinct values.  A head of the vector that
        # already holds `thresh` of them answers it -- and travels to the host together with the sum in ONE copy (the
        # reference's three device questions, `sum() == 0` and two `unique()`, were three synchronisations and two sorts per
        # call); only a head that does not is followed by the full answer (same verdict either way).
        if weights.is_cuda:
            head = weights[:4096]
            both = torch.cat([weights.sum().reshape(1).to(head.dtype), head]).cpu()
            if float(both[0]) == 0:
                return False
            seen = set()
            for v in both[1:].tolist():                   # (NaNs are each "distinct" for torch.unique too)
                seen.add(v if v == v else object())
                if len(seen) >= self.thresh:
                    return True
            if len(weights) > len(head) and len(weights.unique()) >= self.thresh:
                return True
            return False
        if weights.sum() == 0:
            return False
        return len(weights.unique()) >= self.thresh

    def weighted_resampling(self, weights, n_nys):
        """SOBER/_weights.py:57-77 (torch.multinomial: RNG-defined, stays in torch)."""
        n_positive_weights = (weights > 0).sum()
        if n_positive_weights > n_nys:
            if getattr(self, "reference_stream", False):
                idx_nys = torch.multinomial(weights.cpu(), n_nys).to(weights.device)
            else:
                idx_nys = torch.multinomial(weights, n_nys)
        else:
            idx_positive = torch.arange(len(weights), device=weights.device)[weights > 0]
            idx_rand = torch.randperm(len(weights))[:int(n_nys - n_positive_weights)].to(weights.device)
            idx_nys = torch.cat([idx_positive, idx_rand])
            warnings.warn("Non-zero weights are fewer than n_Nys: " + str(idx_nys.sum()))
        return idx_nys

    def deweighted_resampling(self, weights, n_samples):
        """SOBER/_weights.py:79-93."""
        weights_inv = 1 / weights
        weights_inv = self.cleansing_weights(weights_inv)
        return self.weighted_resampling(weights_inv, n_samples)

    def kmeans_resampling(self, X, n_clusters=100):
        """SOBER/_weights.py:95-97."""
        _, X_sparse = KMeans(X, n_clusters)
        return X_sparse


def KMeans(x, K=10, Niter=10):
    """Lloyd's algorithm as SOBER/_weights.py:100-126 runs it (first-K init, exactly Niter
    iterations, empty cluster -> NaN centroid) -> (labels int64, centroids)."""
    if not x.is_cuda:
        raise nat.SoberHipError("KMeans: x must live on the HIP device")
    xx = x.detach().to(torch.float64).contiguous()
    N, D = xx.shape
    c = torch.empty(K, D, dtype=torch.float64, device=x.device)
    cl = torch.empty(N, dtype=torch.int32, device=x.device)
    nat.kmeans_lloyd(xx, K, Niter, c, cl)
    return cl.long(), c.to(x.dtype)
