"""`RecombinationSampler` (SOBER/_sampler.py:11-59): the funnel from `Sober.next_batch` into
`recombination`; `adaptive_pruning` (SOBER/_sampler.py:325-349): the candidate pruning of the dataset path
(`sampling_datasets`, :351-382) that sits directly in front of it."""
import torch

from ._rchq import recombination
from ._utils import TensorManager
from ._weights import WeightsStabiliser


class RecombinationSampler(WeightsStabiliser, TensorManager):
    def __init__(self, kernel, thresh=5):
        WeightsStabiliser.__init__(self, thresh=thresh)
        TensorManager.__init__(self)
        self.kernel = kernel

    def sampling_recombination(self, X_cand, X_nys, weights, batch_size, calc_obj=None):
        """SOBER/_sampler.py:27-59 -> (idx_rchq, w_rchq)."""
        idx_rchq, w_rchq = recombination(
            X_cand, X_nys, batch_size, self.kernel, self.device, self.dtype,
            init_weights=weights, calc_obj=calc_obj)
        return idx_rchq, w_rchq

    def adaptive_pruning(self, weights, n_rec, n_nys, thresh=1e-3):
        """SOBER/_sampler.py:325-349: keep the heaviest candidates of a dataset prior -- all those with weight
        > thresh, but at most n_rec and at least n_nys; no weight above thresh -> the n_nys heaviest (the
        reference's `except` branch).  One device sort; returns the kept indices, heaviest first."""
        return adaptive_pruning(weights, n_rec, n_nys, thresh)


class EmpiricalSampler(RecombinationSampler):
    """The dataset path of SOBER/_sampler.py:61-85,351-382 (`prior.type == "dataset"`, e.g. the malaria
    fingerprints): candidates are the prior's available rows, their weights come from `pi` (sober_amd.PI for the
    LFI sampler), the heaviest are kept (`adaptive_pruning`), scrubbed (`cleansing_weights`) and a Nystrom sample is
    drawn with probability ~ 1 / weight (`deweighted_resampling`) -- everything on the device; `sampling_recombination`
    then takes (X_cand, X_nys, weights).  The sampled-prior path (`sampling_candidates`, SOBER/_sampler.py:163-323) keeps
    the reference's control flow (`sober_amd/_sampled_prior.py`): the draws and densities are the prior object's, the
    refit of the prior on the weighted sample (`_prior_update.py`, SURVEY.md section 2: out of scope) is the caller's
    `prior_updater` hook, the Nystrom subsample at its end is `nystrom_subsample` (:316-320, contract row a10)."""

    def __init__(self, prior, pi, kernel, thresh=5, label="dataset", dataset_pruning=True, prior_updater=None):
        super().__init__(kernel, thresh=thresh)
        self.thresh_initial = thresh
        self.prior = prior
        self.pi = pi
        self.label = label
        self.dataset_pruning = dataset_pruning
        self.prior_updater = prior_updater
        self.flag = False

    def nystrom_subsample(self, X_cand, weights, n_nys):
        """SOBER/_sampler.py:316-320: KMeans centroids for a continuous prior (HIP Lloyd iterations), a draw with
        probability ~ 1 / weight otherwise."""
        if self.label == "continuous":
            return self.kmeans_resampling(X_cand, n_clusters=n_nys)
        return X_cand[self.deweighted_resampling(weights, n_nys)]

    def sampling_candidates(self, n_rec, n_nys, verbose=False):
        """SOBER/_sampler.py:264-323 -> (X_cand, X_nys, weights) for a continuous / mixed / categorical prior."""
        from ._sampled_prior import sampling_candidates
        return sampling_candidates(self, n_rec, n_nys, verbose)

    def sampling_datasets(self, n_rec, n_nys):
        """SOBER/_sampler.py:351-382 -> (idx_sampled, X_cand, X_nys, weights) with pruning, else (X_cand, X_nys,
        weights)."""
        assert n_rec > n_nys
        X_cand = self.prior.available_candidates()
        weights = self.pi(X_cand)
        if self.dataset_pruning:
            idx_sampled = self.adaptive_pruning(weights, n_rec, n_nys)
            X_cand = X_cand[idx_sampled]
            weights = weights[idx_sampled]
        weights = self.cleansing_weights(weights.contiguous())
        idx_nys = self.deweighted_resampling(weights, n_nys)
        X_nys = X_cand[idx_nys]
        if self.dataset_pruning:
            return idx_sampled, X_cand, X_nys, weights
        return X_cand, X_nys, weights


def adaptive_pruning(weights, n_rec, n_nys, thresh=1e-3):
    indices = weights.argsort(descending=True)
    above = torch.where(weights[indices] > thresh)[0]
    if above.numel() == 0:                                 # IndexError in the reference -> n_nys
        n_pruned = n_nys
    else:
        n_accepted = int(above[-1]) + 1
        if n_accepted >= n_rec:
            n_pruned = n_rec
        elif n_nys >= n_accepted:
            n_pruned = n_nys
        else:
            n_pruned = n_accepted
    return indices[:n_pruned]
