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
    then takes (X_cand, X_nys, weights).  The continuous path (`sampling_candidates`: prior updates, WKDE refits) is
    candidate GENERATION and stays with the reference."""

    def __init__(self, prior, pi, kernel, thresh=5, label="dataset", dataset_pruning=True, prior_updater=None):
        super().__init__(kernel, thresh=thresh)
        self.thresh_initial = thresh
        self.prior = prior
        self.pi = pi
        self.label = label
        self.dataset_pruning = dataset_pruning
        self.prior_updater = prior_updater
        self.flag = False

    # ---- the sampled-prior path (SOBER/_sampler.py:163-323).  CONVENIENCE OUTSIDE THE SURVEY.md 8 CONTRACT: candidate
    #      generation is not the hot path (SURVEY.md 2 marks the sampler's internals other than the funnel out of
    #      scope); these few methods only keep `Sober.next_batch` callable with a sampled prior -- control flow as in
    #      the reference, drawing from / refitting / resetting the prior delegated to the prior object and the
    #      caller's hooks.  Of this block only `nystrom_subsample` (:316-320) is on the contract (row a10).
    def check_categorical(self):
        """SOBER/_sampler.py:163-176."""
        return self.label in ("mixedcategorical", "categorical")

    def update_prior(self, X_cand, weights, verbose=False):
        """SOBER/_sampler.py:110-161 refits the prior on the weighted sample (`_prior_update.py`): outside the hot
        path -- delegated to `prior_updater(sampler, X_cand, weights)`."""
        if self.prior_updater is None:
            raise NotImplementedError("updating a sampled prior needs `prior_updater` (see sober_amd/_sober.py)")
        self.prior_updater(self, X_cand, weights)

    def sampling(self, n_rec):
        """SOBER/_sampler.py:178-192."""
        X_cand = self.prior.sample(n_rec)
        weights = self.pi(X_cand) / self.prior.pdf(X_cand)
        return X_cand, self.cleansing_weights(weights.contiguous())

    def categorical_sampling(self, n_rec):
        """SOBER/_sampler.py:194-208."""
        X_cand, X_indices = self.prior.sample_both(n_rec)
        weights = self.pi(X_cand) / self.prior.pdf(X_indices)
        return X_cand, X_indices, self.cleansing_weights(weights.contiguous())

    def recursive_sampling(self, n_rec, n_repeat=5, verbose=False):
        """SOBER/_sampler.py:210-262: repeat the weighted draw until more than `thresh` candidates carry weight;
        none at all -> uniform weights (`flag`)."""
        n_accepted, X_acc, I_acc, w_acc = 0, [], [], []
        self.flag = False
        cat = self.check_categorical()
        for _ in range(n_repeat):
            if cat:
                X_cand, X_indices, weights = self.categorical_sampling(n_rec)
            else:
                X_cand, weights = self.sampling(n_rec)
            idx = weights > 0
            if not idx.sum() == 0:
                X_acc.append(X_cand[idx])
                w_acc.append(weights[idx])
                n_accepted += int(idx.sum())
                if cat:
                    I_acc.append(X_indices[idx])
            if n_accepted > self.thresh:
                break
        if n_accepted == 0:
            self.flag = True
            if cat:
                X_cand, X_indices, weights = self.categorical_sampling(n_rec)
                return X_cand, X_indices, torch.ones(n_rec, dtype=weights.dtype, device=weights.device) / n_rec
            X_cand, weights = self.sampling(n_rec)
            return X_cand, torch.ones(n_rec, dtype=weights.dtype, device=weights.device) / n_rec
        X_cand = torch.vstack(X_acc)
        weights = self.cleansing_weights(torch.cat(w_acc).contiguous())
        if cat:
            return X_cand, torch.vstack(I_acc), weights
        return X_cand, weights

    def nystrom_subsample(self, X_cand, weights, n_nys):
        """SOBER/_sampler.py:316-320: KMeans centroids for a continuous prior (HIP Lloyd iterations), a draw with
        probability ~ 1 / weight otherwise."""
        if self.label == "continuous":
            return self.kmeans_resampling(X_cand, n_clusters=n_nys)
        return X_cand[self.deweighted_resampling(weights, n_nys)]

    def sampling_candidates(self, n_rec, n_nys, verbose=False):
        """SOBER/_sampler.py:264-323 -> (X_cand, X_nys, weights)."""
        assert n_rec > n_nys
        cat = self.check_categorical()
        if cat:
            X_cand, X_indices, weights = self.categorical_sampling(n_rec)
        else:
            X_cand, weights = self.sampling(n_rec)
        if self.check_weights(weights):
            self.update_prior(X_indices if cat else X_cand, weights, verbose=verbose)
            self.thresh = n_nys
            out = self.recursive_sampling(n_rec, n_repeat=self.thresh, verbose=verbose)
            X_cand, weights = out[0], out[-1]
        else:
            out = self.recursive_sampling(n_rec, n_repeat=self.thresh, verbose=verbose)
            X_cand, weights = out[0], out[-1]
            if self.flag:
                return X_cand, X_cand[:n_nys], weights
            self.update_prior(out[1] if cat else X_cand, weights, verbose=verbose)
            self.thresh = n_nys
            out = self.recursive_sampling(n_rec, n_repeat=self.thresh, verbose=verbose)
            X_cand, weights = out[0], out[-1]
        X_nys = self.nystrom_subsample(X_cand, weights, n_nys)
        self.thresh = self.thresh_initial
        return X_cand, X_nys, weights

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
