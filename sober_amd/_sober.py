"""`Sober` with the reference's constructor and `next_batch` (SOBER/_sober.py:9-195) for the recombination hot
path: pi = `sober_amd.PI` (LFI weights over the pool), kernel = `sober_amd.Kernel`, the candidate funnel of
`sober_amd.EmpiricalSampler`, `sampling_recombination` -> HIP.  What the reference does AROUND the path stays with
the reference: GP fitting, the FBGP / BQ model families (`PI_FBGP`, `PI_BQ`), prior updates and WKDE refits
(`_prior_update.py`).  A continuous/mixed prior runs through `sampling_candidates` with the reference's control flow
(`sober_amd/_sampled_prior.py`, the default; `candidate_funnel=(sober, n_rec, n_nys, verbose) -> (X_cand, X_nys, weights)`
replaces it) and needs two hooks from the caller: `prior_updater` (`(sampler, X,
weights) -> None`, e.g. the reference's `update_prior` bound to its own prior classes) and `prior_initialiser`
(`(sampler) -> None`: what the reference's `initialise_prior`, SOBER/_sampler.py:87-111, does with its own prior
classes); WHEN the prior is reset is decided here exactly like the reference (`should_reset_prior`).  The dataset prior
(`prior.type == "dataset"`) needs none of them."""
import torch

from ._kernel import Kernel
from ._pi import PI
from ._sampler import EmpiricalSampler


class Sober(EmpiricalSampler):
    def __init__(self, prior, model, thresh=5, sampler_type="lfi", kernel_type="predictive_covariance",
                 dataset_pruning=True, prior_updater=None, prior_initialiser=None, candidate_funnel=None):
        """SOBER/_sober.py:10-39."""
        self.prior_initialiser = prior_initialiser
        self.candidate_funnel = candidate_funnel
        self.sampler_type = sampler_type
        self.kernel_type = kernel_type
        self.dataset_pruning = dataset_pruning
        self.check_model_type(model)
        pi, kernel = self.initialisation(model)
        self.n_batches_until_reset = 3
        super().__init__(prior, pi, kernel, thresh=thresh, label=prior.type, dataset_pruning=dataset_pruning,
                         prior_updater=prior_updater)

    def check_model_type(self, model):
        """SOBER/_sober.py:41-54.  The fully Bayesian and the BQ model families are outside this path."""
        if hasattr(model, "is_fbgp") or hasattr(model, "is_bq"):
            raise NotImplementedError(
                "sober_amd.Sober covers exact-GP models; FBGP / BQ models bring their own kernel callable "
                "(marginal_predictive_covariance / gspace_kernel): pass it to sober_amd.RecombinationSampler")
        self.fbgp = False
        self.is_bq = False
        targets = getattr(model, "train_targets", None)
        self.n_init = len(targets) if targets is not None else 0

    def initialisation(self, model):
        """SOBER/_sober.py:56-72."""
        return PI(model, label=self.sampler_type), Kernel(model, mode=self.kernel_type)

    def update_model(self, model):
        """SOBER/_sober.py:74-82."""
        self.pi, self.kernel = self.initialisation(model)

    def should_reset_prior(self, batch_size, recycle_prior):
        """SOBER/_sober.py:84-123: the prior is reset when the running maximum of the observations has not moved for
        `n_batches_until_reset` batches, or at every call when `recycle_prior` is False (never before the second
        batch)."""
        targets = self.pi.model.train_targets
        n_targets = len(targets)
        y_max = targets.max()
        cummax = targets.cummax(0).values
        learning_length = n_targets - self.n_init
        if (learning_length == 0) or (learning_length == batch_size):
            return False
        hit = torch.where((cummax >= y_max).diff())[0]                      # first index whose successor reaches the maximum
        idx_max = int(hit[0]) if hit.numel() else 0
        n_iterations = -(-learning_length // batch_size)                    # ceil
        for n_batches in range(1, n_iterations + 1):
            if n_batches * batch_size >= idx_max:
                break
        n_nonimproved_batches = n_iterations - n_batches + 2
        if n_nonimproved_batches >= self.n_batches_until_reset:
            return True
        return not recycle_prior

    def initialise_prior(self):
        """SOBER/_sampler.py:87-111 rebuilds the prior object of the sampler's label from the reference's own prior
        classes -- candidate generation, outside this path: the caller's `prior_initialiser(sampler)` does it."""
        if self.prior_initialiser is None:
            raise NotImplementedError(
                "the prior is due for a reset (SOBER/_sober.py:152-155: no improvement for "
                f"{self.n_batches_until_reset} batches, or recycle_prior=False) and no `prior_initialiser` was given: "
                "pass Sober(..., prior_initialiser=lambda sampler: ...) that puts a fresh prior into sampler.prior")
        self.prior_initialiser(self)

    def sampling_candidates(self, n_rec, n_nys, verbose=False):
        """SOBER/_sampler.py:264-323: the caller's `candidate_funnel` if one was given, else the reference's control
        flow (`EmpiricalSampler.sampling_candidates` -> sober_amd/_sampled_prior.py)."""
        if getattr(self, "candidate_funnel", None) is None:
            return super().sampling_candidates(n_rec, n_nys, verbose=verbose)
        return self.candidate_funnel(self, n_rec, n_nys, verbose)

    def next_batch(self, n_rec, n_nys, batch_size, calc_obj=None, return_weights=False, recycle_prior=True,
                   verbose=False):
        """SOBER/_sober.py:125-195: candidates + Nystrom sample + weights -> recombination -> one of the
        reference's three return shapes: (w_rchq, X_batch) | (idx_rchq, X_batch) for a dataset prior (indices
        into the prior's available rows when pruning is on) | X_batch."""
        if not self.label == "dataset":
            if self.should_reset_prior(batch_size, recycle_prior):           # :152-155
                print("The prior was initialised.")
                self.initialise_prior()
            X_cand, X_nys, weights = self.sampling_candidates(n_rec, n_nys, verbose=verbose)
        else:
            empirical_measure = self.sampling_datasets(n_rec, n_nys)
            if self.dataset_pruning:
                idx_sampled, X_cand, X_nys, weights = empirical_measure
            else:
                X_cand, X_nys, weights = empirical_measure
        idx_rchq, w_rchq = self.sampling_recombination(X_cand, X_nys, weights, batch_size, calc_obj=calc_obj)
        X_batch = X_cand[idx_rchq]
        if return_weights:
            return w_rchq, X_batch
        elif self.label == "dataset":
            if self.dataset_pruning:
                idx_rchq = idx_sampled[idx_rchq]
            return idx_rchq, X_batch
        return X_batch
