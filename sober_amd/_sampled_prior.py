"""The sampled-prior (continuous / mixed / categorical) candidate funnel in front of `Sober.next_batch`
(SOBER/_sampler.py:163-323): the DEFAULT `candidate_funnel` of `sober_amd.Sober`, and what
`EmpiricalSampler.sampling_candidates` delegates to -- so `next_batch` works out of the box for these priors like the
reference's.

The control flow is the reference's `sampling_candidates` (same order of draws, same thresholds); drawing and densities
are the prior object's (`prior.sample`, `prior.sample_both`, `prior.pdf`), refitting the prior on the weighted sample
(`_prior_update.py`: WKDE refits etc., SURVEY.md section 2: out of scope) is the caller's `prior_updater` hook; the Nystrom
subsample at its end is the package's `nystrom_subsample` (KMeans on the device for a continuous prior).  Any other

    candidate_funnel(sober, n_rec, n_nys, verbose) -> (X_cand, X_nys, weights)

can be given to `sober_amd.Sober(prior, model, candidate_funnel=...)` instead.  tests/ drive this one against the
reference's own `Sober.next_batch` fixture (tests/test_hip_parity.py::test_sober_next_batch_vs_reference)."""
import torch


def _categorical(sober):
    return sober.label in ("mixedcategorical", "categorical")            # SOBER/_sampler.py:163-176


def _update_prior(sober, X, weights):
    """SOBER/_sampler.py:110-161 refits the prior on the weighted sample (`_prior_update.py`): the caller's hook."""
    if sober.prior_updater is None:
        raise NotImplementedError("updating a sampled prior needs Sober(..., prior_updater=callable(sampler, X, weights))")
    sober.prior_updater(sober, X, weights)


def _draw(sober, n_rec):
    """SOBER/_sampler.py:178-208 -> (X_cand, X_indices or None, weights)."""
    if _categorical(sober):
        X_cand, X_indices = sober.prior.sample_both(n_rec)
        weights = sober.pi(X_cand) / sober.prior.pdf(X_indices)
        return X_cand, X_indices, sober.cleansing_weights(weights.contiguous())
    X_cand = sober.prior.sample(n_rec)
    weights = sober.pi(X_cand) / sober.prior.pdf(X_cand)
    return X_cand, None, sober.cleansing_weights(weights.contiguous())


def _recursive_draws(sober, n_rec, n_repeat):
    """SOBER/_sampler.py:210-262: repeat the weighted draw until more than `thresh` candidates carry weight; none at
    all -> uniform weights (`sober.flag`)."""
    n_accepted, X_acc, I_acc, w_acc = 0, [], [], []
    sober.flag = False
    for _ in range(n_repeat):
        X_cand, X_indices, weights = _draw(sober, n_rec)
        # (the reference's `idx = weights > 0; X_cand[idx]; weights[idx]; X_indices[idx]` is a mask count and three masked
        #  gathers, each with its own synchronisation on a device: ONE list of the positive positions serves them all, and a
        #  draw whose weights are all positive is taken as it is)
        pos = torch.nonzero(weights > 0).squeeze(1)
        n_pos = int(pos.numel())
        if n_pos != 0:
            whole = n_pos == weights.numel()
            X_acc.append(X_cand if whole else X_cand.index_select(0, pos))
            w_acc.append(weights if whole else weights.index_select(0, pos))
            n_accepted += n_pos
            if X_indices is not None:
                I_acc.append(X_indices if whole else X_indices.index_select(0, pos))
        if n_accepted > sober.thresh:
            break
    if n_accepted == 0:
        sober.flag = True
        X_cand, X_indices, weights = _draw(sober, n_rec)
        return X_cand, X_indices, torch.ones(n_rec, dtype=weights.dtype, device=weights.device) / n_rec
    one = len(w_acc) == 1                                     # (a single accepted draw: nothing to concatenate)
    weights = sober.cleansing_weights((w_acc[0].clone() if one else torch.cat(w_acc)).contiguous())
    return (X_acc[0] if one else torch.vstack(X_acc)), ((I_acc[0] if one else torch.vstack(I_acc)) if I_acc else None), weights


def sampling_candidates(sober, n_rec, n_nys, verbose=False):
    """SOBER/_sampler.py:264-323 -> (X_cand, X_nys, weights)."""
    assert n_rec > n_nys
    X_cand, X_indices, weights = _draw(sober, n_rec)
    if sober.check_weights(weights):
        _update_prior(sober, X_indices if X_indices is not None else X_cand, weights)
        sober.thresh = n_nys
        X_cand, _, weights = _recursive_draws(sober, n_rec, sober.thresh)
    else:
        X_cand, X_indices, weights = _recursive_draws(sober, n_rec, sober.thresh)
        if sober.flag:
            return X_cand, X_cand[:n_nys], weights
        _update_prior(sober, X_indices if X_indices is not None else X_cand, weights)
        sober.thresh = n_nys
        X_cand, _, weights = _recursive_draws(sober, n_rec, sober.thresh)
    X_nys = sober.nystrom_subsample(X_cand, weights, n_nys)
    sober.thresh = sober.thresh_initial
    return X_cand, X_nys, weights
