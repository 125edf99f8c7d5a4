"""Drop-in proof against the reference's OWN callers (build container only: needs /root/reference; skipped on the
GPU box, where tests/test_hip_parity.py::test_sober_next_batch_vs_reference replays the same calls against fixtures).

The reference's `RecombinationSampler` (SOBER/_sampler.py:11-59) and `Sober` (SOBER/_sober.py:9-195) are loaded from
their own files and run UNCHANGED, with the three names of the hot path swapped for this package's:
`SOBER._sampler.recombination -> sober_amd.recombination`, `SOBER._sober.Kernel -> sober_amd.Kernel` (and with them
`RecombinationEngine`, the host logic of the product).  Without a GPU the bulk kernels behind `recombination` are the
CPU test double (tests/_oracle_ops.py); everything else is the code that ships.  Checked: the reference's callers
get the reference's results (goldens / an unpatched run) through the swapped functions, for `sampling_recombination`
and for the three return shapes of `Sober.next_batch`."""
import os
import warnings

import numpy as np
import pytest
import torch

REF = "/root/reference/SOBER"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference checkout lives in the build container only")

import sober_amd                                                        # noqa: E402
from tests._oracle_ops import OracleOps                                 # noqa: E402
from tests.golden.synth import SEED_CALL, build_spec, load_case, synth  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


@pytest.fixture(scope="module")
def ref():
    from tests.golden import make_golden as MG
    mods = MG.load_sampler_sober(MG.load_reference())
    mods["MG"] = MG
    return mods


def _ours(*a, **k):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return sober_amd.recombination(*a, **k, _ops=OracleOps())


class _Swap:
    """The three names of the hot path, swapped inside the reference's modules."""

    def __init__(self, ref):
        self.ref = ref

    def __enter__(self):
        r = self.ref
        self.old = (r["_sampler"].recombination, r["_sober"].Kernel)
        r["_sampler"].recombination = _ours
        r["_sober"].Kernel = sober_amd.Kernel
        return self

    def __exit__(self, *exc):
        self.ref["_sampler"].recombination, self.ref["_sober"].Kernel = self.old


@pytest.mark.parametrize("name", ["cfg1_rbf_ard", "rbf_b30", "matern_b20", "tanimoto_weighted", "rbf_zero_weights"])
def test_reference_sampler_with_swapped_recombination(ref, name):
    """SOBER/_sampler.py:27-59 calling sober_amd.recombination with a sober_amd.Kernel: the reference golden."""
    case, inp, spec, z = load_case(os.path.join(GOLD, f"recomb_{name}.npz"))
    ks = sober_amd.KernelSpec(spec.kind, spec.lengthscale, spec.outputscale, spec.X_obs, spec.S_cache, spec.noise,
                              spec.mean_const, spec.alpha)
    with _Swap(ref):
        sampler = ref["_sampler"].RecombinationSampler(sober_amd.Kernel(ks, case["mode"]))
        w0 = _t(inp["mu0"].copy())
        torch.manual_seed(SEED_CALL)
        idx, w = sampler.sampling_recombination(_t(inp["X_cand"]), _t(inp["X_nys"]), w0, case["b"])
    assert np.array_equal(idx.numpy(), z["idx"])
    np.testing.assert_allclose(w.numpy(), z["w"], rtol=1e-9)
    nz = torch.nonzero(w0).flatten().numpy()
    assert np.array_equal(nz, z["mu_after_idx"])                            # Q3 through the reference's caller


def _model(ref, kind, d, n_obs, seed):
    case = dict(kind=kind, N=64, M=8, d=d, n_obs=n_obs, seed=seed, mean_const=0.2)
    inp = synth(case)
    spec = build_spec(case, inp)
    model = ref["MG"].DuckModel(spec)
    model.train_targets = torch.zeros(n_obs, dtype=torch.double)
    # (a gpytorch model is read by its class names, sober_amd/_kernel.py:spec_from_model; the duck model of the
    #  fixtures has none, so it carries the spec itself)
    model.kernel_spec = sober_amd.KernelSpec(spec.kind, spec.lengthscale, spec.outputscale, spec.X_obs, spec.S_cache,
                                             spec.noise, spec.mean_const, spec.alpha)
    return model


def _next_batch(ref, prior, model, swapped, seed, **kw):
    import contextlib
    cm = _Swap(ref) if swapped else contextlib.nullcontext()
    with cm, warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sober = ref["_sober"].Sober(prior, model, kernel_type=kw.pop("kernel_type", "predictive_covariance"),
                                    dataset_pruning=kw.pop("dataset_pruning", True))
        torch.manual_seed(seed)
        return sober.next_batch(**kw)


@pytest.mark.parametrize("pruning", [True, False])
def test_reference_sober_next_batch_dataset_prior(ref, pruning):
    """`Sober.next_batch` on a dataset prior (SOBER/_sober.py:158-195), both dataset return shapes and the
    return_weights shape: swapped == unswapped."""
    rng = np.random.default_rng(3)
    pool = _t((rng.random((900, 64)) < 0.1).astype(np.float64))
    model = _model(ref, "tanimoto", 64, 20, 77)
    prior = ref["MG"].DatasetPrior(pool)
    for rw in (False, True):
        kw = dict(n_rec=600, n_nys=40, batch_size=8, return_weights=rw, dataset_pruning=pruning,
                  kernel_type="weighted_predictive_covariance")
        a = _next_batch(ref, prior, model, False, 123, **dict(kw))
        b = _next_batch(ref, prior, model, True, 123, **dict(kw))
        assert len(a) == len(b) == 2
        if rw:                                                              # (w_rchq, X_batch)
            np.testing.assert_allclose(b[0].numpy(), a[0].numpy(), rtol=1e-9)
        else:                                                               # (idx_rchq, X_batch)
            assert torch.equal(a[0], b[0]) and a[0].dtype == torch.int64
        assert torch.equal(a[1], b[1])


def test_reference_sober_next_batch_sampled_prior(ref):
    """The third return shape (X_batch alone) through `sampling_candidates` (SOBER/_sampler.py:264-323, KMeans
    Nystrom subsample for a continuous prior): swapped == unswapped."""
    model = _model(ref, "rbf", 3, 15, 78)
    prior = ref["MG"].UniformPrior(3)
    kw = dict(n_rec=1500, n_nys=48, batch_size=8)
    a = _next_batch(ref, prior, model, False, 321, **dict(kw))
    b = _next_batch(ref, prior, model, True, 321, **dict(kw))
    assert isinstance(a, torch.Tensor) and a.shape == (8, 3)
    assert torch.equal(a, b)


def test_should_reset_prior_equals_the_reference(ref):
    """SOBER/_sober.py:84-123 (when a sampled prior is reset) against `sober_amd.Sober.should_reset_prior` on random
    observation histories; and `next_batch` asks for the caller's `prior_initialiser` instead of dropping
    `recycle_prior` (SOBER/_sober.py:152-155)."""
    import types
    rng = np.random.default_rng(5)
    RefSober = ref["_sober"].Sober
    n_true = 0
    for trial in range(300):
        n_init = int(rng.integers(1, 12))
        bs = int(rng.integers(1, 9))
        n_batches = int(rng.integers(0, 7))
        extra = int(rng.integers(0, bs)) if n_batches and rng.random() < 0.3 else 0
        y = torch.from_numpy(rng.standard_normal(n_init + n_batches * bs + extra))
        if rng.random() < 0.3:                                           # a maximum early in the history
            y[int(rng.integers(0, n_init))] = 10.0
        model = types.SimpleNamespace(train_targets=y)
        for recycle in (True, False):
            a = object.__new__(RefSober)
            a.fbgp, a.is_bq, a.n_init, a.n_batches_until_reset = False, False, n_init, 3
            a.pi = types.SimpleNamespace(model=model)
            a.tensor = torch.tensor
            b = object.__new__(sober_amd.Sober)
            b.n_init, b.n_batches_until_reset, b.pi = n_init, 3, types.SimpleNamespace(model=model)
            ra, rb = bool(a.should_reset_prior(bs, recycle)), bool(b.should_reset_prior(bs, recycle))
            assert ra == rb, (trial, n_init, bs, n_batches, extra, recycle)
            n_true += ra
    assert 50 < n_true < 550                                             # (both verdicts occur)

    # a reset that is due and no hook: an error that says so, not a silently recycled prior
    s = object.__new__(sober_amd.Sober)
    s.label, s.n_init, s.n_batches_until_reset, s.prior_initialiser, s.candidate_funnel = "continuous", 2, 3, None, None
    s.pi = types.SimpleNamespace(model=types.SimpleNamespace(train_targets=torch.tensor([5.0, 1, 0, 0, 0, 0, 0, 0])))
    assert s.should_reset_prior(2, True)
    with pytest.raises(NotImplementedError, match="prior_initialiser"):
        s.next_batch(100, 10, 2)
    called = []
    s.prior_initialiser = lambda smp: called.append(smp)
    class _Stop(Exception):
        pass

    def _stop(*a, **k):                                                  # (stop right behind the reset)
        raise _Stop()
    s.candidate_funnel = _stop
    with pytest.raises(_Stop):
        s.next_batch(100, 10, 2)
    assert called == [s]
