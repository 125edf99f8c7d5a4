"""The REAL-MODEL branch of the boundary (sober_amd/_kernel.py: spec_from_model): a gpytorch exact GP is read by the
CLASS NAMES and attribute shapes gpytorch gives it -- what /root/reference/SOBER/_sober_wrapper.py:611-638 builds and
SOBER/_gp.py:268-276 reads.  gpytorch is not installed here (requirements.txt:2 pins 1.10; SURVEY.md 8c: boundary
unpinned), so the classes below carry gpytorch's names and shapes and nothing else:
  ScaleKernel.outputscale            0-d tensor (softplus of raw_outputscale)
  RBFKernel / MaternKernel.lengthscale   (1, d) with ARD, (1, 1) without;  MaternKernel.nu
  likelihood.noise                   (1,)
  ExactGP.prediction_strategy        None until the first call in eval mode; then .covar_cache (n, n) and .mean_cache (n,)
  mean_module.constant               0-d tensor (ConstantMean)
Driven through `sober_amd.Kernel(model)`, `sober_amd.recombination` and `sober_amd.Sober(prior, model)` against the
reference goldens (CPU test double for the bulk kernels; tests/test_hip_parity.py repeats one case on the GPU)."""
import os
import types
import warnings

import numpy as np
import pytest
import torch

import sober_amd
from sober_amd._kernel import spec_from_model
from tests._oracle_ops import OracleOps
from tests.golden.synth import SEED_CALL, load_case

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


class RBFKernel:
    def __init__(self, lengthscale):
        self.lengthscale = lengthscale


class MaternKernel:
    def __init__(self, lengthscale, nu=2.5):
        self.lengthscale, self.nu = lengthscale, nu


class TanimotoKernel:
    pass


class ScaleKernel:
    def __init__(self, base_kernel, outputscale):
        self.base_kernel = base_kernel
        self.raw_outputscale = torch.nn.Parameter(torch.log(torch.expm1(torch.tensor(float(outputscale), dtype=torch.float64))))

    @property
    def outputscale(self):                                   # gpytorch: softplus of the raw parameter, 0-d
        return torch.nn.functional.softplus(self.raw_outputscale)


class _Strategy:
    def __init__(self, covar_cache, mean_cache):
        self.covar_cache, self.mean_cache = covar_cache, mean_cache


class ExactGPModel:
    """gpytorch.models.ExactGP as far as the path reads it."""

    def __init__(self, spec, ard_shape=True, scale=True):
        d = spec.X_obs.shape[1]
        ls = spec.lengthscale.reshape(1, -1).clone()
        if ls.shape[1] == 1 and not ard_shape:
            ls = ls.reshape(1, 1)
        base = {"rbf": lambda: RBFKernel(torch.nn.Parameter(ls)),
                "matern52": lambda: MaternKernel(torch.nn.Parameter(ls)),
                "tanimoto": TanimotoKernel}[spec.kind]()
        self.covar_module = ScaleKernel(base, spec.outputscale) if scale else base
        self.train_inputs = (spec.X_obs,)
        self.train_targets = torch.zeros(len(spec.X_obs), dtype=torch.float64)
        self.likelihood = types.SimpleNamespace(noise=torch.tensor([spec.noise], dtype=torch.float64))
        self.mean_module = types.SimpleNamespace(constant=torch.tensor(spec.mean_const, dtype=torch.float64))
        self.prediction_strategy = None                      # gpytorch: built by the first call in eval mode
        self._spec, self.calls, self.in_eval = spec, [], False
        del d

    def eval(self):
        self.in_eval = True
        return self

    def __call__(self, x):
        assert self.in_eval, "SOBER/_gp.py:273: model.eval() comes first"
        self.calls.append(tuple(x.shape))
        alpha = self._spec.alpha if self._spec.alpha is not None else torch.zeros(len(self._spec.X_obs), dtype=torch.float64)
        self.prediction_strategy = _Strategy(self._spec.S_cache, alpha)
        return None


def _ks(spec):
    return sober_amd.KernelSpec(spec.kind, spec.lengthscale, spec.outputscale, spec.X_obs, spec.S_cache, spec.noise,
                                spec.mean_const, spec.alpha)


@pytest.mark.parametrize("name", ["cfg1_rbf_ard", "rbf_b30", "matern_b20", "tanimoto_weighted", "rbf_weighted"])
def test_spec_read_off_a_gpytorch_shaped_model(name):
    case, inp, spec, z = load_case(os.path.join(GOLD, f"recomb_{name}.npz"))
    model = ExactGPModel(spec)
    got = spec_from_model(model)
    # the missing-cache path of SOBER/_gp.py:272-276: eval(), one call on the first observation, then the cache
    assert model.in_eval and model.calls == [(1, spec.X_obs.shape[1])]
    want = _ks(spec)
    assert got.kind == want.kind
    if spec.kind != "tanimoto":
        assert got.lengthscale.dim() == 1                                 # (1, d) -> (d,)
        np.testing.assert_array_equal(got.lengthscale.numpy(), want.lengthscale.reshape(-1).numpy())
    np.testing.assert_allclose(got.outputscale, want.outputscale, rtol=1e-15, atol=0)   # softplus(inverse softplus)
    assert got.noise == want.noise and got.mean_const == want.mean_const
    assert got.X_obs is spec.X_obs and got.S_cache is spec.S_cache
    if spec.alpha is not None:
        np.testing.assert_array_equal(got.alpha.numpy(), spec.alpha.numpy())
    # a second read finds the cache: no further call
    spec_from_model(model)
    assert len(model.calls) == 1


def test_unsupported_covar_modules_say_so():
    case, inp, spec, z = load_case(os.path.join(GOLD, "recomb_matern_b20.npz"))
    model = ExactGPModel(spec)
    model.covar_module.base_kernel.nu = 1.5
    with pytest.raises(ValueError, match="nu=1.5"):
        spec_from_model(model)

    class PeriodicKernel:
        pass
    model.covar_module = PeriodicKernel()
    with pytest.raises(ValueError, match="PeriodicKernel"):
        spec_from_model(model)


@pytest.mark.parametrize("name", ["cfg1_rbf_ard", "matern_b20", "tanimoto_weighted"])
def test_recombination_with_a_live_model_reproduces_the_golden(name):
    """`sober_amd.Kernel(model)` -- not a KernelSpec -- through `recombination`: the reference's result."""
    case, inp, spec, z = load_case(os.path.join(GOLD, f"recomb_{name}.npz"))
    model = ExactGPModel(spec, scale=spec.kind != "tanimoto" or spec.outputscale != 1.0)
    kernel = sober_amd.Kernel(model, case["mode"])
    mu = _t(inp["mu0"].copy())
    torch.manual_seed(SEED_CALL)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        idx, w = sober_amd.recombination(_t(inp["X_cand"]), _t(inp["X_nys"]), case["b"], kernel, init_weights=mu,
                                         _ops=OracleOps())
    assert np.array_equal(idx.numpy(), z["idx"])
    np.testing.assert_allclose(w.numpy(), z["w"], rtol=1e-9)
    # the model is LIVE: a lengthscale changed in place is what the next call reads (SOBER/_gp.py:292-294)
    if spec.kind != "tanimoto":
        with torch.no_grad():
            model.covar_module.base_kernel.lengthscale.mul_(1.5)
        assert torch.equal(kernel.spec("cpu").lengthscale, spec.lengthscale.reshape(-1) * 1.5)


@pytest.mark.gpu
def test_sober_on_a_gpytorch_shaped_model_vs_reference_fixture():
    """`sober_amd.Sober(prior, model)` on the device with the model read BY CLASS NAMES (ScaleKernel / base kernel,
    lengthscale (1, d), prediction_strategy filled by the first eval call, mean_cache, ConstantMean): the batch the
    reference's own `Sober.next_batch` produced (fixture of tests/golden/make_golden.py::gen_sober), dataset prior
    with pruning, indices and weights."""
    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    from tests.golden import make_golden as MG
    z = np.load(os.path.join(GOLD, "sober_next_batch.npz"))
    c = MG.SOBER_CASES["dataset"]
    _, spec = MG.sober_model(c)
    rng = np.random.default_rng(c["pool_seed"])
    pool = _t((rng.random((c["pool_n"], c["d"])) < c["pool_p"]).astype(np.float64)).to(dev)
    model = ExactGPModel(spec, scale=True)
    assert model.prediction_strategy is None
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for rw in (False, True):
            sober = sober_amd.Sober(MG.DatasetPrior(pool), model, kernel_type=c["kernel_type"], dataset_pruning=True)
            sober.reference_stream = True
            torch.manual_seed(c["seed_call"])
            a, Xb = sober.next_batch(c["n_rec"], c["n_nys"], c["batch"], return_weights=rw)
            tag = f"dataset_p1_w{int(rw)}"
            if rw:
                np.testing.assert_allclose(a.cpu().numpy(), z[tag + "_first"], rtol=1e-7)
            else:
                assert np.array_equal(a.cpu().numpy(), z[tag + "_first"])
            assert np.array_equal(Xb.cpu().numpy(), z[tag + "_X"])
    assert model.calls == [(1, spec.X_obs.shape[1])]         # the cache was built once, by the path's own first read
    # a continuous prior: ARD lengthscale (1, d) through the KMeans Nystrom subsample and the level kernels
    c = MG.SOBER_CASES["continuous"]
    _, spec = MG.sober_model(c)
    model = ExactGPModel(spec, scale=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sober = sober_amd.Sober(MG.UniformPrior(c["d"], device=dev), model, kernel_type=c["kernel_type"],
                                prior_updater=lambda s_, X, w: None)    # (default candidate_funnel: sober_amd/_sampled_prior.py)
        torch.manual_seed(c["seed_call"])
        Xb = sober.next_batch(c["n_rec"], c["n_nys"], c["batch"])
    assert np.array_equal(Xb.cpu().numpy(), z["continuous_X"])
