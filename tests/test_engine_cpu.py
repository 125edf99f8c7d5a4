"""Host logic of the product (sober_amd._engine) driven through a CPU test double of the device
ops (tests/_oracle_ops.py), against the reference-generated goldens: same indices, weights
within 1e-9 (the engine sums set-first, the reference per candidate)."""
import glob
import os
import warnings

import numpy as np
import pytest
import torch

import sober_amd
from sober_amd._engine import RecombinationEngine, car_host, ker_svd_sparsify_host, survivors_before
from tests._oracle_ops import OracleOps
from tests.golden.synth import SEED_CALL, calc_obj_fn, load_case

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = sorted(p for p in glob.glob(os.path.join(GOLD, "recomb_*.npz")) if "cfg2" not in p)


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def kernel_from(spec, mode):
    ks = sober_amd.KernelSpec(spec.kind, spec.lengthscale, spec.outputscale, spec.X_obs, spec.S_cache,
                              spec.noise, spec.mean_const, spec.alpha)
    return sober_amd.Kernel(ks, mode)


def run_engine(path, ops=None, trace=None):
    case, inp, spec, z = load_case(path)
    mu = _t(inp["mu0"].copy())
    torch.manual_seed(SEED_CALL)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        idx, w = sober_amd.recombination(_t(inp["X_cand"]), _t(inp["X_nys"]), case["b"],
                                         kernel_from(spec, case["mode"]), init_weights=mu,
                                         calc_obj=calc_obj_fn if case["calc_obj"] else None,
                                         _ops=ops or OracleOps(), _trace=trace)
    return z, idx, w, mu


@pytest.mark.parametrize("path", CASES, ids=lambda p: os.path.basename(p)[7:-4])
def test_engine_matches_reference(path):
    trace = {}
    z, idx, w, mu = run_engine(path, trace=trace)
    assert np.array_equal(idx.numpy(), z["idx"])
    np.testing.assert_allclose(w.numpy(), z["w"], rtol=1e-9, atol=0)
    nz = torch.nonzero(mu).flatten().numpy()
    assert np.array_equal(nz, z["mu_after_idx"])                        # Q3
    np.testing.assert_allclose(mu.numpy()[nz], z["mu_after_val"], rtol=1e-9)
    assert len(trace["levels"]) == int(z["n_levels"])
    for i, lv in enumerate(trace["levels"]):
        assert np.array_equal(lv["idx_star"].numpy(), z[f"L{i}_idx_star"]), i
        np.testing.assert_allclose(lv["tot_weights"].numpy(), z[f"L{i}_tot_weights"], rtol=1e-10)
        np.testing.assert_allclose(lv["X_tmp"].numpy(), z[f"L{i}_X_tmp"], rtol=1e-7, atol=1e-11)


def test_car_host_bit_exact():
    z = np.load(os.path.join(GOLD, "recomb_matern_b20.npz"))
    for i in range(int(z["n_levels"])):
        w, idx = car_host(_t(z[f"L{i}_X_tmp"]), _t(z[f"L{i}_tot_weights"].copy()))
        assert np.array_equal(idx.numpy(), z[f"L{i}_idx_star"])
        assert np.array_equal(w.numpy(), z[f"L{i}_w_star"])


def test_nystrom_basis_from_reference_gram():
    """Product host path (make_cov_psd with the symmetric eigen-solver + svd_lowrank) reproduces the
    reference's U from the reference's Gram."""
    for name in ("cfg1_rbf_ard", "rbf_b30", "matern_b20", "tanimoto_weighted", "rbf_basekernel"):
        z = np.load(os.path.join(GOLD, f"recomb_{name}.npz"))
        torch.manual_seed(SEED_CALL)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            _, U = ker_svd_sparsify_host(_t(z["gram_in"].copy()), int(z["b"]) - 1)
        assert np.array_equal(U.numpy(), z["U"]), name


def test_make_cov_psd_matches_reference():
    z = np.load(os.path.join(GOLD, "psd.npz"))
    tm = sober_amd.SafeTensorOperator()
    for k in "abcde":
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out = tm.make_cov_psd(_t(z[f"{k}_in"].copy()))
        assert np.array_equal(out.numpy(), z[f"{k}_out"], equal_nan=True), k


def test_survivors_before_is_a_prefix_count():
    rng = np.random.default_rng(0)
    for _ in range(50):
        S = int(rng.integers(4, 40))
        E = int(rng.integers(1, 9))
        r = int(rng.integers(0, S))
        R = E * S + r
        kept = np.sort(rng.choice(S, size=int(rng.integers(1, S)), replace=False))
        keep = np.zeros(S, bool)
        keep[kept] = True
        last = bool(keep[S - 1])
        prefix = np.r_[0, np.cumsum(keep)].tolist()
        alive = np.array([(keep[p % S] if p < E * S else last) for p in range(R)])
        for p in range(R + 1):
            assert survivors_before(p, S, E, prefix, int(keep.sum()), last) == int(alive[:p].sum())


def test_api_errors():
    spec = sober_amd.KernelSpec("rbf", torch.ones(1, dtype=torch.double), 1.0,
                                torch.zeros(3, 2, dtype=torch.double), torch.eye(3, dtype=torch.double))
    x = torch.rand(50, 2, dtype=torch.double)
    with pytest.raises(TypeError):
        sober_amd.recombination(x, x[:10], 5, "not a kernel", _ops=OracleOps())
    with pytest.raises(ValueError):
        sober_amd.recombination(x, x[:10], 5, sober_amd.Kernel(spec, "nope"), _ops=OracleOps())


def test_no_cpu_fallback():
    """On a box without a HIP device the product path must fail loudly."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    spec = sober_amd.KernelSpec("rbf", torch.ones(1, dtype=torch.double), 1.0,
                                torch.zeros(3, 2, dtype=torch.double), torch.eye(3, dtype=torch.double))
    x = torch.rand(50, 2, dtype=torch.double)
    with pytest.raises(RuntimeError):
        sober_amd.recombination(x, x[:10], 5, sober_amd.Kernel(spec))
    with pytest.raises(RuntimeError):
        sober_amd.KMeans(x, 4)
    with pytest.raises(RuntimeError):
        sober_amd.WeightsStabiliser().cleansing_weights(torch.rand(8, dtype=torch.double))


def test_wkde_fit_matches_reference():
    """Host-side fit of sober_amd.WeightedKernelDensityEstimation (no GPU needed) against the golden."""
    z = np.load(os.path.join(GOLD, "wkde.npz"))
    for tag in "ab":
        d = z[f"{tag}_X"].shape[1]
        bounds = torch.tensor([[0.0] * d, [1.0] * d], dtype=torch.double) if bool(z[f"{tag}_bounded"]) else None
        torch.manual_seed(11)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            kde = sober_amd.WeightedKernelDensityEstimation(_t(z[f"{tag}_X"].copy()), _t(z[f"{tag}_W"].copy()), d,
                                                            bounds=bounds, n_kde=int(z[f"{tag}_n_kde"]))
        assert np.array_equal(kde.Xobs.numpy(), z[f"{tag}_Xobs"])
        assert np.array_equal(kde.weights.numpy(), z[f"{tag}_weights"])
        np.testing.assert_allclose(kde.covariance.numpy(), z[f"{tag}_cov"], rtol=1e-13)
        assert abs(float(kde.bw) - float(z[f"{tag}_bw"])) < 1e-15
        with pytest.raises(RuntimeError):
            kde.pdf(_t(z[f"{tag}_Xq"]))                       # CPU tensor: no fallback


def test_ladder_borderline_rule():
    """The rule that sends a numerically undecided jitter rung to the host's LAPACK (sober_amd/_ops_hip.py)."""
    from sober_amd._ops_hip import HipOps
    f = HipOps.ladder_borderline
    ok3 = [1, 1, 1, 0, 0]
    assert not f(ok3, [-1e-3, -5e-4, -1e-4, 2e-3, 5e-3], 1.0)            # clear
    assert f(ok3, [-1e-3, -5e-4, -1e-4, 2e-12, 5e-3], 1.0)               # accepted rung numerically singular
    assert f(ok3, [-1e-3, -5e-4, -3e-12, 2e-3, 5e-3], 1.0)               # rejected by a hair
    assert not f([0, 0], [0.5, 0.6], 1.0)                                # PSD at rung 0
    assert f([0, 0], [1e-13, 0.6], 1.0)
    assert not f([1, 1], [-1e-2, -1e-3], 1.0)                            # no rung: diagonal fallback, clearly
    assert f([1, 1], [-1e-2, -1e-12], 1.0)
    assert f([1, 0], [float("nan"), 1.0], 1.0)
