"""Seeded synthetic inputs shared by the golden generator and the tests.

numpy's PCG64 ``default_rng(seed)`` streams are stable across numpy versions and
platforms, so large cases store only the seed (plus checksums of the inputs).
"""
import numpy as np
import torch

from oracle import sober_oracle as O

SEED_CALL = 123     # torch.manual_seed immediately before recombination


def synth(case):
    rng = np.random.default_rng(case["seed"])
    N, M, d, n_obs = case["N"], case["M"], case["d"], case["n_obs"]
    if case["kind"] == O.TANIMOTO:
        p = case.get("bit_p", 0.1)
        X_cand = (rng.random((N, d)) < p).astype(np.float64)
        X_obs = (rng.random((n_obs, d)) < p).astype(np.float64)
    else:
        X_cand = rng.random((N, d))
        X_obs = rng.random((n_obs, d))
    mu0 = rng.random(N)
    if case.get("zero_frac", 0) > 0:
        mu0[rng.random(N) < case["zero_frac"]] = 0.0
    mu0 /= mu0.sum()
    X_nys = X_cand[rng.permutation(N)[:M]].copy()
    y_obs = rng.standard_normal(n_obs)
    if case["kind"] == O.TANIMOTO:
        ls = np.ones(1)
    elif case.get("ard", False):
        ls = 0.25 * np.sqrt(d) * (1.0 + np.arange(d) / d)
    else:
        ls = np.array([0.25 * np.sqrt(d)])
    return dict(X_cand=X_cand, X_obs=X_obs, mu0=mu0, X_nys=X_nys, y_obs=y_obs, lengthscale=ls)


def build_spec(case, inp):
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    return O.make_spec(case["kind"], t(inp["X_obs"]), t(inp["lengthscale"]),
                       outputscale=case.get("outputscale", 1.0), noise=case.get("noise", 1e-2),
                       mean_const=case.get("mean_const", 0.0), y_obs=t(inp["y_obs"]))


def checksum(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return np.array([a.sum(), np.abs(a).sum(), (a * np.cos(np.arange(a.size)).reshape(a.shape)).sum()])


def calc_obj_fn(samp):
    # a smooth deterministic "acquisition" for the calc_obj branch
    return torch.sin(3.0 * samp).sum(-1) + samp[:, 0]


def case_from_fixture(z):
    """Rebuild the case dict (for ``synth``) from the scalar fields of a
    ``recomb_*.npz`` fixture."""
    return dict(kind=str(z["kind"]), mode=str(z["mode"]), N=int(z["N"]), M=int(z["M"]), d=int(z["d"]),
                b=int(z["b"]), n_obs=int(z["n_obs"]), seed=int(z["seed"]), ard=bool(z["ard"]),
                bit_p=float(z["bit_p"]), zero_frac=float(z["zero_frac"]),
                outputscale=float(z["outputscale"]), noise=float(z["noise"]),
                mean_const=float(z["mean_const"]), calc_obj=bool(z["calc_obj"]))


def load_case(path):
    """-> (case, inputs dict of numpy arrays, GPSpec, fixture).  Inputs are read
    from the fixture when stored, regenerated from the seed otherwise (and checked
    against the stored checksums)."""
    z = np.load(path, allow_pickle=False)
    case = case_from_fixture(z)
    inp = synth(case)
    if "X_cand" in z.files:
        for k in ("X_cand", "X_nys", "mu0", "X_obs", "y_obs"):
            assert np.array_equal(inp[k], z[k]), f"synth() drifted from fixture for {k}"
    else:
        for k in ("X_cand", "X_nys", "mu0"):
            assert np.allclose(checksum(inp[k]), z["cks_" + k], rtol=1e-13, atol=0), k
    spec = build_spec(case, inp)
    assert np.allclose(checksum(spec.S_cache.numpy()), z["cks_S_cache"], rtol=1e-9), "S_cache drift"
    return case, inp, spec, z
