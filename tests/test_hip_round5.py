"""GPU tests added in round 5 (run with -m gpu on an MI355X), all through the C ABI:
  * the SHIPPED KMeans path (E step screened on the BF16 matrix cores, the default at cfg-2 / cfg-4 shapes) against labels
    and centroids the REFERENCE's own KMeans produced on such a shape (tests/golden/kmeans_screened.npz), not only against
    the device's exact kernel;
  * the fused GP prediction kernel (csrc/predict.hip) against the materialised route and the oracle;
  * the pivots' screened ratio test: the accuracy of the v_rcp_f64 seed it rests on, and bit equality of every output with the
    exact test (kernel level: 75+ steps incl. ties, negative masses, multi-CU sizes; whole sampling steps)."""
import os

import numpy as np
import pytest
import torch

import sober_amd

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "the -m gpu tests need the MI355X"
    from sober_amd import _native
    _native.load()
    return torch.device("cuda:0")


@pytest.mark.parametrize("tag", ["unit", "offset"])
def test_kmeans_screened_default_path_vs_reference(tag, dev):
    """SOBER/_weights.py:100-126 at 120000 x 10 into 64 clusters (the reference's (N, K, D) temporary: 614 MB), unit cube
    and the same pool moved by 1e4: `sober_amd.KMeans` takes the screened E step by default here
    (sober_kmeans_ws_bytes asks for the screen's workspace from N * ceil((d + 1) / 4) >= 300000 on) -- labels bit-equal
    to the reference's, centroids to 1e-11."""
    from sober_amd import _native as nat
    z = np.load(os.path.join(GOLD, "kmeans_screened.npz"))
    N, d, K = int(z["N"]), int(z["d"]), int(z["K"])
    lib = nat.load()
    assert lib.sober_kmeans_ws_bytes(N, d, K) == lib.sober_kmeans_ws_bytes_screened(N, d, K) > 0     # the screen IS the default here
    x = np.random.default_rng(int(z["seed"])).random((N, d)) + float(z[f"{tag}_off"])
    cl, c = sober_amd.KMeans(_t(x).to(dev), K=K)
    assert np.array_equal(cl.cpu().numpy().astype(np.uint8), z[f"{tag}_cl"])
    np.testing.assert_allclose(c.cpu().numpy(), z[f"{tag}_c"], rtol=1e-11)


@pytest.mark.parametrize("kind,d,n_obs,N", [("rbf", 10, 200, 10007), ("matern52", 6, 37, 4096), ("rbf", 3, 255, 777),
                                             ("tanimoto", 2048, 64, 3000), ("tanimoto", 100, 200, 1031)])
def test_fused_prediction_equals_the_materialised_route(kind, d, n_obs, N, dev, monkeypatch):
    """csrc/predict.hip (one launch: K(X_obs, x) once into LDS, W k on the FP64 matrix cores) against the four-launch
    route it replaces (SOBER_PREDICT_MATERIALISED=1: posterior mean, materialised K(X_obs, pool), V = W KX, the column-wise
    quadratic form): mean, variance and pi(x) of SOBER/_gp.py:212-238 / _pi.py:20-38 -- the same numbers up to the order
    of the n_obs-term sums."""
    from oracle import sober_oracle as O
    rng = np.random.default_rng(n_obs + N)
    if kind == "tanimoto":
        X = (rng.random((N, d)) < 0.1).astype(np.float64)
        Xo = (rng.random((n_obs, d)) < 0.1).astype(np.float64)
        ls = np.ones(1)
    else:
        X, Xo = rng.random((N, d)), rng.random((n_obs, d))
        ls = (0.3 + 0.4 * rng.random(d)) * np.sqrt(d)
    spec = O.make_spec(kind, _t(Xo), _t(ls), outputscale=1.7, noise=1e-2, mean_const=0.3, y_obs=_t(rng.standard_normal(n_obs)))
    ks = sober_amd.KernelSpec(spec.kind, spec.lengthscale, spec.outputscale, spec.X_obs, spec.S_cache, spec.noise,
                              spec.mean_const, spec.alpha)
    Xd = _t(X).to(dev)
    out = []
    for mat in (True, False):
        if mat:
            monkeypatch.setenv("SOBER_PREDICT_MATERIALISED", "1")
        else:
            monkeypatch.delenv("SOBER_PREDICT_MATERIALISED", raising=False)
        mean, var = sober_amd.predict(Xd, ks)
        w = sober_amd.PI(ks)(Xd)
        lw = sober_amd.PI(ks)(Xd, log=True)
        out.append([v.cpu().numpy() for v in (mean, var, w, lw)])
    # (256 observations in three dimensions: alpha ~ 1e3, the mean is a cancelling sum -- 1e-10 of its terms' scale)
    for a, b, tol in zip(out[0], out[1], (1e-10, 1e-10, 1e-9, 1e-9)):
        scale = np.abs(a).max()
        assert np.abs(a - b).max() <= tol * max(scale, 1.0), (np.abs(a - b).max(), scale)
    # ... and the oracle (the reference's arithmetic) on a slice
    m_ref, v_ref = O.predict(_t(X[:500]), spec)
    np.testing.assert_allclose(out[1][0][:500], m_ref.numpy(), rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(out[1][1][:500], v_ref.numpy(), rtol=1e-8, atol=1e-12)


def test_rcp_seed_is_accurate_enough_for_the_screened_ratio_test(dev):
    """The pivots' screened ratio test (csrc/car.hip, sp_ratio_test) ranks mu * v_rcp_f64(col) by its high word and
    accepts a lone candidate within SP_BAND = 4 high-word steps (each >= 2^-21 relative): sound while the seed's
    relative error stays below 4 * 2^-22 = 2^-20.  Pinned here with a factor 4 to spare, over the whole exponent range."""
    from sober_amd import _native as nat
    rng = np.random.default_rng(7)
    x = rng.uniform(1.0, 2.0, 1 << 20) * np.exp2(rng.integers(-1000, 1000, 1 << 20)) * rng.choice([-1.0, 1.0], 1 << 20)
    x[:4] = [1.0, 3.0, 1.0 - 2.0 ** -53, 2.0 - 2.0 ** -52]
    r = nat.probe_rcp(_t(x).to(dev)).cpu().numpy()
    rel = np.abs(r * x - 1.0)
    print("v_rcp_f64 max relative error: 2^%.2f" % np.log2(rel.max()))
    assert rel.max() < 2.0 ** -22


def _car_once(nat, dev, X, mu):
    N = X.shape[0]
    keep = torch.empty(N + 1, dtype=torch.int32, device=dev)
    w = torch.zeros(N, dtype=torch.float64, device=dev)
    mo = torch.empty(N, dtype=torch.float64, device=dev)
    nat.car_device(_t(X).to(dev), _t(mu).to(dev), keep, w, keep[N:], mo)
    nk = int(keep[N].item())
    return keep[:N].cpu().numpy(), w.cpu().numpy(), nk, mo.cpu().numpy()


def test_screened_ratio_test_is_the_exact_one_bit_for_bit(dev, monkeypatch):
    """SOBER_CAR_EXACT_RATIO=1 switches the screened fast path of the pivots' ratio test off (four IEEE divisions and
    64-bit keys per pivot, the round-2..4 form).  The screen only ever picks the lane the exact test would pick and
    computes alpha and 1 / pivot for it with the same two divisions, so EVERY output must be bit-identical: kept ranks,
    weights, the full weight vector -- on the reference's level inputs, on random steps of every one-CU size class and of
    the multi-CU kernels' sizes (csrc/car_mc.hip: the same screen), with
    zero masses (alpha = 0 ties -> several candidates -> exact test), tiny negative masses (keys outside the screen's
    range -> exact test) and duplicated points."""
    import glob
    from sober_amd import _native as nat
    from tests.test_car_algorithm import nullspace_gebrd, pivots
    cases = []
    for p in sorted(glob.glob(os.path.join(GOLD, "recomb_*.npz"))):
        z = np.load(p)
        if "L0_X_tmp" not in z.files:
            continue
        for i in range(int(z["n_levels"])):
            X, mu = z[f"L{i}_X_tmp"], z[f"L{i}_tot_weights"]
            if X.shape[1] + 1 < X.shape[0] and nat.car_safe_supported(X.shape[0], X.shape[1] + 1):
                cases.append((f"{os.path.basename(p)}:L{i}", X, mu, None))
    rng = np.random.default_rng(2025)
    for k, (N, m) in enumerate([(200, 100), (200, 100), (150, 60), (128, 64), (100, 37), (64, 20), (40, 12), (199, 99),
                                (208, 100), (190, 80), (400, 200), (300, 180), (448, 230), (260, 40)] * 3):
        X = rng.standard_normal((N, m - 1)) * np.exp(-0.03 * rng.random() * np.arange(m - 1))[None, :]
        mu = rng.random(N) + 0.05
        kind = k % 3
        if kind == 1:
            mu[:: 5 + k % 4] = 0.0                                         # alpha = 0 ties
        elif kind == 2:
            mu[3::17] = -1e-18                                             # what rounding leaves behind an exact cancellation
            X[N // 2:N // 2 + 5] = X[:5]                                   # duplicated points
        A = np.vstack([np.ones(N), X.T])
        cases.append((f"random{k}:{N}x{m}:{kind}", X, mu, pivots(nullspace_gebrd(A), mu.copy()) if kind != 2 else None))
    n_oracle = 0
    for name, X, mu, ref in cases:
        outs = []
        for exact in (False, True):
            if exact:
                monkeypatch.setenv("SOBER_CAR_EXACT_RATIO", "1")
            else:
                monkeypatch.delenv("SOBER_CAR_EXACT_RATIO", raising=False)
            nat.reload_switches()
            outs.append(_car_once(nat, dev, X, mu))
        (k0, w0, n0, m0), (k1, w1, n1, m1) = outs
        assert n0 == n1 and n0 > 0, name
        assert np.array_equal(k0, k1), name
        assert np.array_equal(w0.view(np.int64), w1.view(np.int64)), name
        assert np.array_equal(m0.view(np.int64), m1.view(np.int64)), name
        if ref is not None:
            w_np, idx_np = ref
            assert np.array_equal(np.flatnonzero(k0 >= 0), idx_np), name
            np.testing.assert_allclose(w0[:n0], w_np, rtol=1e-8)
            n_oracle += 1
    monkeypatch.delenv("SOBER_CAR_EXACT_RATIO", raising=False)
    nat.reload_switches()
    assert len(cases) >= 70 and n_oracle >= 24


def test_whole_steps_with_and_without_the_screen_are_bit_equal(dev, monkeypatch):
    """The same A/B one level up: sampling steps through the queued level loop and the final level (cfg-2's golden, a Matern
    one with batch 100, a batch-200 step on the multi-CU kernels, a calc_obj one) with SOBER_CAR_EXACT_RATIO on and off --
    identical indices, weights and mutated init_weights, bit for bit."""
    import glob
    from sober_amd import _native as nat
    from tests.test_hip_parity import run_hip
    names = ["recomb_cfg2_rbf.npz", "recomb_matern_medium.npz", "recomb_rbf_medium.npz"]
    names += [os.path.basename(p) for p in sorted(glob.glob(os.path.join(GOLD, "recomb_*calc_obj*.npz")))[:1]]
    for name in names:
        path = os.path.join(GOLD, name)
        if not os.path.exists(path):
            continue
        out = []
        for exact in (False, True):
            if exact:
                monkeypatch.setenv("SOBER_CAR_EXACT_RATIO", "1")
            else:
                monkeypatch.delenv("SOBER_CAR_EXACT_RATIO", raising=False)
            nat.reload_switches()
            _, _, _, idx, w, mu = run_hip(path, dev)
            out.append((idx, w, mu))
        assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1]) and torch.equal(out[0][2], out[1][2]), name
    monkeypatch.delenv("SOBER_CAR_EXACT_RATIO", raising=False)
    nat.reload_switches()
