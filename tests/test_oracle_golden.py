"""The CPU oracle (oracle/sober_oracle.py) against the golden fixtures that
tests/golden/make_golden.py captured from the reference's own code.  This is
what pins the oracle; the GPU parity tests then compare the HIP path with it."""
import glob
import os
import warnings

import numpy as np
import pytest
import torch

from oracle import sober_oracle as O
from tests.golden.synth import SEED_CALL, calc_obj_fn, load_case

GOLD = os.path.join(os.path.dirname(__file__), "golden")
SMALL = sorted(p for p in glob.glob(os.path.join(GOLD, "recomb_*.npz"))
               if "cfg2" not in p and "medium" not in p)
MEDIUM = sorted(glob.glob(os.path.join(GOLD, "recomb_*medium.npz")))


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _run_oracle(path):
    case, inp, spec, z = load_case(path)
    kernel = O.Kernel(spec, case["mode"])
    mu = _t(inp["mu0"].copy())
    trace = {}
    torch.manual_seed(SEED_CALL)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        idx, w = O.recombination(_t(inp["X_cand"]), _t(inp["X_nys"]), case["b"], kernel,
                                 init_weights=mu, calc_obj=calc_obj_fn if case["calc_obj"] else None,
                                 trace=trace)
    return z, idx, w, mu, trace


@pytest.mark.parametrize("path", SMALL + MEDIUM, ids=lambda p: os.path.basename(p)[7:-4])
def test_recombination_matches_reference(path):
    z, idx, w, mu, trace = _run_oracle(path)
    # same torch ops in the same order in the same process family: bit-for-bit
    assert np.array_equal(idx.numpy(), z["idx"])
    assert np.array_equal(w.numpy(), z["w"])
    nz = torch.nonzero(mu).flatten().numpy()
    assert np.array_equal(nz, z["mu_after_idx"])                       # Q3: mutated in place
    assert np.array_equal(mu.numpy()[nz], z["mu_after_val"])
    assert len(trace["levels"]) == int(z["n_levels"])
    assert np.array_equal(trace["U"].numpy(), z["U"])
    for i, lv in enumerate(trace["levels"]):
        assert np.array_equal(lv["tot_weights"].numpy(), z[f"L{i}_tot_weights"])
        assert np.array_equal(lv["X_tmp"].numpy(), z[f"L{i}_X_tmp"])
        assert np.array_equal(lv["idx_star"].numpy(), z[f"L{i}_idx_star"])
        assert np.array_equal(lv["w_star"].numpy(), z[f"L{i}_w_star"])


def test_golden_cases_are_self_consistent():
    """Only cases whose reference run agrees with itself across MKL thread counts
    may gate parity."""
    for p in glob.glob(os.path.join(GOLD, "recomb_*.npz")):
        z = np.load(p)
        assert bool(z["self_threads_same_idx"]), p
        assert float(z["self_threads_dw"]) <= 1e-6, p
        assert float(z["condW"]) < 1e6, p


def test_invariants_without_leftovers():
    """Sum of weights preserved, positive weights, <= b points; with r == 0 at every
    level the Nystrom moments are matched exactly."""
    path = os.path.join(GOLD, "recomb_rbf_noleft.npz")
    case, inp, spec, z = load_case(path)
    z, idx, w, mu, trace = _run_oracle(path)
    assert all(lv.get("r", 0) == 0 for lv in trace["levels"])
    assert len(idx) <= case["b"] and (w > 0).all()
    assert abs(float(w.sum()) - float(inp["mu0"].sum())) < 1e-12
    kernel = O.Kernel(spec, case["mode"])
    C = kernel(_t(inp["X_nys"]), _t(inp["X_cand"]))
    lhs = trace["U"] @ (C @ _t(inp["mu0"]))
    rhs = trace["U"] @ (C[:, idx] @ w)
    assert float((lhs - rhs).norm() / lhs.norm()) < 1e-10


def test_car_matches_reference():
    z = np.load(os.path.join(GOLD, "recomb_rbf_b30.npz"))
    for i in range(int(z["n_levels"])):
        w, idx = O.tchernychova_lyons_car(_t(z[f"L{i}_X_tmp"]), _t(z[f"L{i}_tot_weights"].copy()))
        assert np.array_equal(idx.numpy(), z[f"L{i}_idx_star"])
        assert np.array_equal(w.numpy(), z[f"L{i}_w_star"])


def test_make_cov_psd_matches_reference():
    z = np.load(os.path.join(GOLD, "psd.npz"))
    expect = dict(a="psd", b="jitter", c="abs", d="abs", e="diag")
    for k, branch in expect.items():
        tr = {}
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out = O.make_cov_psd(_t(z[f"{k}_in"].copy()), trace=tr)
        assert np.array_equal(out.numpy(), z[f"{k}_out"], equal_nan=True), k
        assert tr["branch"] == branch, (k, tr)


def test_gram_psd_branch_recorded():
    """The reference's Gram is never bit-symmetric (A W A^T rounding), so the
    predictive-covariance cases all take the |cov| (+ jitter) route."""
    for p in SMALL:
        z = np.load(p)
        if str(z["mode"]) == "kernel":
            continue
        gi, go = z["gram_in"], z["gram_out"]
        assert not np.array_equal(gi, gi.T)
        assert np.array_equal(go, go.T)


def test_kernel_calls_match_reference():
    z = np.load(os.path.join(GOLD, "kernel_calls.npz"))
    for kind in (O.RBF, O.MATERN52, O.TANIMOTO):
        spec = O.GPSpec(kind, _t(z[f"{kind}_ls"]), 1.3, _t(z[f"{kind}_X_obs"]), _t(z[f"{kind}_S_cache"]),
                        1e-2, 0.5, _t(z[f"{kind}_alpha"]))
        x, y2 = _t(z[f"{kind}_x"]), _t(z[f"{kind}_y2"])
        y3 = y2.reshape(5, 8, -1)
        for mode in O.Kernel.MODES:
            k = O.Kernel(spec, mode)
            assert np.array_equal(k(x, y2).numpy(), z[f"{kind}_{mode}_2d"]), (kind, mode)
            assert np.array_equal(k(x, y3).numpy(), z[f"{kind}_{mode}_3d"]), (kind, mode)
            assert k(x, y3).shape == (5, x.shape[0], 8)
    with pytest.raises(ValueError):
        O.Kernel(spec, "nope")(x, y2)


def test_tanimoto_matches_reference():
    z = np.load(os.path.join(GOLD, "tanimoto.npz"))
    assert np.array_equal(O.batch_tanimoto_sim(_t(z["x1"]), _t(z["x2"])).numpy(), z["k12"])
    assert np.array_equal(O.batch_tanimoto_sim(_t(z["x1"]), _t(z["x3"])).numpy(), z["k13"])


def test_kmeans_matches_reference():
    z = np.load(os.path.join(GOLD, "kmeans.npz"))
    cl, c = O.kmeans(_t(z["a_x"].copy()), K=int(z["a_K"]))
    assert np.array_equal(cl.numpy(), z["a_cl"]) and np.array_equal(c.numpy(), z["a_c"])
    cl, c = O.kmeans(_t(z["b_x"].copy()), K=int(z["b_K"]))
    assert np.array_equal(c.numpy(), z["b_c"], equal_nan=True)
    assert np.isnan(c.numpy()).any()                      # empty cluster -> NaN centroid
    cl2, c2 = O.kmeans_chunked(_t(z["a_x"].copy()), K=int(z["a_K"]), chunk=700)
    assert np.array_equal(cl2.numpy(), z["a_cl"]) and np.array_equal(c2.numpy(), z["a_c"])


def test_kmeans_screened_shape_matches_reference():
    """The oracle's KMeans on the shape whose E step the device runs screened by default (tests/golden/kmeans_screened.npz:
    the reference's own labels and centroids at 120000 x 10 x 64, unit cube and the same pool at offset 1e4)."""
    z = np.load(os.path.join(GOLD, "kmeans_screened.npz"))
    x = np.random.default_rng(int(z["seed"])).random((int(z["N"]), int(z["d"])))
    for tag in ("unit", "offset"):
        cl, c = O.kmeans_chunked(_t(x + float(z[f"{tag}_off"])), K=int(z["K"]), chunk=8192)
        assert np.array_equal(cl.numpy().astype(np.uint8), z[f"{tag}_cl"])
        np.testing.assert_allclose(c.numpy(), z[f"{tag}_c"], rtol=1e-13)


def test_weights_match_reference():
    z = np.load(os.path.join(GOLD, "weights.npz"))
    assert float(z["eps"]) == O.EPS_WEIGHTS                # Q5: FP32 eps even in FP64
    assert np.array_equal(O.cleansing_weights(_t(z["a_in"].copy())).numpy(), z["a_out"])
    assert np.array_equal(O.cleansing_weights(_t(z["b_in"].copy())).numpy(), z["b_out"])
    torch.manual_seed(7)
    assert np.array_equal(O.deweighted_resampling(_t(z["c_in"].copy()), 40).numpy(), z["c_idx_deweighted"])
    torch.manual_seed(8)
    w = z["c_in"] / z["c_in"].sum()
    assert np.array_equal(O.weighted_resampling(_t(w), 40).numpy(), z["c_idx_weighted"])
    assert O.check_weights(_t(z["c_in"])) == bool(z["check_true"])
    assert O.check_weights(_t(np.r_[np.ones(10), np.zeros(5)])) == bool(z["check_false"])


def test_pi_and_predict_match_reference():
    """SOBER/_pi.py (PI.lfi, eta) and SOBER/_gp.py:predict through the duck model."""
    z = np.load(os.path.join(GOLD, "pi.npz"))
    for kind in (O.RBF, O.MATERN52, O.TANIMOTO):
        spec = O.GPSpec(kind, _t(z[f"{kind}_ls"]), 1.4, _t(z[f"{kind}_X_obs"]), _t(z[f"{kind}_S_cache"]),
                        1e-2, 0.25, _t(z[f"{kind}_alpha"]))
        X = _t(z[f"{kind}_X"])
        mean, var = O.predict(X, spec)
        assert np.array_equal(mean.numpy(), z[f"{kind}_mean"]) and np.array_equal(var.numpy(), z[f"{kind}_var"])
        pi = O.PI(spec)
        assert pi.eta == float(z[f"{kind}_eta"])
        assert np.array_equal(pi(X).numpy(), z[f"{kind}_lfi"])
        assert np.array_equal(pi(X, log=True).numpy(), z[f"{kind}_loglfi"])
    with pytest.raises(NotImplementedError):
        O.PI(spec, "ts")(X)
    with pytest.raises(ValueError):
        O.PI(spec, "nope")(X)


def test_wkde_matches_reference():
    """SOBER/_wkde.py: fit (seeded component selection, bandwidth, covariance) and pdf."""
    z = np.load(os.path.join(GOLD, "wkde.npz"))
    for tag in "ab":
        d = z[f"{tag}_X"].shape[1]
        torch.manual_seed(11)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            Xobs, w, cov, bw = O.wkde_fit(_t(z[f"{tag}_X"].copy()), _t(z[f"{tag}_W"].copy()), d,
                                          n_kde=int(z[f"{tag}_n_kde"]))
            assert np.array_equal(Xobs.numpy(), z[f"{tag}_Xobs"]) and np.array_equal(w.numpy(), z[f"{tag}_weights"])
            assert np.array_equal(cov.numpy(), z[f"{tag}_cov"]) and float(bw) == float(z[f"{tag}_bw"])
            bounds = torch.tensor([[0.0] * d, [1.0] * d], dtype=torch.double) if bool(z[f"{tag}_bounded"]) else None
            pdf = O.wkde_pdf(Xobs, w, cov, _t(z[f"{tag}_Xq"]), bounds)
            torch.manual_seed(5)
            smp = O.wkde_sample(Xobs, w, cov, int(z[f"{tag}_n_rec"]), bounds)
        assert np.array_equal(pdf.numpy(), z[f"{tag}_pdf"])
        assert np.array_equal(smp.numpy(), z[f"{tag}_sample"])                      # same CPU generator stream


def _basq_spec(z, tag):
    kind = str(z[f"{tag}_kind"])
    return O.GPSpec(kind, _t(z[f"{tag}_ls"]), 1.3, _t(z[f"{tag}_X_obs"]), _t(z[f"{tag}_S_cache"]), 1e-3, 0.15,
                    _t(z[f"{tag}_alpha"]))


def test_basq_gspace_matches_reference():
    """SOBER/BASQ/_scale_mmlt.py:206-275 and SOBER/BASQ/_basq.py:43-81 (row f4): g-space kernel (2-D and 3-D
    second argument), g-space prediction, and the quadrature built on recombination with that kernel."""
    z = np.load(os.path.join(GOLD, "basq.npz"))
    for tag in "ab":
        spec = _basq_spec(z, tag)
        Xc = _t(z[f"{tag}_X_cand"])
        d = Xc.shape[1]
        assert np.array_equal(O.gspace_kernel(Xc[:10], Xc[100:150], spec).numpy(), z[f"{tag}_K2"])
        assert np.array_equal(O.gspace_kernel(Xc[:10], Xc[200:260].reshape(3, 20, d), spec).numpy(), z[f"{tag}_K3"])
        mug, varg = O.gspace_predict(Xc[:100], spec)
        assert np.array_equal(mug.numpy(), z[f"{tag}_mug"]) and np.array_equal(varg.numpy(), z[f"{tag}_varg"])
        torch.manual_seed(SEED_CALL)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            idx, w, ELML, AVLML = O.basq_quadrature(Xc, int(z[f"{tag}_M"]), int(z[f"{tag}_b"]), spec, -3.25)
        assert np.array_equal(idx.numpy(), z[f"{tag}_idx"])
        assert np.array_equal(w.numpy(), z[f"{tag}_w"])
        assert ELML == float(z[f"{tag}_ELML"]) and AVLML == float(z[f"{tag}_AVLML"])


def test_adaptive_pruning_matches_reference():
    """Dataset path (SOBER/_sampler.py:325-349): oracle and the product's torch-level mirror vs the reference's
    own outputs on every branch (weights are distinct, so the sort order is unique)."""
    import sober_amd
    z = np.load(os.path.join(GOLD, "pruning.npz"))
    tags = sorted({k[:-4] for k in z.files if k.endswith("_idx")})
    assert len(tags) == 4
    for tag in tags:
        w = torch.from_numpy(z[f"{tag}_w"])
        n_rec, n_nys = (int(v) for v in z[f"{tag}_args"])
        assert np.array_equal(O.adaptive_pruning(w, n_rec, n_nys).numpy(), z[f"{tag}_idx"]), tag
        assert np.array_equal(sober_amd.adaptive_pruning(w, n_rec, n_nys).numpy(), z[f"{tag}_idx"]), tag


def test_d1_goldens_and_exact_moments():
    """d = 1 (make_golden.py::gen_d1_sensitivity): the oracle reproduces the reference on both one-dimensional fixtures,
    and on the pool without leftovers (N = 1024) the reduced measure integrates the step's own test functions exactly
    -- the property the device result is held to on this ill-conditioned input (tests/test_hip_parity.py)."""
    from tests.golden import make_golden as MG
    from tests.golden.synth import synth, build_spec
    for case, name in ((MG.D1_CASE, "d1_sensitivity.npz"), (MG.D1_EXACT_CASE, "d1_exact.npz")):
        z = np.load(os.path.join(os.path.dirname(SMALL[0]), name))
        inp = synth(case)
        spec = build_spec(case, inp)
        tr = {}
        torch.manual_seed(SEED_CALL)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            idx, w = O.recombination(_t(inp["X_cand"]), _t(inp["X_nys"]), case["b"], O.Kernel(spec, case["mode"]),
                                     init_weights=_t(inp["mu0"].copy()), trace=tr)
        assert np.array_equal(idx.numpy(), z["idx"]) and np.array_equal(w.numpy(), z["w"])
        if case["N"] == 1024:
            K = O.Kernel(spec, case["mode"])(_t(inp["X_nys"]), _t(inp["X_cand"])).numpy()
            U = tr["U"].numpy()
            m = U @ (K @ inp["mu0"])
            assert np.linalg.norm(U @ (K[:, z["idx"]] @ z["w"]) - m) < 1e-12 * np.linalg.norm(m)
