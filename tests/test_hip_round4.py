"""GPU tests added in round 4 (run with -m gpu on an MI355X), all through the C ABI:
  * a fixed-seed slice of the randomised device-vs-oracle sweep (tests/tools/fuzz_parity.py) with its one-ulp classifier, so
    that the driver repeats what used to be evidence under profiles/ only;
  * the far side of the two size limits of the device kernels (N_nys > 536: host Nystrom route; batch > 224: host
    Caratheodory route) against the oracle, asserting WHICH route ran and that it says so once;
  * the two advisor findings of round 3 (a converted pool copy is never served from the packed-pool cache; a non-square
    covar_cache root takes the Python route of the row table);
  * `bench.py --gpus 2` end to end on the one GPU (gloo between the ranks; the direct-peer all-reduce forced in a second
    run): the N > 1 branch of the bench must not meet its first run on an 8-GPU node."""
import json
import os
import subprocess
import sys
import warnings

import numpy as np
import pytest
import torch

import sober_amd
from oracle import sober_oracle as O
from tests.golden.synth import SEED_CALL, build_spec, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "the -m gpu tests need the MI355X"
    from sober_amd import _native
    _native.load()
    return torch.device("cuda:0")


def kspec(spec):
    return sober_amd.KernelSpec(spec.kind, spec.lengthscale, spec.outputscale, spec.X_obs, spec.S_cache,
                                spec.noise, spec.mean_const, spec.alpha)


# --------------------------------------------------------------------------- #
# the randomised sweep, a fixed slice of it
# --------------------------------------------------------------------------- #
def test_fuzz_slice_vs_oracle(dev):
    """30 random continuous-kernel recombinations (batch 5-120: every size instantiation of the one-CU Caratheodory kernels
    and the multi-CU ones, leftovers of every kind, three modes, calc_obj in a quarter) and 10 fingerprint ones, device
    against oracle on the same seeded inputs.  Bar per case: identical indices and weights within 1e-6, or the case is
    ill-posed IN THE REFERENCE (its own indices change / its own weights move by > 1e-5 when its inputs move by one ulp --
    the oracle is held to the reference on 64 random cases by tests/test_oracle_vs_reference_fuzz.py).  At most a tenth
    of the slice may be ill-posed (observed over the 770-case sweep: 4 %): the classifier is not a way out."""
    from tests.tools import fuzz_parity as F
    bad, ill, lines = [], 0, []
    for tani, n, seed in ((False, 30, 2026), (True, 10, 2027)):
        rng = np.random.default_rng(seed)
        for i in range(n):
            c = F.make_case(rng, tani, batches=[5, 8, 9, 16, 17, 24, 32, 33, 40, 56, 57, 64, 65, 80, 100, 112, 120], n_factor=25)
            ok, verdict, same, relw = F.check_case(c, dev)
            lines.append("%s: %s idx_equal=%s max rel w %.1e" % (F.describe(i, c), verdict, same, relw))
            ill += verdict.startswith("ill-posed")
            if not ok:
                bad.append(lines[-1])
    print("\n".join(lines))
    assert not bad, bad
    assert ill <= 4, (ill, [ln for ln in lines if "ill-posed" in ln])


# --------------------------------------------------------------------------- #
# beyond the compiled size set: the host routes, named
# --------------------------------------------------------------------------- #
def _vs_oracle(case, dev, timers, rtol):
    inp = synth(case)
    spec = build_spec(case, inp)
    mu_ref = _t(inp["mu0"].copy())
    torch.manual_seed(SEED_CALL)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        idx_ref, w_ref = O.recombination(_t(inp["X_cand"]), _t(inp["X_nys"]), case["b"], O.Kernel(spec, case["mode"]), init_weights=mu_ref)
    mu = _t(inp["mu0"].copy()).to(dev)
    from sober_amd._ops_hip import HipOps
    ops = HipOps(dev)                                        # (a fresh backend: the size warning is given once per backend)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        torch.manual_seed(SEED_CALL)
        idx, w = sober_amd.recombination(_t(inp["X_cand"]).to(dev), _t(inp["X_nys"]).to(dev), case["b"],
                                         sober_amd.Kernel(kspec(spec), case["mode"]), init_weights=mu, _timers=timers, _ops=ops)
        torch.manual_seed(SEED_CALL)                         # a second step on the same backend: no second warning
        mu2 = _t(inp["mu0"].copy()).to(dev)
        sober_amd.recombination(_t(inp["X_cand"]).to(dev), _t(inp["X_nys"]).to(dev), case["b"],
                                sober_amd.Kernel(kspec(spec), case["mode"]), init_weights=mu2, _ops=ops)
    assert np.array_equal(idx.cpu().numpy(), idx_ref.numpy())
    np.testing.assert_allclose(w.cpu().numpy(), w_ref.numpy(), rtol=rtol)
    np.testing.assert_allclose(mu.cpu().numpy(), mu_ref.numpy(), rtol=rtol, atol=0)
    return [str(r.message) for r in rec if issubclass(r.category, RuntimeWarning) and "sober_amd:" in str(r.message)]


def test_nystrom_beyond_the_device_route_goes_to_the_host_and_says_so(dev):
    """N_nys = 2100 > 2048 (csrc/chol.hip: CB_MAXN; 1024 until round 6): make_cov_psd + svd_lowrank on host LAPACK (SOBER/_rchq.py:34-39 takes any
    N_nys), everything else on the device; ONE warning naming the limit."""
    timers = {}
    msgs = _vs_oracle(dict(kind=O.RBF, mode="predictive_covariance", N=8000, M=2100, d=5, b=40, n_obs=60, seed=41, ard=True),
                      dev, timers, 1e-7)
    assert "nystrom_host" in timers and "car_host" not in timers, timers
    assert len(msgs) == 1 and "N_nys = 2100" in msgs[0] and "2048" in msgs[0], msgs


@pytest.mark.parametrize("M,b", [(600, 40), (1000, 100), (1500, 64), (2048, 100)])
def test_nystrom_between_536_and_2048_stays_on_the_device(M, b, dev):
    """536 < N_nys <= 2048 (1024 until round 6): the jitter ladder's probes go panel by panel (two launches per 32 columns, every rung in the same
    launches: csrc/chol.hip k_cb_diag / k_cb_update), the rest of the chain is the one N_nys <= 536 takes; against the
    oracle (make_cov_psd + svd_lowrank of SOBER/_rchq.py:34-39 on host LAPACK): identical points, weights to 1e-7 (1e-6 beyond
    1024 points: at 2048 one weight of 2e-5 sits 8e-7 -- 2e-11 absolute -- from the oracle's)."""
    timers = {}
    msgs = _vs_oracle(dict(kind=O.RBF, mode="predictive_covariance", N=20000, M=M, d=6, b=b, n_obs=80, seed=43 + M, ard=True),
                      dev, timers, 1e-7 if M <= 1024 else 1e-6)
    assert "nystrom_device" in timers and "nystrom_host" not in timers and "car_host" not in timers, timers
    assert msgs == [], msgs


@pytest.mark.parametrize("n", [537, 600, 777, 1000, 1024])
def test_panelwise_ladder_probes_agree_with_lapack(n, dev):
    """sober_cholesky_probe_batched on an 11-rung ladder whose lower rungs are NOT positive definite: info == 0 exactly on
    the rungs numpy's Cholesky accepts, a failing rung's info = the order of the first leading minor that is not positive
    definite (dpotrf's convention), min pivot^(1/2) of an accepted rung = min diag(L) of numpy's factor."""
    from sober_amd import _native as nat
    lib = nat.load()
    rng = np.random.default_rng(n)
    B = rng.standard_normal((n, n // 3))
    A = B @ B.T / n                                            # rank n / 3: singular, PD only from a shift on
    A += 1e-3 * np.diag(rng.random(n))
    lam = np.linalg.eigvalsh(A)
    shifts = np.concatenate([[-(lam[0] + 2e-3), -(lam[0] + 5e-4), -lam[0] * 0.5], 1e-5 * (2.0 ** np.arange(8) - 1)])
    n_r = len(shifts)
    Ad, sh = _t(A).to(dev), _t(shifts).to(dev)
    work = torch.empty(n_r * n * n, dtype=torch.float64, device=dev)
    info = torch.full((n_r,), -99, dtype=torch.int32, device=dev)
    piv = torch.zeros(n_r, dtype=torch.float64, device=dev)
    ws = torch.empty(n_r * 8192, dtype=torch.uint8, device=dev)
    nat._check(lib.sober_cholesky_probe_batched(Ad.data_ptr(), n, n, sh.data_ptr(), n_r, work.data_ptr(), info.data_ptr(),
                                                piv.data_ptr(), ws.data_ptr(), ws.numel(), nat._stream(Ad)), "probe_batched")
    info, piv = info.cpu().numpy(), piv.cpu().numpy()
    for r, s_ in enumerate(shifts):
        As = A + s_ * np.eye(n)
        try:
            L = np.linalg.cholesky(As)
            ok = True
        except np.linalg.LinAlgError:
            ok = False
        if ok:
            assert info[r] == 0, (r, s_, info[r])
            np.testing.assert_allclose(np.sqrt(piv[r]), np.diag(L).min(), rtol=1e-6)
        else:
            lead = next(k for k in range(1, n + 1) if np.linalg.eigvalsh(As[:k, :k])[0] <= 0)
            # (the first non-positive pivot in floating point can come a little after the first minor that is singular
            #  in exact arithmetic; never before it)
            assert info[r] >= lead and info[r] <= min(n, lead + 40), (r, s_, info[r], lead)
            assert piv[r] <= 0.0 or not np.isfinite(piv[r]), (r, piv[r])


@pytest.mark.parametrize("b,M,N,kind,d", [(250, 400, 2600, O.RBF, 6), (300, 500, 5000, O.RBF, 6), (540, 900, 6000, O.MATERN52, 20),
                                          (1000, 1500, 4300, O.MATERN52, 20)])
def test_batch_beyond_the_register_resident_kernels_stays_on_the_device(b, M, N, kind, d, dev):
    """batch = 250 / 300 / 540 / 1000: a 500- ... 2000-point Caratheodory step is beyond csrc/car_mc.hip (N <= 448) -- host LAPACK +
    the C++ pivots for every level until round 5, the memory-resident kernels of csrc/car_big.hip since round 6
    (SOBER/_rchq.py:224-270 takes any batch): no host step, no warning about the step.  The device Nystrom route takes ranks up
    to 1072 since the same round (256 before: one workgroup's Cholesky panel holds 536 columns, a wider block is orthonormalised
    in two halves -- block Gram-Schmidt with the same kernels, csrc/nystrom_exec.cpp) and N_nys up to 2048: no host phase at all.
    (Batches 540 and 1000 on a Matern kernel in 20 dimensions: 539 Nystrom functions of a smooth low-dimensional kernel are
    rounding noise beyond the first ~200 -- in the reference as much as here -- and the step's weights then move by 1e-3 with the
    last bits of anything.)"""
    timers = {}
    # (batch 1000: the same 1000 points; one weight of 7e-7 sits 2.5e-6 -- 1.7e-12 absolute -- from the oracle's: 1e-5 there)
    msgs = _vs_oracle(dict(kind=kind, mode="predictive_covariance", N=N, M=M, d=d, b=b, n_obs=50, seed=42, ard=False),
                      dev, timers, 1e-6 if b < 1000 else 1e-5)
    assert "car_host" not in timers and "nystrom_host" not in timers, timers
    assert msgs == [], msgs


def test_batch_beyond_every_device_kernel_goes_to_host_lapack_and_says_so(dev):
    """batch = 1030: a 2060-point step is beyond csrc/car_big.hip too (N <= 2048), N_nys = 2100 beyond the device Nystrom route:
    host LAPACK + the C++ pivots for every level, the literal Nystrom route with them; ONE warning per phase.  (The host route
    against the oracle is what every batch beyond 224 ran until round 5 and what test_host_and_device_car_agree still holds; at
    this size the oracle's own pivot loop takes a minute, so the result is held to the invariants of a recombination here.)"""
    from sober_amd._ops_hip import HipOps
    case = dict(kind=O.RBF, mode="kernel", N=4400, M=2100, d=6, b=1030, n_obs=0, seed=42, ard=False)
    inp = synth(case)
    spec = build_spec(case, inp)
    mu = _t(inp["mu0"].copy()).to(dev)
    timers, ops = {}, HipOps(dev)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        torch.manual_seed(SEED_CALL)
        idx, w = sober_amd.recombination(_t(inp["X_cand"]).to(dev), _t(inp["X_nys"]).to(dev), case["b"],
                                         sober_amd.Kernel(kspec(spec), case["mode"]), init_weights=mu, _timers=timers, _ops=ops)
    msgs = [str(r.message) for r in rec if issubclass(r.category, RuntimeWarning) and "sober_amd:" in str(r.message)]
    assert "car_host" in timers and "nystrom_host" in timers, timers
    assert sum("batch = 1030" in m and "2048" in m for m in msgs) == 1, msgs
    assert len(msgs) == 1, msgs                              # (the literal Nystrom route goes with the host steps: no second cliff to name)
    i, wn, mun = idx.cpu().numpy(), w.cpu().numpy(), mu.cpu().numpy()
    assert 0 < len(i) <= 1030 and (np.diff(i) > 0).all() and (wn > 0).all() and abs(wn.sum() - 1.0) < 1e-12
    assert np.array_equal(np.flatnonzero(mun), i) and np.array_equal(mun[i], wn)


# --------------------------------------------------------------------------- #
# advisor findings of round 3
# --------------------------------------------------------------------------- #
def test_converted_pool_copy_is_never_served_from_the_cache(dev):
    """A fingerprint pool that is NOT float64-on-the-device (here: float32 on the host) reaches the backend as a converted
    copy whose memory is freed after the call; the allocator hands the same block to the next call's copy, so pointer,
    layout and a fresh version counter all match -- the packed-pool / pool-mean caches must not answer for it.  The pool
    is modified IN PLACE between two calls; the second result must be the oracle's on the modified pool."""
    case = dict(kind=O.TANIMOTO, mode="weighted_predictive_covariance", N=3000, M=64, d=256, b=10, n_obs=25, seed=77,
                mean_const=0.3, bit_p=0.1)
    inp = synth(case)
    spec = build_spec(case, inp)
    from sober_amd._ops_hip import HipOps
    ops = HipOps(dev)
    X32 = _t(inp["X_cand"]).to(torch.float32)                # the caller's pool: float32, host memory
    Xn = _t(inp["X_nys"]).to(dev)

    def both(X32_now):
        mu_ref = _t(inp["mu0"].copy())
        torch.manual_seed(SEED_CALL)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            i_ref, w_ref = O.recombination(X32_now.to(torch.float64), _t(inp["X_nys"]), case["b"], O.Kernel(spec, case["mode"]),
                                           init_weights=mu_ref)
            mu = _t(inp["mu0"].copy()).to(dev)
            torch.manual_seed(SEED_CALL)
            i, w = sober_amd.recombination(X32_now, Xn, case["b"], sober_amd.Kernel(kspec(spec), case["mode"]), init_weights=mu, _ops=ops)
        return i_ref.numpy(), w_ref.numpy(), i.cpu().numpy(), w.cpu().numpy()

    a = both(X32)
    assert np.array_equal(a[0], a[2]) and np.allclose(a[1], a[3], rtol=1e-7)
    rng = np.random.default_rng(3)
    flip = rng.random(X32.shape) < 0.2
    X32[_t(flip)] = 1.0 - X32[_t(flip)]                      # in place: same object, same memory
    b_ = both(X32)
    assert not np.array_equal(a[0], b_[0])                   # (the modification matters: the reference selects other points)
    assert np.array_equal(b_[0], b_[2]) and np.allclose(b_[1], b_[3], rtol=1e-7)


def test_non_square_covar_cache_root_takes_the_python_route(dev):
    """gpytorch's covar_cache is n_obs x k (a Lanczos root) once n_obs exceeds max_cholesky_size; W = S S^T (SOBER/_gp.py:277)
    is still n_obs x n_obs.  sober_plan_rows assumes a square root: such a spec must build its row table through the
    Python route (woodbury) -- same result as the oracle with the same root."""
    case = dict(kind=O.RBF, mode="predictive_covariance", N=2500, M=64, d=3, b=10, n_obs=40, seed=91)
    inp = synth(case)
    spec = build_spec(case, inp)
    k = 24
    S_k = spec.S_cache[:, :k].contiguous()                   # a rank-k root: W_k = S_k S_k^T
    import dataclasses
    spec_k = dataclasses.replace(spec, S_cache=S_k)
    mu_ref = _t(inp["mu0"].copy())
    torch.manual_seed(SEED_CALL)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        i_ref, w_ref = O.recombination(_t(inp["X_cand"]), _t(inp["X_nys"]), case["b"], O.Kernel(spec_k, case["mode"]), init_weights=mu_ref)
        mu = _t(inp["mu0"].copy()).to(dev)
        torch.manual_seed(SEED_CALL)
        i, w = sober_amd.recombination(_t(inp["X_cand"]).to(dev), _t(inp["X_nys"]).to(dev), case["b"],
                                       sober_amd.Kernel(kspec(spec_k), case["mode"]), init_weights=mu)
    assert np.array_equal(i.cpu().numpy(), i_ref.numpy())
    np.testing.assert_allclose(w.cpu().numpy(), w_ref.numpy(), rtol=1e-7)


# --------------------------------------------------------------------------- #
# bench.py with two ranks on the one GPU
# --------------------------------------------------------------------------- #
def _bench_two_ranks(extra_env, port):
    env = dict(os.environ, SOBER_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "2", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--check-unsharded"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


@pytest.mark.parametrize("peer", [False, True], ids=["group_allreduce", "direct_peer_forced"])
def test_bench_two_ranks_one_gpu(peer, dev):
    """`bench.py --gpus 2 --config 2` as the driver launches it (torch.distributed.run), two ranks sharing the one GPU over
    gloo: the JSON line of rank 0 says two GPUs-worth of ranks, the sharded route, twice the pool -- and the sharded result
    IS the unsharded one (rank 0 repeats the step on the gathered pool: identical indices on every rank, weights 1e-9).
    Second run: the direct-peer all-reduce forced over the non-nccl group (IPC-mapped regions on the one GPU)."""
    d = _bench_two_ranks({"SOBER_PEER_ALLREDUCE": "force"} if peer else {"SOBER_PEER_ALLREDUCE": "0"}, 29671 + int(peer))
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak"
    assert "row-sharded x2" in d["config"]["parallelism"]
    if peer:
        assert "direct-peer" in d["config"]["parallelism"], d["config"]["parallelism"]
    assert "200000 in all" in d["config"]["workload"]
    chk = d["parity_sharded_vs_unsharded"]
    assert chk["idx_equal_unsharded"] and chk["ranks_agree"], chk
    assert chk["max_rel_w_vs_unsharded"] < 1e-9, chk
    assert d["n_selected"] <= 100 and d["value"] > 0 and d["roofline"] is not None


def _lloyd(lib, nat, Xd, K, dev, nbytes, iters=10):
    N, d = Xd.shape
    c = torch.empty(K, d, dtype=torch.float64, device=dev)
    cl = torch.empty(N, dtype=torch.int32, device=dev)
    ws = torch.zeros(max(nbytes, 8), dtype=torch.uint8, device=dev)
    nat._check(lib.sober_kmeans_lloyd(Xd.data_ptr(), N, d, K, iters, c.data_ptr(), cl.data_ptr(),
                                      ws.data_ptr() if nbytes else None, nbytes, nat._stream(Xd)), "kmeans")
    off = int(lib.sober_kmeans_stat_offset(N, d, K)) if nbytes else -1
    listed = int(ws[off:off + 4].view(torch.int32).item()) if off >= 0 and nbytes > off else None
    return cl.cpu().numpy(), c.cpu().numpy(), listed


@pytest.mark.parametrize("N,d,K", [(100000, 10, 500), (40000, 20, 500), (30011, 3, 77), (20000, 9, 512), (20000, 31, 64),
                                   (16384, 1, 33), (50000, 20, 17), (20000, 4, 1), (20000, 5, 2), (300000, 8, 880)])
def test_kmeans_screened_e_step_equals_the_exact_kernel(N, d, K, dev):
    """The E step screened on the BF16 matrix cores (csrc/kmeans.hip: two-piece splits, FP32 accumulation, a margin, the
    FP64 kernel on whatever the margin does not decide) against the (x - c)^2 kernel that runs without a workspace: labels
    bit-equal after ten iterations -- uniform pools, a pool far from the origin (the centring), tiny and huge scales
    (everything goes to the list), a lattice (exact ties everywhere), tight blobs (near-ties at the margin's scale),
    duplicated initial centroids, an empty cluster (NaN centroid), NaN and Inf coordinates."""
    from sober_amd import _native as nat
    lib = nat.load()
    assert int(lib.sober_kmeans_stat_offset(N, d, K)) >= 0, "this shape is meant to take the screened path"
    rng = np.random.default_rng(7 * N + d)
    X = rng.random((N, d))
    variants = {"uniform": X, "offset 1e4": X + 1e4, "offset -3e7, scale 5": 5.0 * X - 3e7, "scale 1e-14": 1e-14 * X,
                "scale 1e18": 1e18 * X, "scale 1e160": 1e160 * X}
    Xd2 = X.copy(); Xd2[1] = Xd2[0]; Xd2[K + 5] = Xd2[7]      # two identical initial centroids: the second one's cluster
    variants["duplicates"] = Xd2                               # is empty after one step (NaN centroid: everything listed)
    variants["lattice"] = np.floor(4.0 * X)                   # integer coordinates 0 .. 3: ties between many centroids
    blobs = rng.random((K, d))[rng.integers(0, K, N)] + 1e-4 * rng.standard_normal((N, d))
    variants["blobs 1e-4"] = blobs
    Xe = X.copy(); Xe[:K] = 2.0 + np.arange(K)[:, None] * 3.0; Xe[K:] = Xe[0] + 1e-3 * rng.random((N - K, d))
    variants["empty clusters"] = Xe
    Xn = X.copy(); Xn[N // 2, d // 2] = np.nan; Xn[N // 3, 0] = np.inf
    variants["nan and inf"] = Xn
    full = int(lib.sober_kmeans_ws_bytes_screened(N, d, K))   # (also below the pool size from which the screen pays)
    assert full >= int(lib.sober_kmeans_ws_bytes(N, d, K))
    fractions = {}
    for name, Xv in variants.items():
        Xd = _t(Xv).to(dev)
        cl_s, c_s, listed = _lloyd(lib, nat, Xd, K, dev, full)
        cl_e, c_e, _ = _lloyd(lib, nat, Xd, K, dev, 0)
        assert listed is not None
        fractions[name] = listed / (10.0 * N)
        assert np.array_equal(cl_s, cl_e), "%s: %d labels differ" % (name, int((cl_s != cl_e).sum()))
        np.testing.assert_allclose(c_s, c_e, rtol=1e-12, equal_nan=True)
    # the screen decides nearly everything on ordinary pools (what makes it worth running), and nothing where it must not
    if d > 1:                                                 # (d = 1: the gaps ARE small against |x|^2 + |c|^2 for many points)
        assert fractions["uniform"] < 0.05 and fractions["offset 1e4"] < 0.05, fractions
    assert fractions["scale 1e-14"] == 1.0 and fractions["scale 1e160"] == 1.0, fractions
