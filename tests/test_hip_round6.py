"""GPU tests added in round 6 (run with -m gpu on an MI355X), all through the C ABI."""
import json
import os
import subprocess
import sys

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


def _json_line(stdout):
    return json.loads([ln for ln in stdout.strip().splitlines() if ln.startswith("{")][-1])


def test_bench_gpus_2_without_a_launcher_starts_its_own_ranks(dev):
    """`python bench.py --gpus 2` with NO torch.distributed.run in front and no WORLD_SIZE in the environment -- the form the
    driver uses for N = 1 -- used to die on `assert world == args.gpus`.  It now starts the two ranks as a child
    torch.distributed.run (gloo hook: both ranks on the one GPU), forwards rank 0's JSON line and returns 0."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(SOBER_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", SOBER_PEER_ALLREDUCE="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "1", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--check-unsharded"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _json_line(out.stdout)
    assert d["n_gpus"] == 2 and d["steps"] == 2 and "row-sharded x2" in d["config"]["parallelism"]
    chk = d["parity_sharded_vs_unsharded"]
    assert chk["idx_equal_unsharded"] and chk["ranks_agree"], chk


def test_bench_default_line_carries_the_other_configs_and_the_acquisition_step(dev):
    """The ONE command the driver runs (`python bench.py`, here with few steps and without the CPU leg) prints ONE JSON line
    whose `other_configs` holds BASELINE.json configurations 1, 3, 4 and 5 -- each with ms_per_step, the dominant kernel's
    roofline fraction and a parity verdict -- and whose `acquisition_step` is `Sober.next_batch` at configuration 2's shapes."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-sweep",
           "--other-steps", "3"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["parity"]["idx_equal_reference"] and d["parity"]["max_rel_w_vs_reference"] < 1e-7
    assert sorted(d["other_configs"]) == ["1", "3", "4", "5"]
    for c, r in d["other_configs"].items():
        assert "error" not in r, (c, r)
        assert r["ms_per_step"] > 0 and 0 < r["roofline"]["frac"] < 1, (c, r)
        p = r["parity"]
        if "idx_equal_reference" in p:
            assert p["idx_equal_reference"] and p["max_rel_w_vs_reference"] < 1e-7, (c, p)
        else:
            assert p["by"] == "invariants" and all(v for k, v in p.items() if isinstance(v, bool)), (c, p)
            assert p["mass_error"] < 1e-12, (c, p)
    x = d["beyond_baseline"]["7"]                             # (batch 250: every Caratheodory step on csrc/car_big.hip)
    assert "error" not in x, x
    assert 0 < x["ms_per_step"] < 200 and x["n_selected"] == 250 and x["parity"]["by"] == "invariants", x
    assert all(v for k, v in x["parity"].items() if isinstance(v, bool)) and x["parity"]["mass_error"] < 1e-12, x
    a = d["acquisition_step"]
    assert "error" not in a, a
    assert a["ms_per_step"] > 0 and a["parity"]["reference_fixture_sober_next_batch_equal"] is True, a


# --------------------------------------------------------------------------- #
# levels derived from level 0's class sums (csrc/level_class.hip)
# --------------------------------------------------------------------------- #
def _kspec(spec):
    return sober_amd.KernelSpec(spec.kind, spec.lengthscale, spec.outputscale, spec.X_obs, spec.S_cache, spec.noise,
                                spec.mean_const, spec.alpha)


def _both_ways(case, dev):
    """The same step with the levels derived from class sums (default) and with every level's sums evaluated."""
    import warnings
    from sober_amd._ops_hip import HipOps
    from tests.golden.synth import SEED_CALL, build_spec, synth
    inp = synth(case)
    spec = build_spec(case, inp)
    out = []
    for classes in (True, False):
        ops = HipOps(dev)
        ops.level_classes = classes
        mu = _t(inp["mu0"].copy()).to(dev)
        torch.manual_seed(SEED_CALL)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            i, w = sober_amd.recombination(_t(inp["X_cand"]).to(dev), _t(inp["X_nys"]).to(dev), case["b"],
                                           sober_amd.Kernel(_kspec(spec), case["mode"]), init_weights=mu, _ops=ops)
        out.append((i.cpu().numpy(), w.cpu().numpy(), mu.cpu().numpy(), dict(ops.last_levels)))
    return inp, spec, out


@pytest.mark.parametrize("case,depth", [
    (dict(kind="rbf", mode="predictive_covariance", N=100000, M=500, d=10, b=100, n_obs=200, seed=0), 2),      # BASELINE configs[1]
    (dict(kind="rbf", mode="predictive_covariance", N=2000, M=100, d=2, b=10, n_obs=30, seed=0, ard=True), 2),  # configs[0]
    (dict(kind="matern52", mode="predictive_covariance", N=40 * 8 * 30, M=96, d=6, b=20, n_obs=40, seed=3), 4),
    (dict(kind="rbf", mode="weighted_predictive_covariance", N=64 * 8 * 7, M=120, d=20, b=32, n_obs=50, seed=4, mean_const=0.1), 3),
    (dict(kind="rbf", mode="predictive_covariance", N=200 * 1024, M=300, d=5, b=100, n_obs=60, seed=6, zero_frac=0.0), 4),
    (dict(kind="tanimoto", mode="weighted_predictive_covariance", N=100 * 2 * 21, M=300, d=512, b=50, n_obs=100, seed=12, bit_p=0.06, mean_const=0.3), 1),
    (dict(kind="tanimoto", mode="predictive_covariance", N=200 * 2 * 125, M=500, d=1024, b=100, n_obs=200, seed=13, bit_p=0.04), 1),
    (dict(kind="rbf", mode="predictive_covariance", N=60 * 2 * 9, M=80, d=3, b=30, n_obs=20, seed=9), 1),
    (dict(kind="rbf", mode="predictive_covariance", N=60 * 2 * 9 + 7, M=80, d=3, b=30, n_obs=20, seed=9), 0),   # leftovers: evaluated
    (dict(kind="rbf", mode="predictive_covariance", N=60 * 9, M=80, d=3, b=30, n_obs=20, seed=9), 0),           # odd element count
], ids=["cfg2", "cfg1", "matern_d4", "weighted_d3", "rbf_205k_d4", "tanimoto_small", "tanimoto_50k", "depth1", "leftovers", "odd"])
def test_levels_derived_from_class_sums_equal_the_evaluated_ones(case, depth, dev):
    """SOBER/_rchq.py:116-126 on levels 1 .. D is a gather-and-scale of level 0's sums by element class e mod 2^D (:198-221:
    element-major compaction) -- the default path -- against the same step with every level's kernel evaluated
    (`HipOps.level_classes = False`): identical indices, weights and the mutated init_weights to 1e-9; the executor reports the
    depth it used; and both against the oracle."""
    import warnings
    from oracle import sober_oracle as O
    from tests.golden.synth import SEED_CALL
    inp, spec, ((i1, w1, m1, lv1), (i0, w0, m0, lv0)) = _both_ways(case, dev)
    # (depth: what the pool's size allows -- R % (2^D S) == 0 with two elements left -- capped by the executor where the class
    #  launch's partial slots x classes would not fit its buffers: sober_level_class_depth)
    assert (lv1["derived"] == depth or (depth > 2 and 2 <= lv1["derived"] < depth)) and lv0["derived"] == 0, (lv1, lv0)
    assert lv1["R"] == lv0["R"]
    assert np.array_equal(i1, i0)
    np.testing.assert_allclose(w1, w0, rtol=1e-9)
    np.testing.assert_allclose(m1, m0, rtol=1e-9, atol=0)
    if case["N"] <= 30000:
        mu_ref = _t(inp["mu0"].copy())
        torch.manual_seed(SEED_CALL)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            i_ref, w_ref = O.recombination(_t(inp["X_cand"]), _t(inp["X_nys"]), case["b"], O.Kernel(spec, case["mode"]),
                                           init_weights=mu_ref)
        assert np.array_equal(i1, i_ref.numpy())
        np.testing.assert_allclose(w1, w_ref.numpy(), rtol=1e-7)


def test_class_chain_stops_unless_exactly_b_sets_survive(dev):
    """sober_level_update_queued_cls: the successor's sums can only be derived when exactly b of the 2b sets survived and the
    level had no leftovers -- otherwise *dR_next = -1 with the weights and the list untouched (the synchronised loop redoes
    the level); with b survivors it writes the rank -> set table and the factors w*_k / tot_k the derive kernel reads."""
    from sober_amd import _native as nat
    lib = nat.load()
    S, b, E = 8, 4, 6
    R = S * E
    i32, f64 = torch.int32, torch.float64
    idx = torch.arange(R, dtype=i32, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def run(keep, r_extra=0):
        kr = torch.full((S + 1,), -1, dtype=i32, device=dev)
        kept = [s for s in range(S) if keep[s]]
        for k, s in enumerate(kept):
            kr[s] = k
        kr[S] = len(kept)
        w_star = torch.arange(1, S + 1, dtype=f64, device=dev) * 0.125
        tot = torch.arange(2, S + 2, dtype=f64, device=dev)
        mu = torch.ones(R + r_extra, dtype=f64, device=dev)
        idx_ = torch.arange(R + r_extra, dtype=i32, device=dev)
        new = torch.full((R + r_extra,), -7, dtype=i32, device=dev)
        dR = torch.tensor([R + r_extra, 123], dtype=torch.int64, device=dev)
        scale, sof = torch.full((S,), -1.0, dtype=f64, device=dev), torch.full((S,), -1, dtype=i32, device=dev)
        nat._check(lib.sober_level_update_queued_cls(idx_.data_ptr(), R + r_extra, S, kr.data_ptr(), w_star.data_ptr(), tot.data_ptr(),
                                                     mu.data_ptr(), new.data_ptr(), dR.data_ptr(), dR[1:].data_ptr(), R, b,
                                                     scale.data_ptr(), sof.data_ptr(), st), "update_cls")
        torch.cuda.synchronize()
        return int(dR[1].item()), mu.cpu().numpy(), new.cpu().numpy(), scale.cpu().numpy(), sof.cpu().numpy(), kept

    nxt, mu, new, scale, sof, kept = run([1, 0, 1, 0, 0, 1, 1, 0])
    assert nxt == E * b and sof[:b].tolist() == kept
    np.testing.assert_array_equal(scale[:b], [(k + 1) * 0.125 / (s + 2) for k, s in enumerate(kept)])
    assert sorted(new[:E * b].tolist()) == sorted(e * S + s for e in range(E) for s in kept)
    nxt, mu, new, *_ = run([1, 0, 1, 0, 0, 1, 0, 0])                  # three survivors
    assert nxt == -1 and (mu == 1.0).all() and (new == -7).all()
    nxt, mu, new, *_ = run([1, 0, 1, 0, 0, 1, 1, 0], r_extra=3)       # leftovers
    assert nxt == -1 and (mu == 1.0).all() and (new == -7).all()


# --------------------------------------------------------------------------- #
# acquisition-guided branch: the second elimination's direction in one launch (csrc/null_vector.hip)
# --------------------------------------------------------------------------- #
@pytest.mark.parametrize("N,nfun", [(200, 99), (200, 100), (60, 29), (40, 12), (24, 11), (208, 103), (130, 64), (224, 110), (16, 7),
                                   (30, 1)])
def test_null_vector_kernel_on_the_survivors_matrix(N, nfun, dev):
    """sober_null_vector on the survivors of a Caratheodory step with the objective as one more function (X = [features |
    objective], N x (nfun + 1)): the vector it returns by set annihilates A2 = [features of the survivors; 1]
    (SOBER/_rchq.py:88-91: the reference takes that vector from an SVD of this very matrix) -- residual at rounding level,
    the angle to numpy's SVD null vector below 1e-12 --, is zero on cancelled sets; and the status word says when the first
    step did not leave nfun + 2 sets."""
    from sober_amd import _native as nat
    rng = np.random.default_rng(100 * N + nfun)
    X = rng.standard_normal((N, nfun + 1)) * np.exp(-0.02 * np.arange(nfun + 1))[None, :]
    mu = rng.random(N) + 0.05
    Xd, mud = _t(X).to(dev), _t(mu).to(dev)
    i32, f64 = torch.int32, torch.float64
    kr, w1, nk, mo = (torch.empty(N, dtype=i32, device=dev), torch.empty(N, dtype=f64, device=dev),
                      torch.empty(1, dtype=i32, device=dev), torch.empty(N, dtype=f64, device=dev))
    nat.car_device(Xd, mud, kr, w1, nk, mo)
    n1 = nfun + 2
    null_row = torch.full((N,), 7.0, dtype=f64, device=dev)
    status = torch.full((1,), 99, dtype=i32, device=dev)
    nat.null_vector(Xd, nfun, kr, nk, n1, null_row, status)
    torch.cuda.synchronize()
    assert int(nk.item()) == n1 and int(status.item()) == 0
    krh, v = kr.cpu().numpy(), null_row.cpu().numpy()
    kept = krh >= 0
    assert (v[~kept] == 0.0).all() and np.abs(v[kept]).max() >= 1.0
    A2 = np.concatenate([X[kept, :nfun].T, np.ones((1, n1))], 0)        # (nfun + 1) x (nfun + 2)
    res = np.abs(A2 @ v[kept]).max() / (np.abs(A2).max() * np.abs(v[kept]).max())
    assert res < 1e-12, res
    ref = np.linalg.svd(A2)[2][-1]
    assert 1.0 - abs(ref @ v[kept]) / np.linalg.norm(v[kept]) < 1e-12
    # the first step's verdict is checked on the device
    nk_bad = torch.tensor([n1 - 1], dtype=i32, device=dev)
    nat.null_vector(Xd, nfun, kr, nk_bad, n1, null_row, status)
    assert int(status.item()) == -2
    nk_bad.fill_(-1)
    nat.null_vector(Xd, nfun, kr, nk_bad, n1, null_row, status)
    assert int(status.item()) == -1


def test_null_vector_kernel_reports_a_rank_deficient_matrix(dev):
    """Two identical feature columns among the survivors: A2 has a zero pivot row, its null space is a plane, and the
    reference's vector is whatever its SVD returns -- status -3, the caller's host route takes the level."""
    from sober_amd import _native as nat
    N, nfun = 12, 9
    rng = np.random.default_rng(3)
    X = rng.standard_normal((N, nfun + 1))
    X[:, 4] = X[:, 2]
    i32, f64 = torch.int32, torch.float64
    kr = torch.full((N,), -1, dtype=i32, device=dev)
    kr[:nfun + 2] = torch.arange(nfun + 2, dtype=i32, device=dev)
    nk = torch.tensor([nfun + 2], dtype=i32, device=dev)
    null_row, status = torch.empty(N, dtype=f64, device=dev), torch.full((1,), 99, dtype=i32, device=dev)
    nat.null_vector(_t(X).to(dev), nfun, kr, nk, nfun + 2, null_row, status)
    assert int(status.item()) == -3


def test_second_elimination_reads_the_null_vector_kernels_verdict(dev):
    """sober_second_elimination_rows with the status word of sober_null_vector (what the queued chain of the acquisition-guided
    branch relies on: nothing is read back between the launches): a regular level (status 0, the first step left n1 sets) is
    eliminated as without the word; status != 0 (a rank-deficient survivors' matrix) or another survivor count reports
    n_keep = -2, a first step that gave up -1 -- and writes nothing else."""
    from sober_amd import _native as nat
    i32, f64 = torch.int32, torch.float64
    rng = np.random.default_rng(5)
    Nsets, n1 = 30, 12
    kr1 = torch.full((Nsets,), -1, dtype=i32)
    kr1[torch.from_numpy(np.sort(rng.permutation(Nsets)[:n1]))] = torch.arange(n1, dtype=i32)
    kr1 = kr1.to(dev)
    null_row = _t(rng.standard_normal(Nsets)).to(dev)
    obj_row = _t(rng.standard_normal(Nsets)).to(dev)
    w1 = _t(rng.random(Nsets) + 0.1).to(dev)

    def run(nk1_val, status_val):
        nk1 = torch.tensor([nk1_val], dtype=i32, device=dev)
        status = None if status_val is None else torch.tensor([status_val], dtype=i32, device=dev)
        kr = torch.full((Nsets,), -7, dtype=i32, device=dev)
        ws = torch.full((Nsets,), -7.0, dtype=f64, device=dev)
        nk = torch.full((1,), -7, dtype=i32, device=dev)
        nat.second_elimination_rows(null_row, obj_row, w1, kr1, nk1, n1, kr, ws, nk, status=status)
        return kr.cpu().numpy(), ws.cpu().numpy(), int(nk.item())
    k0, w0, n0 = run(n1, None)
    k1, w1_, n1_ = run(n1, 0)
    assert n0 == n1_ == n1 - 1 and np.array_equal(k0, k1) and np.array_equal(w0, w1_)
    for nk1_val, st, want in ((n1, -3, -2), (n1 - 1, 0, -2), (n1 + 1, 0, -2), (-1, 0, -1)):
        k, w, n = run(nk1_val, st)
        assert n == want and (k == -7).all() and (w == -7.0).all(), (nk1_val, st, n)


def test_calc_obj_step_with_the_null_vector_kernel_equals_the_second_step_route(dev):
    """The acquisition-guided recombination (SOBER/_rchq.py:67-69, :87-106, :177-196) with the second elimination's direction
    from csrc/null_vector.hip (default) against the route of rounds 2-5 (a second Caratheodory step on the b + 1 survivors,
    `HipOps.obj_null_kernel = False`) and against the reference golden: same points, weights 1e-6 (the golden's bar)."""
    import warnings
    from sober_amd._ops_hip import HipOps
    from tests.golden.synth import SEED_CALL, calc_obj_fn, load_case
    case, inp, spec, z = load_case(os.path.join(GOLD, "recomb_rbf_calc_obj.npz"))
    res = []
    for flag in (True, False):
        ops = HipOps(dev)
        ops.obj_null_kernel = flag
        mu = _t(inp["mu0"].copy()).to(dev)
        torch.manual_seed(SEED_CALL)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            i, w = sober_amd.recombination(_t(inp["X_cand"]).to(dev), _t(inp["X_nys"]).to(dev), case["b"],
                                           sober_amd.Kernel(_kspec(spec), case["mode"]), init_weights=mu, calc_obj=calc_obj_fn, _ops=ops)
        res.append((i.cpu().numpy(), w.cpu().numpy()))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][0], z["idx"])
    np.testing.assert_allclose(res[0][1], res[1][1], rtol=1e-8)
    np.testing.assert_allclose(res[0][1], z["w"], rtol=1e-6)


def test_kmeans_with_an_empty_cluster_is_bounded_not_a_cliff(dev):
    """An empty cluster leaves a NaN centroid (SOBER/_weights.py:117-124 divides by a zero count); the reference's argmin then
    sends EVERY point to it.  Up to round 5 that cost the screened path (csrc/kmeans.hip) 10 ms per iteration twice over at
    1M x 20 -- every point through the list pass's tile walk sixteen at a time, then one workgroup summing a cluster that holds
    the whole pool: 105 ms for ten iterations against 2.4 (the advisor's "silent cliff", measured by this very test).  Now
    the list pass answers a NaN centroid without a tile and such a cluster's sum is shared by 64 workgroups: the degenerate
    run within 8 x the clean one, labels identical to the kernel that runs without the screen's workspace."""
    import time
    from sober_amd import _native as nat
    lib = nat.load()
    N, d, K = 1000000, 20, 500
    g = torch.Generator(device=dev); g.manual_seed(5)
    X = torch.rand(N, d, generator=g, dtype=torch.float64, device=dev)
    Xb = X.clone(); Xb[1] = Xb[0]                                      # two identical initial centroids: one cluster goes empty

    def run(Xd, nbytes):
        c = torch.empty(K, d, dtype=torch.float64, device=dev)
        cl = torch.empty(N, dtype=torch.int32, device=dev)
        ws = torch.zeros(max(nbytes, 8), dtype=torch.uint8, device=dev)
        for _ in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            nat._check(lib.sober_kmeans_lloyd(Xd.data_ptr(), N, d, K, 10, c.data_ptr(), cl.data_ptr(), ws.data_ptr() if nbytes else None,
                                              nbytes, nat._stream(Xd)), "kmeans")
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        return dt, cl.cpu().numpy(), c.cpu().numpy()
    nb = int(lib.sober_kmeans_ws_bytes(N, d, K))
    assert nb == int(lib.sober_kmeans_ws_bytes_screened(N, d, K)) > 0
    t_clean, _, _ = run(X, nb)
    t_bad, cl_bad, c_bad = run(Xb, nb)
    assert np.isnan(c_bad).any(), "the duplicated centroid was meant to empty a cluster"
    _, cl_ref, _ = run(Xb, 0)                                           # (no workspace: the (x - c)^2 kernel, no screen)
    assert np.array_equal(cl_bad, cl_ref)
    assert t_bad < 8.0 * t_clean, (t_bad, t_clean)


# --------------------------------------------------------------------------------------------- the live list in one launch
@pytest.mark.parametrize("N,density", [(1, 1.0), (63, 0.5), (2048, 1.0), (2049, 0.0), (20000, 1.0), (20000, 0.37), (100003, 0.5),
                                       (1 << 20, 0.9), (3_000_001, 0.01), (10_000_000, 1.0)])
def test_live_list_kernel_is_torch_nonzero(N, density, dev):
    """idx_story = arange(N)[mu != 0] (SOBER/_rchq.py:63-65) by csrc/compact.hip's single launch (ticket-ordered tiles,
    decoupled look-back): the positions torch.nonzero returns, in its order, NaN counted as live; the workspace is left zero,
    so the same one serves call after call (three calls here, the weights changed in between)."""
    from sober_amd import _native as nat
    g = torch.Generator().manual_seed(N)
    mu = torch.rand(N, generator=g, dtype=torch.float64)
    mu[torch.rand(N, generator=g) >= density] = 0.0
    if N > 70:
        mu[7], mu[N - 1], mu[64] = float("nan"), -0.0, -1e-300
    mu = mu.to(dev)
    ws = torch.zeros(nat.nonzero_ws_bytes(N), dtype=torch.uint8, device=dev)
    cnt = torch.full((1,), -5, dtype=torch.int64, device=dev)
    for rep in range(3):
        out = torch.full((N,), -1, dtype=torch.int32, device=dev)
        nat.nonzero_i32(mu, out, cnt, ws)
        ref = torch.nonzero(mu != 0).flatten()
        assert int(cnt.item()) == ref.numel()
        assert torch.equal(out[:ref.numel()].long(), ref)
        assert bool((out[ref.numel():] == -1).all())
        assert not bool(ws.any())
        mu = torch.roll(mu, 977)
        if N > 10:
            mu[rep] = 0.0


# ------------------------------------------------------------------ the memory-resident Caratheodory route (csrc/car_big.hip)
def _car_big_case(N, m, seed, dev, decay=0.0, zero_every=0):
    from sober_amd import _native as nat
    from tests.test_car_algorithm import nullspace_gebrd
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((N, m - 1)) * np.exp(-decay * np.arange(m - 1))[None, :]
    mu = rng.random(N) + 0.05
    if zero_every:
        mu[::zero_every] = 0.0
    mu /= mu.sum()
    Xd, mud = _t(X).to(dev), _t(mu).to(dev)
    kr = torch.empty(N, dtype=torch.int32, device=dev)
    ws = torch.zeros(N, dtype=torch.float64, device=dev)
    nk = torch.empty(1, dtype=torch.int32, device=dev)
    mo = torch.empty(N, dtype=torch.float64, device=dev)
    phi = torch.empty(N, N - m, dtype=torch.float64, device=dev)
    nat.car_device(Xd, mud, kr, ws, nk, mo, phi_out=phi, big=True)
    torch.cuda.synchronize()
    A = np.vstack([np.ones(N), X.T])
    return X, mu, kr.cpu().numpy(), ws.cpu().numpy(), int(nk.item()), mo.cpu().numpy(), phi.cpu().numpy(), nullspace_gebrd(A), A


@pytest.mark.parametrize("N,m,decay,zero_every", [(64, 20, 0.0, 0), (13, 12, 0.0, 0), (200, 101, 0.03, 0), (400, 201, 0.0, 7), (500, 251, 0.01, 0),
                                                  (600, 301, 0.0, 0), (1024, 513, 0.005, 0), (1100, 540, 0.0, 5), (700, 120, 0.0, 0),
                                                  (1536, 800, 0.002, 0)])
def test_car_big_vs_gebrd_restatement(N, m, decay, zero_every, dev):
    """The memory-resident Caratheodory kernels (csrc/car_big.hip: a launch per dependency, any N <= 2048): null-space basis = the
    dgebd2 reflectors' (numpy restatement, itself pinned to LAPACK's SVD on the reference's inputs in tests/test_car_algorithm.py)
    and = torch.linalg.svd's Vh[m:] (SOBER/_rchq.py:231-234); the pivots of :237-266 select the restatement's sets with the same
    weights (zero masses: alpha = 0 pivots and first-index ties included); the result is a recombination."""
    from tests.test_car_algorithm import pivots
    X, mu, kr, ws, nk, mo, phi, Phi_np, A = _car_big_case(N, m, 3000 + N + m, dev, decay, zero_every)
    np.testing.assert_allclose(phi, Phi_np, rtol=0, atol=5e-12)         # (held at 2048 x 1025 too: 49 s of numpy, not in the suite)
    if N <= 1100:
        Vh = torch.linalg.svd(torch.from_numpy(A))[2].numpy()
        np.testing.assert_allclose(phi, Vh[m:].T, rtol=0, atol=1e-9)
    w_np, idx_np = pivots(Phi_np, mu)
    idx = np.flatnonzero(kr >= 0)
    assert np.array_equal(idx, idx_np)
    assert np.array_equal(kr[idx], np.arange(nk))
    np.testing.assert_allclose(ws[:nk], w_np, rtol=1e-8)
    assert np.array_equal(np.flatnonzero(mo > 0), idx)
    assert (ws[:nk] > 0).all() and nk <= m
    np.testing.assert_allclose(ws[:nk].sum(), mu.sum(), rtol=1e-12)
    np.testing.assert_allclose(ws[:nk] @ X[idx], mu @ X, rtol=0, atol=1e-11)


def test_car_big_equals_the_register_resident_routes_on_reference_levels(dev):
    """The same step through all three device implementations on the reference's own level inputs (batch 100): same sets,
    weights to rounding -- and bit-equal from run to run (fixed reduction trees, no atomics)."""
    from sober_amd import _native as nat
    z = np.load(os.path.join(GOLD, "recomb_matern_medium.npz"))
    for i in range(int(z["n_levels"])):
        X, mu = _t(z[f"L{i}_X_tmp"]).to(dev), _t(z[f"L{i}_tot_weights"]).to(dev)
        N = X.shape[0]
        outs = []
        for kw in (dict(), dict(multi_cu=True), dict(big=True), dict(big=True)):
            kr = torch.empty(N, dtype=torch.int32, device=dev)
            ws = torch.zeros(N, dtype=torch.float64, device=dev)
            nk = torch.empty(1, dtype=torch.int32, device=dev)
            mo = torch.empty(N, dtype=torch.float64, device=dev)
            nat.car_device(X, mu, kr, ws, nk, mo, **kw)
            outs.append((kr.cpu().numpy(), ws.cpu().numpy(), int(nk.item())))
        (k1, w1, n1), (k2, w2, n2), (k3, w3, n3), (k4, w4, n4) = outs
        assert n1 == n3 and np.array_equal(k1, k3) and np.array_equal(k2, k3), i
        np.testing.assert_allclose(w3[:n3], w1[:n1], rtol=1e-10)
        assert np.array_equal(k3, k4) and np.array_equal(w3, w4), i
        assert np.array_equal(np.flatnonzero(k3 >= 0), z[f"L{i}_idx_star"])


# ------------------------------------------------------------------ the acquisition-guided branch's levels as one queued chain
@pytest.mark.parametrize("kind,mode,N,M,d,b,seed", [("rbf", "predictive_covariance", 100000, 500, 10, 100, 1), ("matern52", "kernel", 30000, 300, 5, 40, 2),
                                                    ("rbf", "weighted_predictive_covariance", 20000, 200, 4, 30, 3), ("tanimoto", "kernel", 40000, 300, 1024, 60, 4),
                                                    ("rbf", "predictive_covariance", 12345, 150, 3, 17, 5)])
def test_calc_obj_levels_queued_equal_the_level_by_level_route(kind, mode, N, M, d, b, seed, dev):
    """recombination(..., calc_obj=...) (SOBER/_rchq.py:67-69, :138-150, :173-196) with its levels as ONE queued chain of the level
    executor (sober_level_loop_obj: every verdict read on the device, one synchronisation) against the level-by-level route of the
    same package (`HipOps.queue_obj_levels = False`: a visit to Python and a read-back per level): the same kernels on the same
    data in the same order -- bit-identical points and weights; the chain really ran (fewer read-backs is its point)."""
    import warnings
    from oracle import sober_oracle as O
    from sober_amd._ops_hip import HipOps
    from sober_amd import _native as nat
    from tests.golden.synth import SEED_CALL, build_spec, calc_obj_fn, synth
    case = dict(kind=kind, mode=mode, N=N, M=M, d=d, b=b, n_obs=0 if mode == "kernel" else 40, seed=seed, ard=False, bit_p=0.06)
    inp = synth(case)
    spec = build_spec(case, inp)
    res, done = [], []
    for flag in (True, False):
        ops = HipOps(dev)
        ops.queue_obj_levels = flag
        calls = []
        orig = nat.level_loop_obj
        nat.level_loop_obj = lambda *a, **k: (calls.append(orig(*a, **k)), calls[-1])[1]
        try:
            mu = _t(inp["mu0"].copy()).to(dev)
            torch.manual_seed(SEED_CALL)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                i, w = sober_amd.recombination(_t(inp["X_cand"]).to(dev), _t(inp["X_nys"]).to(dev), b,
                                               sober_amd.Kernel(_kspec(spec), mode), init_weights=mu, calc_obj=calc_obj_fn, _ops=ops)
        finally:
            nat.level_loop_obj = orig
        res.append((i.cpu().numpy(), w.cpu().numpy(), mu.cpu().numpy()))
        done.append(sum(len(c[0]) for c in calls))
    assert done[1] == 0 and done[0] >= 2, done
    assert np.array_equal(res[0][0], res[1][0])
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])


def test_calc_obj_chain_hands_an_irregular_level_to_the_level_by_level_route(dev):
    """A pool of only eight distinct fingerprints: the barycentre matrix of a level has rank <= 8 < batch, the first step leaves
    fewer than n + 2 sets, and the reference then reads a singular vector of a full-rank matrix (SOBER/_rchq.py:87-106) -- not a
    case for the device kernels.  The queued chain must stop there with weights and list untouched (n_keep = -2 in the flags ->
    the update stops the chain) and the level-by-level route take over: the same result as with the chain switched off, bit for
    bit; and a valid one (positive weights, unit mass, at most batch points)."""
    import warnings
    from oracle import sober_oracle as O
    from sober_amd._ops_hip import HipOps
    from tests.golden.synth import SEED_CALL, build_spec, calc_obj_fn
    rng = np.random.default_rng(8)
    d, b, N, M = 512, 16, 6000, 40                           # (512 bits: the fingerprint level kernel whose levels are queued)
    pats = (rng.random((8, d)) < 0.1).astype(np.float64)
    X = pats[rng.integers(0, 8, N)]
    Xn = X[rng.permutation(N)[:M]].copy()
    mu0 = rng.random(N); mu0 /= mu0.sum()
    spec = O.make_spec(O.TANIMOTO, _t(np.zeros((1, d))), _t(np.ones(1)), outputscale=1.0, noise=1e-2, mean_const=0.0, y_obs=_t(np.zeros(1)))
    from sober_amd import _native as nat
    res, chains = [], []
    orig = nat.level_loop_obj
    nat.level_loop_obj = lambda *a, **k: (chains.append(orig(*a, **k)), chains[-1])[1]
    try:
        for flag in (True, False):
            ops = HipOps(dev)
            ops.queue_obj_levels = flag
            mu = _t(mu0.copy()).to(dev)
            torch.manual_seed(SEED_CALL)
            timers = {}
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                i, w = sober_amd.recombination(_t(X).to(dev), _t(Xn).to(dev), b, sober_amd.Kernel(_kspec(spec), "kernel"), init_weights=mu,
                                               calc_obj=calc_obj_fn, _ops=ops, _timers=timers)
            res.append((i.cpu().numpy(), w.cpu().numpy(), mu.cpu().numpy()))
            assert "car_host" in timers, timers             # (the irregular levels are the host route's)
    finally:
        nat.level_loop_obj = orig
    # the chain was enqueued once (first run) and completed fewer levels than the pool has: it stopped at the irregular one
    assert len(chains) == 1 and len(chains[0][0]) < 4 and chains[0][1] > 2 * b, chains
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
    i, w, _ = res[0]
    assert 0 < len(i) <= b and (w > 0).all() and abs(w.sum() - 1.0) < 1e-9


# ------------------------------------------------------------------ GP prediction from the root of W
@pytest.mark.parametrize("kind,d,n_obs,N", [("rbf", 10, 200, 40000), ("matern52", 6, 130, 5000), ("rbf", 3, 17, 3000), ("tanimoto", 512, 90, 6000),
                                            ("rbf", 20, 255, 9000), ("matern52", 8, 300, 7000), ("rbf", 10, 511, 6000), ("tanimoto", 1024, 400, 3000)])
def test_prediction_from_the_root_equals_prediction_from_w(kind, d, n_obs, N, dev, monkeypatch):
    """predict / PI (SOBER/_gp.py:212-238, SOBER/_pi.py:20-38) with the fused kernel working from S^T -- the triangular root
    gpytorch caches: the tile products above the diagonal skipped -- against the same kernel working from W = S S^T
    (SOBER_PREDICT_FROM_W=1, rounds 5-6a), and with a root that is NOT triangular (S Q, Q orthogonal: the same W; the device flag
    then says so and nothing is skipped): mean / variance 1e-10, pi 1e-9.  Beyond 255 observations (up to 511: eight row tiles per
    wave, round 6 -- the materialised route before) also against that materialised route (SOBER_PREDICT_MATERIALISED=1)."""
    from oracle import sober_oracle as O
    from sober_amd import _pi
    from tests.golden.synth import build_spec, synth
    case = dict(kind=kind, mode="predictive_covariance", N=N, M=20, d=d, n_obs=n_obs, b=5, seed=n_obs, ard=False, bit_p=0.08, mean_const=0.2)
    inp = synth(case)
    spec = build_spec(case, inp)
    ks = _kspec(spec).to(dev)
    X = _t(inp["X_cand"]).to(dev)

    def run(ksp):
        pi = sober_amd.PI(ksp)
        m, v = _pi.predict(X, ksp)
        return m.cpu().numpy(), v.cpu().numpy(), pi(X).cpu().numpy()
    m1, v1, p1 = run(ks)
    assert bool((torch.triu(ks.S_cache.t(), 1) == 0).all())     # (the oracle's spec carries the triangular root)
    monkeypatch.setenv("SOBER_PREDICT_FROM_W", "1")
    m0, v0, p0 = run(ks)
    monkeypatch.delenv("SOBER_PREDICT_FROM_W")
    g = torch.Generator().manual_seed(1)
    Q, _ = torch.linalg.qr(torch.randn(n_obs, n_obs, dtype=torch.float64, generator=g))
    ks2 = sober_amd.KernelSpec(ks.kind, ks.lengthscale, ks.outputscale, ks.X_obs, (ks.S_cache @ Q.to(dev)).contiguous(), ks.noise,
                               ks.mean_const, ks.alpha)
    m2, v2, p2 = run(ks2)
    if n_obs > 255:
        monkeypatch.setenv("SOBER_PREDICT_MATERIALISED", "1")
        m3, v3, p3 = run(ks)
        monkeypatch.delenv("SOBER_PREDICT_MATERIALISED")
        np.testing.assert_allclose(m1, m3, rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(v1, v3, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(p1, p3, rtol=1e-8, atol=1e-13)
    for m, v, p_ in ((m1, v1, p1), (m2, v2, p2)):
        np.testing.assert_allclose(m, m0, rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(v, v0, rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(p_, p0, rtol=1e-9, atol=1e-13)


def test_pi_on_a_live_model_follows_the_model_and_caches_what_did_not_change(dev):
    """PI on a LIVE model (anything that is not a KernelSpec snapshot: the reference evaluates `self.model` on every call,
    SOBER/_pi.py:20-38): the model is read on every call, but the prepared observations, the root of W and the threshold are
    rebuilt only when what was read has changed (tensor identity / version counters / scalars).  Unchanged model: the same
    object is reused; alpha changed IN PLACE: the next call follows it (pi equals a fresh PI's); eta assigned by the caller: the
    next call of a live model re-derives it, as before."""
    from oracle import sober_oracle as O
    from tests.golden.synth import build_spec, synth
    case = dict(kind="rbf", mode="predictive_covariance", N=5000, M=20, d=6, n_obs=60, b=5, seed=5, ard=False, mean_const=0.1)
    inp = synth(case)
    ks = _kspec(build_spec(case, inp)).to(dev)

    class Live:                                              # (a duck model: spec_from_model reads .kernel_spec)
        def __init__(self, spec):
            self.spec = spec

        def kernel_spec(self):
            return self.spec
    X = _t(inp["X_cand"]).to(dev)
    live = Live(ks)
    pi = sober_amd.PI(live)
    p0 = pi(X).clone()
    side0 = pi._model_side
    p1 = pi(X)
    assert pi._model_side is side0 and torch.equal(p0, p1)          # nothing changed: nothing rebuilt
    ks.alpha.mul_(1.5)                                               # the model moves in place (version counter)
    p2 = pi(X)
    assert pi._model_side is not side0
    assert torch.equal(p2, sober_amd.PI(Live(ks))(X)) and not torch.equal(p2, p0)
    eta_model = pi.eta
    pi.eta = eta_model + 1.0
    p3 = pi(X)                                                       # a live model's call re-derives the threshold
    assert abs(pi.eta - eta_model) < 1e-15 and torch.equal(p3, p2)
