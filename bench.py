#!/usr/bin/env python3
"""Benchmark of the kernel-recombination step (BASELINE.json metric) on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config C]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W [--config C]          (N > 1)

A "step" is ONE `sampling_recombination` call -- kernel build, Gram + Nystrom basis, every halving level, the final
direct level -- on a synthetic pool that is already resident in HBM.  `--config` picks the BASELINE.json
configuration (default 2, the one the metric is quoted on):
  1  Branin-shaped   d=2  RBF (ARD)      N_rec=2k    N_nys=100  batch=10
  2  Ackley-shaped   d=10 RBF            N_rec=100k  N_nys=500  batch=100
  3  Hartmann-shaped d=6  Matern-5/2     N_rec=50k   N_nys=500  batch=200
  4  Rosenbrock      d=20 RBF            N_rec=1M    N_nys=500  batch=100   (the 8-GPU configuration)
  5  Malaria-shaped  2048-bit Tanimoto   N_rec=250k  N_nys=500  batch=100   (weighted posterior covariance)
  6  big pool        d=20 RBF            N_rec=8M    N_nys=500  batch=100   (not a BASELINE.json configuration: the throughput
                                                                              regime on one GPU; parity by invariants)
`--funnel` (one GPU, configuration 2's shapes): the ACQUISITION step as a user of `Sober.next_batch` sees it -- pi weights over
the pool, cleansing_weights, the KMeans Nystrom subsample, sampling_recombination (SOBER/_sober.py:125-195,
_sampler.py:264-323) -- with per-phase ms and the oracle's same funnel on the host cores beside it.
N > 1: one rank per GPU over RCCL, the pool row-sharded, one small all-reduce per level.  Configurations 1-3 and 5
scale WEAKLY (every rank owns a full-size shard, one recombination over the N-fold pool); configuration 4 is the
reference's 8-GPU case and scales STRONGLY (the 1M-row pool is split N ways: 125k rows per GPU at N = 8).
Rank 0 prints ONE JSON line.  `value` = candidates reduced per second, whole job.
`roofline` is for the dominant kernel (the level reduction): algorithmic FP64 flop ((2d + 2 + C_k)
per kernel entry, C_k = 28 for the software FP64 exp, 40 for Matern-5/2) / the summed durations of ALL its launches in
every --event-every-th step of the timed region (HIP events carried in the dispatches themselves, i.e. the kernels'
own begin / end timestamps -- calls x AverageNs of a rocprofv3 --kernel-trace --stats run of the same command; a
dispatch that carries events costs the stream 10-20 us, hence not every step), against the FP64 peak; for the Tanimoto kernel the integer operations of popcount(x & y) as an INT8 GEMM (2 per bit and (row,
candidate) pair) against the dense INT8 matrix peak.  `roofline.step_frac` is the whole step's algorithmic flop
(SURVEY.md 8d F_alg) / ms_per_step against the same peak; `hbm` = algorithmic bytes / time (small by construction).
`cpu_baseline` = the oracle (a torch-CPU port of the reference's own arithmetic) on this box's host cores, on a
bounded sample of the same workload: reference-shaped (materialises the (E, M, S) tensor of SOBER/_rchq.py:124) and
streaming (the same sums over cache-sized element blocks); median of 3 each, `value` is the faster of the two.
"""
import argparse
import json
import os
import subprocess
import sys
import time
import warnings


def _self_launch():
    """`python bench.py --gpus N` with N > 1 and no launcher in front of it: start the N ranks as a CHILD
    `python -m torch.distributed.run` (one rank per GPU, rendezvous on 127.0.0.1) from this process -- which has imported
    nothing that touches the GPU yet and never does --, pass the children's output through (rank 0 prints the ONE JSON
    line) and return their exit code.  Under a launcher (WORLD_SIZE set) this is a no-op: the ranks run main() below."""
    if "WORLD_SIZE" in os.environ:
        return None
    n = 1
    argv = sys.argv[1:]
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            n = int(argv[i + 1])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
    if n <= 1:
        return None
    import socket
    with socket.socket() as sk:                                      # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.run(cmd, env=env).returncode


if __name__ == "__main__":
    _rc = _self_launch()
    if _rc is not None:
        sys.exit(_rc)

import numpy as np          # noqa: E402
import torch                # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import sober_amd  # noqa: E402
from tests.golden.synth import SEED_CALL, build_spec, synth  # noqa: E402

CONFIGS = {
    1: dict(kind="rbf", mode="predictive_covariance", N=2000, M=100, d=2, b=10, n_obs=30, seed=0, ard=True,
            name="Branin-shaped d=2 RBF (ARD) posterior covariance", golden="recomb_cfg1_rbf_ard.npz"),
    2: dict(kind="rbf", mode="predictive_covariance", N=100000, M=500, d=10, b=100, n_obs=200, seed=0,
            name="Ackley-shaped d=10 RBF posterior covariance", golden="recomb_cfg2_rbf.npz"),
    3: dict(kind="matern52", mode="predictive_covariance", N=50000, M=500, d=6, b=200, n_obs=200, seed=0,
            name="Hartmann-shaped d=6 Matern-5/2 posterior covariance"),
    4: dict(kind="rbf", mode="predictive_covariance", N=1000000, M=500, d=20, b=100, n_obs=200, seed=0,
            name="Rosenbrock-shaped d=20 RBF posterior covariance", strong=True, cpu_sample_N=100000),
    5: dict(kind="tanimoto", mode="weighted_predictive_covariance", N=250000, M=500, d=2048, b=100, n_obs=200,
            seed=10, bit_p=0.04, mean_const=0.3, name="Malaria-shaped 2048-bit Tanimoto weighted posterior covariance",
            cpu_sample_N=20000),
    # beyond BASELINE.json: the throughput regime (one GPU, the level kernel is nearly all of the step) -- the pool is
    # generated on the device, parity by the result's invariants (no CPU reference at this size in bench time)
    6: dict(kind="rbf", mode="predictive_covariance", N=8000000, M=500, d=20, b=100, n_obs=200, seed=0,
            name="big pool d=20 RBF posterior covariance (beyond BASELINE.json: the throughput regime)",
            cpu_sample_N=100000, device_pool=True),
    # beyond BASELINE.json: a batch beyond the register-resident Caratheodory kernels (csrc/car_big.hip: the matrix in memory, a
    # launch per dependency; host LAPACK until round 5) -- cfg-2's pool with batch 250
    7: dict(kind="rbf", mode="predictive_covariance", N=100000, M=600, d=10, b=250, n_obs=200, seed=0,
            name="Ackley-shaped d=10 RBF posterior covariance at batch 250 (beyond BASELINE.json: beyond the register-resident "
                 "Caratheodory kernels)", cpu_sample_N=20000),
}
FP64_PEAK_TFLOPS = 78.6          # MI355X FP64 vector = FP64 matrix (vendor)
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md
FP4_PEAK_TOPS = 10000.0          # dense FP4 / FP6 MFMA (block-scaled f8f6f4 form) = 4 x the bf16 rate (MI355X_MICROARCH.md, matrix cores)
CK = {"rbf": 28, "matern52": 40}
# What the FP64 units of an MI355X sustain chip-wide (scripts/dp_rate_probe.hip, wall clock, all 256 CUs; committed
# output: profiles/r05_dp_rate.txt, first taken in round 3: profiles/r03_dp_rate.txt): the clock under FP64 load is ~1.9 GHz, not the 2.4 GHz of the vendor figure.
FP64_MEASURED_TFLOPS = {"v_fma_f64": 61.0, "v_mfma_f64_16x16x4": 63.0, "level-kernel mix (12 MFMA + 160 FMA)": 69.0,
                        "source": "profiles/r05_dp_rate.txt"}


def pmc_traffic(config):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of THIS configuration
    (profiles/rNN_pmc_traffic.json, newest round; written by scripts/collect_profiles.py from separate FETCH_SIZE and
    WRITE_SIZE runs of `bench.py --config C`: mean over the kernel's launches of (2 * FETCH_SIZE + WRITE_SIZE) * 1024
    -- the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md).  None when no pass is committed for it."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1])).get(str(config))
    except Exception:
        return None, None
    return (d["bytes_per_launch"], os.path.relpath(files[-1], ROOT)) if d else (None, None)


def allreduce_route():
    """Which all-reduce the sharded level loop used in this process (sober_amd/_engine.py: DistComm.native_allreduce)."""
    from sober_amd._engine import DistComm
    if any(ent[1] not in (None, False) for ent in DistComm._PEER.values()):
        return "one-shot direct-peer kernel, sums in rank order: csrc/peer_reduce.hip"
    if any(ent[1] not in (None, False) for ent in DistComm._RCCL.values()):
        return "ncclAllReduce issued from C: csrc/rccl_link.cpp"
    return "the group's own all_reduce"


def host_cpu():
    model = None
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"model": model, "logical_cpus": os.cpu_count()}


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def build_inputs(cfg, rank, world, dev):
    """-> (X_cand, X_nys, mu0 on the device, spec, host inputs or None, rows on this rank)."""
    case = {k: v for k, v in cfg.items() if k not in ("name", "golden", "strong", "cpu_sample_N", "device_pool")}
    N_loc = case["N"] // world if cfg.get("strong") else case["N"]
    if cfg["kind"] == "tanimoto":
        # 250k x 2048 FP64 0/1 = 4 GB: built on the device; the GP side (observations, caches) from the small stream
        small = dict(case, N=case["M"] + 1000)
        base = synth(small)
        spec = build_spec(small, base)
        g0 = torch.Generator(device=dev)                             # rank 0's stream: its shard and the Nystrom points
        g0.manual_seed(case["seed"])
        X0 = (torch.rand(N_loc, case["d"], device=dev, generator=g0) < case["bit_p"]).to(torch.float64)
        Xn = X0[torch.randperm(N_loc, device=dev, generator=g0)[:case["M"]]].clone()
        g = torch.Generator(device=dev)
        g.manual_seed(case["seed"] + 1000 + rank)
        X = X0 if rank == 0 else (torch.rand(N_loc, case["d"], device=dev, generator=g) < case["bit_p"]).to(torch.float64)
        del X0
        mu0 = torch.rand(N_loc, device=dev, generator=g, dtype=torch.float64)
        mu0 /= mu0.sum() * world
        return X, Xn, mu0, spec, None, N_loc
    if cfg.get("device_pool"):
        small = dict(case, N=max(cfg.get("cpu_sample_N", 100000), case["M"] + 1000))
        small.pop("device_pool", None)
        base = synth(small)
        spec = build_spec(small, base)
        g = torch.Generator(device=dev)
        g.manual_seed(case["seed"] + 1 + rank)
        X = torch.rand(N_loc, case["d"], generator=g, dtype=torch.float64, device=dev)
        X[:small["N"]] = t(base["X_cand"]).to(dev)                   # (the head of the pool: what the CPU sample runs on)
        mu0 = torch.rand(N_loc, generator=g, dtype=torch.float64, device=dev)
        mu0 /= mu0.sum() * world
        return X, t(base["X_nys"]).to(dev), mu0, spec, None, N_loc
    shard = dict(case, N=N_loc, seed=case["seed"] + rank)            # every rank synthesises its own shard
    inp = synth(shard)
    base = synth(dict(case, N=N_loc)) if rank else inp               # X_nys / X_obs come from rank 0's stream
    spec = build_spec(case, base)
    return t(inp["X_cand"]).to(dev), t(base["X_nys"]).to(dev), t(inp["mu0"] / world).to(dev), spec, inp, N_loc


def cpu_baseline_for(cfg, spec, inp, dev_inputs, cpu_steps):
    """The oracle on the host cores, on a bounded sample (about 10-30 s of CPU work)."""
    from oracle import sober_oracle as O
    N_s = min(cfg.get("cpu_sample_N", cfg["N"]), cfg["N"])
    if inp is not None:
        Xc, Xn, m0 = t(inp["X_cand"][:N_s]), t(inp["X_nys"]), inp["mu0"][:N_s].copy()
    else:
        X_dev, Xn_dev, mu_dev = dev_inputs
        Xc, Xn, m0 = X_dev[:N_s].cpu(), Xn_dev.cpu(), mu_dev[:N_s].cpu().numpy().copy()
    m0 = m0 / m0.sum()
    ok = O.Kernel(spec, cfg["mode"])

    def cpu_step(stream):
        m = t(m0.copy())
        torch.manual_seed(SEED_CALL)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return O.recombination(Xc, Xn, cfg["b"], ok, init_weights=m, stream_elements=stream)
    # the reference-shaped CPU path does not scale with cores (memory-bound temporaries): both variants at 8 threads.
    # (Rounds 2-5 also timed them at half the host's hardware threads -- 64 on the GPU boxes: 3-7x SLOWER than at 8 every
    # time, and three quarters of the command's wall clock; profiles/r05_bench_cfg2.json keeps the last of those figures.)
    runs = {}
    old = torch.get_num_threads()
    for variant, stream in (("reference_shaped", None), ("streaming", 16)):
        for th in (min(8, os.cpu_count() or 1),):
            torch.set_num_threads(th)
            cpu_step(stream)                                  # warm-up
            each = []
            for _ in range(cpu_steps):
                c0 = time.perf_counter()
                cpu_step(stream)
                each.append(time.perf_counter() - c0)
            runs[(variant, th)] = float(np.median(each))
    torch.set_num_threads(old)
    best = min(runs, key=runs.get)
    return {"value": N_s / runs[best], "unit": "candidates/s", "cores": best[1], "kind": "port", "variant": best[0],
            "host": host_cpu(), "ms_per_step": runs[best] * 1e3, "statistic": f"median of {cpu_steps}",
            "ms_per_step_by_variant_and_threads": {f"{v}@{th}": s * 1e3 for (v, th), s in runs.items()},
            "sample": f"median of {cpu_steps} recombination steps after 1 warm-up on "
                      + ("the full workload" if N_s == cfg["N"] else f"the first {N_s} candidates of the workload")
                      + f" (N_nys={cfg['M']}, batch={cfg['b']}), oracle = torch CPU FP64 port of the reference: "
                        "reference-shaped (materialised (E, M, S) tensor) and streaming (16-element blocks), each at "
                        "8 threads; the faster one reported"}


class _PoolPrior:
    """A continuous prior whose draw IS the resident synthetic pool (density 1 on the unit cube): `Sober.next_batch` then
    runs the reference's sampled-prior control flow (SOBER/_sampler.py:264-323) on device-resident candidates."""
    type = "continuous"

    def __init__(self, X):
        self.X, self.n_dims = X, X.shape[1]

    def sample(self, n):
        return self.X[:n]

    def pdf(self, X):
        return torch.ones(len(X), dtype=X.dtype, device=X.device)


class _Model:
    """What `sober_amd.Sober` touches of an exact-GP model when the kernel specification is given directly."""

    def __init__(self, kernel_spec, n_obs):
        self.kernel_spec = kernel_spec
        self.train_targets = torch.zeros(n_obs, dtype=torch.double)


def funnel(args):
    """`bench.py --funnel`: one acquisition step = `Sober.next_batch(n_rec, n_nys, batch)` on a continuous prior at
    configuration 2's shapes, device-resident; phases timed with a synchronisation either side (a second, unbracketed run
    gives ms_per_step); the oracle's same funnel on the host cores as cpu_baseline; parity: the fixture of the reference's
    own `Sober.next_batch` (tests/golden/sober_next_batch.npz) and the batch against the oracle's funnel at this size."""
    assert int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.gpus == 1, "--funnel is a one-GPU line"
    assert torch.cuda.is_available(), "bench.py needs the MI355X (no CPU fallback)"
    cfg = CONFIGS[2]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    X_cand, X_nys0, mu0, spec, inp, N = build_inputs(cfg, 0, 1, dev)
    ks = sober_amd.KernelSpec(spec.kind, spec.lengthscale, spec.outputscale, spec.X_obs, spec.S_cache,
                              spec.noise, spec.mean_const, spec.alpha)
    sober_amd.setting_parameters(device=dev, dtype=torch.double)
    M, b = cfg["M"], cfg["b"]

    def make():
        return sober_amd.Sober(_PoolPrior(X_cand), _Model(ks, cfg["n_obs"]), kernel_type=cfg["mode"],
                               prior_updater=lambda s_, X, w: None)

    def next_batch(sob):
        torch.manual_seed(SEED_CALL)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return sob.next_batch(N, M, b, return_weights=True)

    sob = make()
    for _ in range(max(args.warmup, 3)):
        w_b, X_b = next_batch(sob)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        w_b, X_b = next_batch(sob)
    torch.cuda.synchronize()
    ms_per_step = (time.perf_counter() - t0) / args.steps * 1e3
    # per phase: the four calls bracketed by synchronisations
    phases = {"pi_weights": 0.0, "cleansing_weights": 0.0, "kmeans_resampling": 0.0, "sampling_recombination": 0.0}
    calls = {k: 0 for k in phases}

    class bracket:                                                   # (keeps the wrapped object's attributes: Sober reads pi.model)
        def __init__(self, name, fn):
            self._name, self._fn = name, fn

        def __getattr__(self, attr):
            return getattr(self._fn, attr)

        def __call__(self, *a, **k):
            torch.cuda.synchronize()
            c0 = time.perf_counter()
            out = self._fn(*a, **k)
            torch.cuda.synchronize()
            phases[self._name] += time.perf_counter() - c0
            calls[self._name] += 1
            return out
    sob2 = make()
    sob2.pi = bracket("pi_weights", sob2.pi)
    sob2.cleansing_weights = bracket("cleansing_weights", sob2.cleansing_weights)
    sob2.kmeans_resampling = bracket("kmeans_resampling", sob2.kmeans_resampling)
    sob2.sampling_recombination = bracket("sampling_recombination", sob2.sampling_recombination)
    next_batch(sob2)
    for k in phases:
        phases[k], calls[k] = 0.0, 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        next_batch(sob2)
    torch.cuda.synchronize()
    ms_bracketed = (time.perf_counter() - t0) / args.steps * 1e3
    phases_ms = {k: v / args.steps * 1e3 for k, v in phases.items()}
    phases_ms["other (draws, checks, gathers, host)"] = ms_bracketed - sum(phases_ms.values())
    # the oracle's same funnel (SOBER/_sampler.py:264-323 with the updater a no-op): once, on the host cores
    cpu = None
    parity = {}
    if not args.no_cpu_baseline:
        from oracle import sober_oracle as O
        Xh = t(inp["X_cand"])
        cph = {}
        old_threads = torch.get_num_threads()
        torch.set_num_threads(8)

        def timed(name, fn):
            c0 = time.perf_counter()
            out = fn()
            cph[name] = cph.get(name, 0.0) + (time.perf_counter() - c0) * 1e3
            return out
        # the reference's control flow (sober_amd/_sampled_prior.py restates SOBER/_sampler.py:264-323) over the ORACLE's
        # arithmetic: pi, cleansing_weights, check_weights, KMeans, recombination all from oracle/sober_oracle.py
        from sober_amd._sampled_prior import sampling_candidates

        class _OracleSober:
            label, thresh, thresh_initial, flag, prior_updater = "continuous", 5, 5, False, staticmethod(lambda s_, X, w: None)

            def __init__(self):
                self.prior = _PoolPrior(Xh)
                opi = O.PI(spec)
                self.pi = lambda X: timed("pi_weights", lambda: opi(X))

            def cleansing_weights(self, w):
                return timed("cleansing_weights", lambda: O.cleansing_weights(w))

            def check_weights(self, w):
                return O.check_weights(w, self.thresh)

            def nystrom_subsample(self, X, w, n):
                return timed("kmeans_resampling", lambda: O.kmeans_chunked(X, K=n, Niter=10)[1])
        c_all = time.perf_counter()
        Xc_ref, Xn_ref, wts = sampling_candidates(_OracleSober(), N, M)
        torch.manual_seed(SEED_CALL)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            idx_ref, w_ref = timed("sampling_recombination", lambda: O.recombination(
                Xc_ref, Xn_ref, b, O.Kernel(spec, cfg["mode"]), init_weights=wts, stream_elements=16))
        cpu_ms = (time.perf_counter() - c_all) * 1e3
        torch.set_num_threads(old_threads)
        cpu = {"value": N / (cpu_ms * 1e-3), "unit": "candidates/s", "cores": 8, "kind": "port", "ms_per_step": cpu_ms,
               "phases_ms_per_step": cph, "host": host_cpu(),
               "sample": "one acquisition step on the full workload (100k candidates): pi weights + cleansing twice, chunked KMeans "
                         "(10 Lloyd iterations, 500 clusters), streaming recombination; oracle = torch CPU FP64 port of the reference"}
        Xb_ref = Xc_ref[idx_ref]
        same = bool(X_b.shape == Xb_ref.shape and torch.equal(X_b.cpu(), Xb_ref))
        parity["batch_equal_oracle_funnel"] = same
        parity["max_rel_w_vs_oracle_funnel"] = float(((w_b.cpu() - w_ref).abs() / w_ref.abs()).max()) if same else None
    # the reference's own Sober.next_batch (fixture): the sampled-prior return shape, bit for bit
    try:
        from tests.golden import make_golden as MG
        z = np.load(os.path.join(ROOT, "tests", "golden", "sober_next_batch.npz"))
        c = MG.SOBER_CASES["continuous"]
        model, fspec = MG.sober_model(c)
        model.kernel_spec = sober_amd.KernelSpec(fspec.kind, fspec.lengthscale, fspec.outputscale, fspec.X_obs, fspec.S_cache,
                                                 fspec.noise, fspec.mean_const, fspec.alpha)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            sf = sober_amd.Sober(MG.UniformPrior(c["d"], device=dev), model, kernel_type=c["kernel_type"],
                                 prior_updater=lambda s_, X, w: None)
            torch.manual_seed(c["seed_call"])
            Xf = sf.next_batch(c["n_rec"], c["n_nys"], c["batch"])
        parity["reference_fixture_sober_next_batch_equal"] = bool(np.array_equal(Xf.cpu().numpy(), z["continuous_X"]))
    except Exception as e:                                              # noqa: BLE001
        parity["reference_fixture_sober_next_batch_equal"] = f"not run: {type(e).__name__}: {e}"
    out = {
        "metric": f"acquisition-step candidates/sec (Sober.next_batch: N_rec={N}, N_nys={M}, d={cfg['d']}, batch={b})",
        "value": N / (ms_per_step * 1e-3), "unit": "candidates/s", "n_gpus": 1, "steps": args.steps,
        "warmup": max(args.warmup, 3), "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"Sober.next_batch on a continuous prior, {cfg['name']}, n_rec={N}, n_nys={M}, batch={b}, "
                               f"n_obs={cfg['n_obs']} (BASELINE.json configs[1] shapes; SOBER/_sober.py:125-195)",
                   "parallelism": "single GPU"},
        "phases_ms_per_step": phases_ms, "phase_calls_per_step": {k: v / args.steps for k, v in calls.items()},
        "ms_per_step_with_phase_brackets": ms_bracketed,
        "cpu_baseline": cpu, "parity": parity, "n_selected": int(X_b.shape[0]),
    }
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--event-every", type=int, default=4,
                    help="the level launches of every n-th timed step carry the HIP event pairs behind roofline.achieved "
                         "(an event-carrying dispatch costs the stream 10-20 us: on every step that is +0.1-0.2 ms at "
                         "configuration 2, profiles/r03_event_cost.txt); 1 = every step")
    ap.add_argument("--funnel", action="store_true",
                    help="one GPU: time `Sober.next_batch` (pi weights -> cleansing_weights -> KMeans Nystrom subsample -> "
                         "sampling_recombination) at configuration 2's shapes instead of the recombination step alone")
    ap.add_argument("--no-sweep", action="store_true", help="skip the n_obs sweep of SURVEY.md 8(d) (configuration 2, one GPU)")
    ap.add_argument("--check-unsharded", action="store_true",
                    help="N > 1: rank 0 gathers every shard and repeats the step UNSHARDED on its one GPU (outside the timed "
                         "region); the JSON then carries the comparison (identical indices, weights) -- the test hook of "
                         "tests/test_hip_round4.py, not for pools that do not fit one GPU")
    ap.add_argument("--no-others", action="store_true",
                    help="default line (one GPU, configuration 2): skip the `other_configs` (BASELINE.json configurations 1, 3, 4, 5, "
                         "--other-steps steps each, no CPU leg) and `acquisition_step` (Sober.next_batch) sub-records")
    ap.add_argument("--other-steps", type=int, default=10)
    args = ap.parse_args()
    if args.funnel:
        print(json.dumps(funnel(args)))
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, (f"--gpus {args.gpus} but WORLD_SIZE={world}: a launcher set another world size (without a "
                                "launcher bench.py starts its own torch.distributed.run child: _self_launch)")
    assert torch.cuda.is_available(), "bench.py needs the MI355X (no CPU fallback)"
    # one rank per GPU over RCCL.  (Test hook: SOBER_BENCH_BACKEND=gloo lets several ranks share the GPUs that
    # exist -- the N > 1 code path on a one-GPU box; never used for a measurement.)
    backend = os.environ.get("SOBER_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    group = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        group = dist.group.WORLD

    out = measure(args, args.config, rank, world, dev, group, backend)
    if rank != 0:
        return
    # The other BASELINE.json configurations and the acquisition step, timed by the SAME command (one GPU, default
    # configuration; outside the headline's timed region, no CPU leg): what used to exist only as builder-run files
    # under profiles/.  Every sub-record carries ms_per_step, the dominant kernel's roofline fraction and a parity verdict.
    if world == 1 and args.config == 2 and not args.no_others:
        light = argparse.Namespace(**dict(vars(args), steps=args.other_steps, warmup=2, no_cpu_baseline=True, no_sweep=True,
                                          light=True))
        others = {}
        for c in (1, 3, 4, 5):
            t_c = time.perf_counter()
            try:
                torch.cuda.empty_cache()
                r = measure(light, c, 0, 1, dev, None, backend)
                others[str(c)] = {
                    "workload": r["config"]["workload"], "steps": r["steps"], "warmup_effective": r["warmup_effective"],
                    "ms_per_step": r["ms_per_step"], "ms_per_step_median": r["ms_per_step_median"], "value": r["value"],
                    "unit": r["unit"], "dtype": r["dtype"],
                    "roofline": {k: r["roofline"].get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac",
                                                                   "step_frac", "kernel_ms_per_step", "launches_per_step",
                                                                   "traffic", "F_alg_per_step", "F_executed_per_step")},
                    "parity": r["parity"], "ms_per_step_fresh_pool": r["ms_per_step_fresh_pool"],
                    "phases_ms_per_step": r["phases_ms_per_step"], "n_selected": r["n_selected"],
                    "wall_s": round(time.perf_counter() - t_c, 2)}
            except Exception as e:                                      # noqa: BLE001  (a sub-record never takes the headline down)
                others[str(c)] = {"error": f"{type(e).__name__}: {e}", "wall_s": round(time.perf_counter() - t_c, 2)}
        out["other_configs"] = others
        # ... and one configuration beyond BASELINE.json's batches (config 7: batch 250, every Caratheodory step on the
        # memory-resident kernels), same protocol, three steps
        t_c = time.perf_counter()
        try:
            torch.cuda.empty_cache()
            r = measure(argparse.Namespace(**dict(vars(light), steps=3)), 7, 0, 1, dev, None, backend)
            out["beyond_baseline"] = {"7": {
                "workload": r["config"]["workload"], "steps": r["steps"], "ms_per_step": r["ms_per_step"], "value": r["value"],
                "unit": r["unit"], "dtype": r["dtype"], "parity": r["parity"], "phases_ms_per_step": r["phases_ms_per_step"],
                "n_selected": r["n_selected"], "wall_s": round(time.perf_counter() - t_c, 2)}}
        except Exception as e:                                          # noqa: BLE001
            out["beyond_baseline"] = {"7": {"error": f"{type(e).__name__}: {e}", "wall_s": round(time.perf_counter() - t_c, 2)}}
        t_c = time.perf_counter()
        try:
            torch.cuda.empty_cache()
            f = funnel(argparse.Namespace(**dict(vars(args), steps=5, warmup=3, no_cpu_baseline=True)))
            out["acquisition_step"] = {k: f[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step",
                                                         "phases_ms_per_step", "phase_calls_per_step", "parity", "n_selected")}
            out["acquisition_step"]["workload"] = f["config"]["workload"]
            out["acquisition_step"]["wall_s"] = round(time.perf_counter() - t_c, 2)
        except Exception as e:                                          # noqa: BLE001
            out["acquisition_step"] = {"error": f"{type(e).__name__}: {e}"}
    print(json.dumps(out))


def measure(args, config, rank, world, dev, group, backend):
    """One configuration, timed as the contract says (warm-up, barrier + synchronise either side of exactly --steps steps,
    max over ranks) -> the JSON record on rank 0, None elsewhere.  `args.light` (the sub-records of the default line): no
    cold-start protocol, no KMeans line, no CPU leg."""
    cfg = CONFIGS[config]
    light = bool(getattr(args, "light", False))
    X_cand, X_nys, mu0, spec, inp, N_loc = build_inputs(cfg, rank, world, dev)
    ks = sober_amd.KernelSpec(spec.kind, spec.lengthscale, spec.outputscale, spec.X_obs, spec.S_cache,
                              spec.noise, spec.mean_const, spec.alpha)
    kernel = sober_amd.Kernel(ks, cfg["mode"])
    sober_amd.setting_parameters(device=dev, dtype=torch.double)
    mu = mu0.clone()
    b = cfg["b"]
    N_total = N_loc * world

    from sober_amd._ops_hip import HipOps
    ops = HipOps(dev)
    timers = {}

    def step(tm=None):
        mu.copy_(mu0)
        # (the path draws from the CPU generator only -- svd_lowrank's randn, like the reference; torch.manual_seed would also
        #  walk the device generators: 19 us of bench bookkeeping per step against 1 us, scripts/bench_overheads.py)
        torch.default_generator.manual_seed(SEED_CALL)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return sober_amd.recombination(X_cand, X_nys, b, kernel, dev, torch.double, init_weights=mu,
                                           group=group, row_offset=rank * N_loc, _ops=ops, _timers=tm)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    # the warm-up steps run exactly what the timed steps run, event brackets included (their first use costs
    # tens of ms in a fresh process)
    ops.prof = []
    ops.prof_reserve(48 * (args.steps + args.warmup + 1))    # event pairs for the level launches, created up front
    idx, w = step()                                      # initialisation (library load, workspaces, first-use paths)
    # the first `--steps` steps of this fresh process, timed like the timed region (ms_per_step_cold: what the same
    # command measured before round 4's last commit put a quarter of a second of steps in front of the warm-ups)
    torch.cuda.synchronize(); barrier()
    t_cold = time.perf_counter()
    n_cold = 2 if light else args.steps
    for _ in range(n_cold):
        idx, w = step()
    torch.cuda.synchronize(); barrier()
    ms_cold = (time.perf_counter() - t_cold) / n_cold * 1e3
    n_init = 1 + n_cold
    # ... and a quarter of a second of the same step: a fresh process on a fresh box starts with the device's clocks and
    # the allocator's pools cold (first line of a box 4.32 ms, the same command again 4.17: ten timed steps are 42 ms).
    # These steps are reported: init_steps, warmup_effective = init_steps + --warmup.
    t_init = time.perf_counter()
    n_spin = 0
    while world == 1 and time.perf_counter() - t_init < (0.1 if light else 0.25) and n_spin < 60:
        idx, w = step(); n_spin += 1
    for _ in range(20 if world > 1 else 0):              # (a fixed count where the steps are collective)
        idx, w = step(); n_spin += 1
    n_init += n_spin
    import gc
    gc.collect(); gc.disable()                           # no collector pauses inside the timed region
    for _ in range(args.warmup):
        idx, w = step()
    torch.cuda.synchronize()
    ops.prof = []
    barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    per_step = []
    ev_every = max(1, int(os.environ.get("SOBER_BENCH_EVENTS_EVERY", args.event_every)))
    prof_all = ops.prof
    ev_steps = 0
    for i_step in range(args.steps):
        with_events = i_step % ev_every == 0                # (every n-th step of the timed region: --event-every)
        ops.prof = prof_all if with_events else None
        ev_steps += 1 if with_events else 0
        ts = time.perf_counter()
        idx, w = step(timers)
        per_step.append(time.perf_counter() - ts)          # (host clock at return: the result tensors are final)
    torch.cuda.synchronize(); barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # dominant kernel: the durations of ALL its launches in the timed region (main and leftover launch of every level).
    # The event pairs ride in the dispatches (hipExtLaunchKernelGGL, sober_set_launch_events): elapsed_time is the
    # kernel's own begin -> end, the figure rocprofv3 --kernel-trace reports; no marker packet enters the stream and
    # nothing is subtracted.  kernel_ms_per_step = calls x AverageNs / steps of profiles/r03_cfgN_kernel_stats.csv.
    prof, ops.prof = prof_all, None
    kern_ms = sum(a.elapsed_time(b_) for a, b_, _, _ in prof)
    entries = sum(e for _, _, e, _ in prof)
    n_launches = sum(n for _, _, _, n in prof)

    # the step together with the Nystrom subsample that precedes it in the reference's funnel for a continuous
    # prior (kmeans_resampling, SOBER/_weights.py:95-126, SOBER/_sampler.py:316-320); outside the timed region
    kmeans_ms = None
    if world == 1 and cfg["kind"] != "tanimoto" and not light:
        sober_amd.KMeans(X_cand, cfg["M"])
        torch.cuda.synchronize()
        k0 = time.perf_counter()
        for _ in range(3):
            sober_amd.KMeans(X_cand, cfg["M"])
        torch.cuda.synchronize()
        kmeans_ms = (time.perf_counter() - k0) / 3 * 1e3

    # fingerprint pools: the step as a default `Sober` sees it.  The timed region above hits the caches a dataset prior
    # WITHOUT pruning allows (the same pool tensor every iteration: its packed words and the pool's posterior mean are
    # kept); with `dataset_pruning=True` -- the reference's default, SOBER/_sober.py:10-39 -- X_cand = available[idx] is a
    # fresh tensor per iteration: re-pack + posterior mean over the pool every time.  Both are reported.
    fresh_ms = None
    if cfg["kind"] == "tanimoto":
        each = []
        for _ in range(6):
            ops.clear_cache()                                 # (what a fresh pool tensor amounts to, without a 4 GB copy)
            barrier(); torch.cuda.synchronize()
            c0 = time.perf_counter()
            step()
            torch.cuda.synchronize()
            each.append(time.perf_counter() - c0)
        if world > 1:
            import torch.distributed as dist
            tt = torch.tensor([float(np.median(each[1:]))], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            fresh_ms = float(tt.item()) * 1e3
        else:
            fresh_ms = float(np.median(each[1:])) * 1e3

    # N > 1, on request: the same step unsharded on rank 0's GPU, from the gathered shards
    sharded_check = None
    if world > 1 and args.check_unsharded:
        import torch.distributed as dist
        on_dev = backend == "nccl"
        Xs, ms = (X_cand, mu0) if on_dev else (X_cand.cpu(), mu0.cpu())
        Xg = [torch.empty_like(Xs) for _ in range(world)]
        mg = [torch.empty_like(ms) for _ in range(world)]
        dist.all_gather(Xg, Xs.contiguous())
        dist.all_gather(mg, ms.contiguous())
        ig = [torch.empty_like(idx if on_dev else idx.cpu()) for _ in range(world)]     # (every rank holds the global result)
        dist.all_gather(ig, (idx if on_dev else idx.cpu()).contiguous())
        if rank == 0:
            X_all, mu_all = torch.cat(Xg).to(dev), torch.cat(mg).to(dev)
            torch.manual_seed(SEED_CALL)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                idx1, w1 = sober_amd.recombination(X_all, X_nys, b, kernel, dev, torch.double, init_weights=mu_all, _ops=HipOps(dev))
            same = bool(torch.equal(idx1.cpu(), idx.cpu()))
            sharded_check = {"idx_equal_unsharded": same, "ranks_agree": all(bool(torch.equal(g_.cpu(), idx.cpu())) for g_ in ig),
                             "max_rel_w_vs_unsharded": float(((w1 - w).abs() / w1.abs()).max()) if same else None}
            del X_all, mu_all

    if rank != 0:
        return None

    # parity of the timed configuration against the reference golden (N=1, configurations with a full-size golden)
    parity = None
    gpath = os.path.join(ROOT, "tests", "golden", cfg.get("golden", "-"))
    if world == 1 and os.path.exists(gpath):
        z = np.load(gpath)
        same = bool(np.array_equal(idx.cpu().numpy(), z["idx"]))
        relw = float(np.max(np.abs(w.cpu().numpy() - z["w"]) / np.abs(z["w"]))) if same else None
        parity = {"idx_equal_reference": same, "max_rel_w_vs_reference": relw}

    if world == 1 and parity is None:
        # no full-size reference in bench time: the result's invariants and run-to-run bit equality
        ii, ww = idx.cpu().numpy(), w.cpu().numpy()
        mu_after = mu.clone()
        idx2, w2 = step()
        dense = torch.zeros_like(mu0)
        dense[idx] = w
        parity = {"by": "invariants", "n_selected_le_batch": bool(0 < len(ii) <= b), "weights_positive": bool((ww > 0).all()),
                  "mass_error": float(abs(ww.sum() - float(mu0.sum()))), "indices_ascending_in_pool": bool(
                      (np.diff(ii) > 0).all() and ii.min() >= 0 and ii.max() < N_loc),
                  "init_weights_hold_the_result": bool(torch.equal(mu_after, dense)),
                  "bit_equal_run_to_run": bool(torch.equal(idx2, idx) and torch.equal(w2, w))}
    cpu_baseline = None
    if world == 1 and not args.no_cpu_baseline:
        cpu_baseline = cpu_baseline_for(cfg, spec, inp, (X_cand, X_nys, mu0), args.cpu_steps)

    ms_per_step = elapsed / args.steps * 1e3
    traffic, traffic_src = pmc_traffic(config) if world == 1 else (None, None)
    n_rows = cfg["M"] + (cfg["n_obs"] if cfg["mode"] != "kernel" else 0)
    V = entries / n_rows / ev_steps                      # list positions the level kernel EVALUATED per step
    # the levels of the step (live positions each) and how many of them took their set sums from level 0's class sums
    # instead of a kernel launch (csrc/level_class.hip): the algorithmic figures of SURVEY.md 8(d) count every level
    lv = getattr(ops, "last_levels", None) or {"R": [], "derived": 0}
    V_alg = float(sum(lv["R"])) if lv["R"] and world == 1 else V
    n_levels = float(len(lv["R"])) if lv["R"] and world == 1 else \
        sum(1 for _, _, e, _ in prof if e > 0 and e >= 2 * b * n_rows) / ev_steps
    S2, n1 = 2 * b, b - 1
    common = {"launches": n_launches, "launches_per_step": n_launches / ev_steps, "steps_with_events": ev_steps,
              "kernel_ms_per_step": kern_ms / ev_steps,
              "timing": "HIP events carried in the kernels' dispatches (their own begin/end timestamps), all launches of "
                        f"the kernel in every {ev_every}-th step of the timed region ({ev_steps} of {args.steps} steps), "
                        "nothing subtracted",
              "traffic": traffic, "traffic_unit": "HBM bytes per launch, mean over the kernel's launches of a step",
              "traffic_source": traffic_src, "V_per_step": V, "V_alg_per_step": V_alg, "entries_per_step": entries / ev_steps,
              "levels_per_step": n_levels, "levels_derived_from_class_sums": int(lv["derived"]) if world == 1 else None}
    if cfg["kind"] == "tanimoto":
        # popcount(x & y) as an FP4 (E2M1: 0.0 / 1.0) GEMM on the matrix cores (csrc/level_reduce_tani.hip): 2 * bits operations
        # per (row, candidate) pair, against the dense FP4 MFMA peak (4 x the bf16 rate, MI355X_MICROARCH.md).  (Rounds 2-3
        # ran it as an INT8 GEMM against the 5 P INT8 peak: this round's numbers are NOT comparable as fractions -- compare
        # roofline.achieved / kernel_ms_per_step.)
        bits = 64 * ((cfg["d"] + 63) // 64)
        ops_step = entries / ev_steps * 2 * bits
        tops = entries * 2 * bits / (kern_ms * 1e-3) / 1e12 if kern_ms > 0 else 0.0
        alg_bytes = V_alg * (bits / 8 + 16) + n_levels * (n_rows * S2 + S2) * 8
        roofline = dict(common, bound="mfma", achieved=tops, peak=FP4_PEAK_TOPS, unit="TFLOP/s", frac=tops / FP4_PEAK_TOPS,
                        step_frac=ops_step / (ms_per_step * 1e-3) / 1e12 / FP4_PEAK_TOPS, kernel="k_level_reduce_tani",
                        F_executed_per_step=ops_step, F_alg_per_step=V_alg * n_rows * 2 * bits,
                        peak_int8=5000.0, frac_of_int8_peak=tops / 5000.0,
                        note="bit-packed fingerprints, every bit expanded to an E2M1 nibble (1.0 / 0.0) in LDS, "
                             "v_mfma_scale_f32_16x16x128_f8f6f4 with unit scales: exact popcounts in FP32 (bit-pair operations "
                             "counted in the TFLOP/s unit, 2 per multiply-add); 256 rows per workgroup, so the 700-row table "
                             "is three row blocks (the INT8 form of rounds 2-3: six, at half the matrix rate)")
    else:
        flop_per_entry = 2 * cfg["d"] + 2 + CK[cfg["kind"]]
        achieved = entries * flop_per_entry / (kern_ms * 1e-3) / 1e12 if kern_ms > 0 else 0.0
        # F_alg = entries (2d + 2 + C_k) + L (2 M n_obs S + 2 n M S)
        f_gemm = n_levels * (2.0 * cfg["M"] * cfg["n_obs"] * S2 + 2.0 * n1 * cfg["M"] * S2)
        f_alg = V_alg * n_rows * flop_per_entry + f_gemm            # every level's kernel entries, as the reference evaluates them
        f_exe = entries / ev_steps * flop_per_entry + f_gemm        # what this step evaluated (derived levels: a gather and a scale)
        alg_bytes = V_alg * (8 * cfg["d"] + 16) + n_levels * (n_rows * S2 + S2) * 8
        roofline = dict(common, bound="mfma", achieved=achieved, peak=FP64_PEAK_TFLOPS, unit="TFLOP/s",
                        frac=achieved / FP64_PEAK_TFLOPS, peak_measured=FP64_MEASURED_TFLOPS,
                        frac_of_measured_mix=achieved / FP64_MEASURED_TFLOPS["level-kernel mix (12 MFMA + 160 FMA)"],
                        F_alg_per_step=f_alg, F_executed_per_step=f_exe,
                        step_frac=f_exe / (ms_per_step * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                        step_frac_alg=f_alg / (ms_per_step * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                        kernel="k_level_reduce_wave",
                        note="FP64 compute-bound: -|x-y|^2/2 on v_mfma_f64_16x16x4 (augmented GEMM), "
                             "table-driven FP64 exp on the VALU; MI355X FP64 vector and matrix peaks are "
                             "both 78.6 TFLOP/s at 2.4 GHz and share the vector ALU: on gfx950 no vector "
                             "instruction of a SIMD overlaps with a running v_mfma_f64, and the clock under "
                             "FP64 load is ~1.9 GHz (peak_measured) -- the kernel is bound by its instruction count; "
                             f"achieved / frac: EXECUTED entries * (2d+2+C_k) = entries * {flop_per_entry} over the kernel's own time "
                             "(levels whose set sums are gathered and scaled from level 0's element-class sums are not counted: "
                             "F_executed_per_step; F_alg_per_step counts every level like SURVEY.md 8d); all launches of "
                             "the kernel are counted (level 0 alone runs ~1.35x the average; the deep levels are "
                             "launch-bound); step_frac: the step is a chain of ~200 dependent Caratheodory "
                             "reflector/pivot steps per level, not a throughput problem")
    hbm = {"alg_bytes_per_step": alg_bytes, "gbs_over_step": alg_bytes / (ms_per_step * 1e-3) / 1e9,
           "gbs_over_kernel": alg_bytes / (kern_ms / ev_steps * 1e-3) / 1e9 if kern_ms > 0 else None,
           "peak_gbs": HBM_PEAK_GBS, "frac_over_step": alg_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "note": "B_alg = V (8d + 16) + L (M_tot S + S) 8 (SURVEY.md 8d); a small fraction by construction: the path "
                   "is FP64-compute and latency bound at ~300 flop/B"}

    # n_obs sweep at the headline configuration (one GPU; outside the timed region)
    sweep = None
    if world == 1 and config == 2 and not args.no_sweep:
        sweep = {}
        for n_obs in (100, 200, 800, 1600):
            c2 = dict(cfg, n_obs=n_obs)
            Xc2, Xn2, mu02, spec2, _, _ = build_inputs(c2, 0, 1, dev)
            k2 = sober_amd.Kernel(sober_amd.KernelSpec(spec2.kind, spec2.lengthscale, spec2.outputscale, spec2.X_obs,
                                                       spec2.S_cache, spec2.noise, spec2.mean_const, spec2.alpha), cfg["mode"])
            m2 = mu02.clone()
            ts = []
            for it in range(8):
                m2.copy_(mu02)
                torch.manual_seed(SEED_CALL)
                torch.cuda.synchronize()
                c0 = time.perf_counter()
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    sober_amd.recombination(Xc2, Xn2, b, k2, dev, torch.double, init_weights=m2, _ops=ops)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - c0) * 1e3)
            sweep[str(n_obs)] = float(np.median(ts[3:]))
            del Xc2, Xn2, mu02, m2, k2
    strong = bool(cfg.get("strong"))
    n_rec = str(cfg["N"]) if (strong or world == 1) else "%dx%d" % (world, cfg["N"])
    out = {
        "metric": f"recombination-step candidates/sec (N_rec={n_rec}, N_nys={cfg['M']}, d={cfg['d']}, batch={cfg['b']})",
        "value": N_total / (elapsed / args.steps),
        "unit": "candidates/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "init_steps": n_init, "warmup_effective": n_init + args.warmup,
        "ms_per_step": ms_per_step,
        "ms_per_step_cold": ms_cold,
        "ms_per_step_protocol": f"untimed: 1 initialisation step, {n_cold} steps timed as ms_per_step_cold (the first steps of "
                                f"a fresh process), {n_spin} steady-state steps (0.25 s), {args.warmup} warm-up steps; then the "
                                f"{args.steps} timed steps behind ms_per_step / value",
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": "u64" if cfg["kind"] == "tanimoto" else "f64", "data": "synthetic",
        "config": {"workload": f"{cfg['name']}, N_rec={N_loc} per GPU ({N_total} in all), N_nys={cfg['M']}, "
                               f"batch={cfg['b']}, n_obs={cfg['n_obs']} (BASELINE.json configs[{config - 1}])",
                   "parallelism": f"pool row-sharded x{world}, one all-reduce of (n*S+S) f64 per level ({allreduce_route()})"
                                  if world > 1 else "single GPU"},
        "roofline": roofline,
        "hbm": hbm,
        "cpu_baseline": cpu_baseline,
        "n_obs_sweep_ms_per_step": sweep,
        "parity": parity,
        "parity_sharded_vs_unsharded": sharded_check,
        "ms_per_step_fresh_pool": fresh_ms,
        "ms_per_step_note": None if fresh_ms is None else
        "ms_per_step / value: the pool tensor comes back unchanged (dataset prior without pruning: packed words and the pool's "
        "posterior mean are kept); ms_per_step_fresh_pool: a fresh pool tensor per call (dataset_pruning=True, the reference's "
        "default): re-pack + posterior mean every step",
        "kmeans_nystrom_subsample_ms": kmeans_ms,
        "ms_per_step_incl_kmeans": None if kmeans_ms is None else ms_per_step + kmeans_ms,
        "phases_ms_per_step": {k: v / args.steps * 1e3 for k, v in timers.items()},
        "ms_each_step": [round(v * 1e3, 3) for v in per_step],
        "ms_per_step_median": round(sorted(per_step)[len(per_step) // 2] * 1e3, 3),    # (the steps without event pairs or host hiccups)
        "n_selected": int(idx.numel()),
    }
    gc.enable()
    return out


if __name__ == "__main__":
    main()
