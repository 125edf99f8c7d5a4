"""The Nystrom phase (make_cov_psd's ladder + svd_lowrank's range finder + projection: HipOps.nystrom_basis_device, timed
with a synchronisation behind it) and the whole step at cfg-2's shape for several N_nys -- 500 (the multi-CU probes), 600 and
1000 (the panel-by-panel probes of round 4)."""
import os, sys, time, warnings
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sober_amd
from sober_amd import _ops_hip
from tests.golden.synth import SEED_CALL, build_spec, synth
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
for M in [int(v) for v in os.environ.get("NYS_M", "500,600,1000").split(",")]:
    cfg = dict(kind="rbf", mode="predictive_covariance", N=100000, M=M, d=10, b=100, n_obs=200, seed=0)
    inp = synth(cfg); spec = build_spec(cfg, inp)
    ks = sober_amd.KernelSpec(spec.kind, spec.lengthscale, spec.outputscale, spec.X_obs, spec.S_cache, spec.noise, spec.mean_const, spec.alpha)
    kernel = sober_amd.Kernel(ks, cfg["mode"])
    sober_amd.setting_parameters(device=dev, dtype=torch.double)
    X_cand, X_nys, mu0 = t(inp["X_cand"]).to(dev), t(inp["X_nys"]).to(dev), t(inp["mu0"]).to(dev)
    ops = _ops_hip.HipOps(dev)
    acc = [0.0, 0]
    f = ops.nystrom_basis_device
    def g(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        a = a[:3]                                              # (no overlap hooks: the phase alone)
        r = f(*a); torch.cuda.synchronize()
        acc[0] += time.perf_counter() - t0; acc[1] += 1
        return r
    ops.nystrom_basis_device = g
    def step():
        mu = mu0.clone(); torch.manual_seed(SEED_CALL)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return sober_amd.recombination(X_cand, X_nys, cfg["b"], kernel, dev, torch.double, init_weights=mu, _ops=ops)
    for _ in range(3): step()
    torch.cuda.synchronize(); acc[0] = 0.0; acc[1] = 0
    K = 10
    t0 = time.perf_counter()
    for _ in range(K): idx, w = step()
    torch.cuda.synchronize()
    print("N_nys %4d: step %.3f ms (the phase synchronised on both sides), Nystrom phase %.3f ms, %d points kept"
          % (M, (time.perf_counter() - t0) / K * 1e3, acc[0] / max(acc[1], 1) * 1e3, idx.numel()))
