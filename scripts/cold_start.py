"""Per-step wall clock of the first process on a fresh box (cold page cache / clocks): which steps are slow, and where."""
import os, sys, time, warnings
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sober_amd
from sober_amd import _ops_hip
from tests.golden.synth import SEED_CALL, build_spec, synth
CFG2 = dict(kind="rbf", mode="predictive_covariance", N=100000, M=500, d=10, b=100, n_obs=200, seed=0)
dev = torch.device("cuda:0")
inp = synth(CFG2); spec = build_spec(CFG2, inp)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
ks = sober_amd.KernelSpec(spec.kind, spec.lengthscale, spec.outputscale, spec.X_obs, spec.S_cache, spec.noise, spec.mean_const, spec.alpha)
kernel = sober_amd.Kernel(ks, CFG2["mode"])
sober_amd.setting_parameters(device=dev, dtype=torch.double)
X_cand, X_nys, mu0 = t(inp["X_cand"]).to(dev), t(inp["X_nys"]).to(dev), t(inp["mu0"]).to(dev)
ops = _ops_hip.HipOps(dev)
svd_t = []
_svd = torch.linalg.svd
def svd(*a, **k):
    t0 = time.perf_counter(); r = _svd(*a, **k); svd_t.append(time.perf_counter() - t0); return r
torch.linalg.svd = svd
for i in range(30):
    mu = mu0.clone(); torch.manual_seed(SEED_CALL); tm = {}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sober_amd.recombination(X_cand, X_nys, CFG2["b"], kernel, dev, torch.double, init_weights=mu, _ops=ops, _timers=tm)
    torch.cuda.synchronize()
    print("step %2d: %.2f ms  nystrom %.2f levels %.2f  host svd %.2f ms  threads %d" % (i, (time.perf_counter() - t0) * 1e3, tm.get("nystrom_device", 0) * 1e3, tm.get("levels_device", 0) * 1e3, sum(svd_t) * 1e3, torch.get_num_threads()))
    svd_t.clear()
