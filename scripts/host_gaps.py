"""Host-side wall clock of the pieces of one recombination step at cfg-2 (where the launch-bound time goes)."""
import os, sys, time, warnings, collections
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sober_amd
from sober_amd import _ops_hip, _engine, _native as nat
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
acc = collections.defaultdict(float); cnt = collections.Counter()

def wrap(obj, name, sync=False):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k)
        if sync: torch.cuda.synchronize()
        acc[name] += time.perf_counter() - t0; cnt[name] += 1
        return r
    setattr(obj, name, g)

for nm in ("build_plan", "gram", "set_projection", "nonzero_start", "nonzero_finish", "level_moments", "level_loop", "level_final",
           "level_car", "level_update", "direct_columns", "scatter_weights", "to_host", "nystrom_basis_device"):
    wrap(ops, nm)
wrap(nat, "cholesky_probe"); wrap(torch.linalg, "svd")

def step():
    mu = mu0.clone(); torch.manual_seed(SEED_CALL)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return sober_amd.recombination(X_cand, X_nys, CFG2["b"], kernel, dev, torch.double, init_weights=mu, _ops=ops)
for _ in range(3): step()
torch.cuda.synchronize(); acc.clear(); cnt.clear()
K = 10
t0 = time.perf_counter()
for _ in range(K): step()
torch.cuda.synchronize(); tot = (time.perf_counter() - t0) / K
print("step %.3f ms" % (tot * 1e3))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("%-24s %8.1f us/step  (%d calls/step, %.1f us each)" % (k, v / K * 1e6, cnt[k] // K, v / cnt[k] * 1e6))
