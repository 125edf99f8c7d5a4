"""Where the host's own time goes between the native calls of one recombination step at cfg-2 (wall clock of the gaps in
which the GPU has nothing queued: entry -> first launch, and the return path)."""
import os, sys, time, warnings, collections
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sober_amd
from sober_amd import _ops_hip, _native as nat
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
marks = []
def wrap(mod, name):
    f = getattr(mod, name)
    def g(*a, **k):
        marks.append((name + ">", time.perf_counter())); r = f(*a, **k); marks.append((name + "<", time.perf_counter())); return r
    setattr(mod, name, g)
for nm in ("plan_rows", "nystrom_basis", "level_loop", "nonzero_i32", "augment_points", "mt19937_uniform53"):
    if hasattr(nat, nm): wrap(nat, nm)
mu = mu0.clone()
def step():
    mu.copy_(mu0); torch.default_generator.manual_seed(SEED_CALL)
    marks.append(("entry", time.perf_counter()))
    r = sober_amd.recombination(X_cand, X_nys, CFG2["b"], kernel, dev, torch.double, init_weights=mu, _ops=ops)
    marks.append(("return", time.perf_counter()))
    return r
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for _ in range(5): step()
    torch.cuda.synchronize(); marks.clear()
    K = 20
    for _ in range(K): step()
torch.cuda.synchronize()
gaps = collections.defaultdict(float)
prev = None
for name, tt in marks:
    if prev is not None:
        gaps[prev[0] + " -> " + name] += tt - prev[1]
    prev = (name, tt)
for k, v in sorted(gaps.items(), key=lambda kv: -kv[1]):
    print("%-46s %8.1f us/step" % (k, v / K * 1e6))
