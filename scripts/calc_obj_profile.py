"""cProfile of the acquisition-guided recombination (calc_obj) at configuration 2: where the host's time per level goes."""
import cProfile, pstats, os, sys, warnings
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sober_amd
from tests.golden.synth import SEED_CALL, build_spec, synth
warnings.simplefilter("ignore")
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
CFG2 = dict(kind="rbf", mode="predictive_covariance", N=100000, M=500, d=10, b=100, n_obs=200, seed=0)
inp = synth(CFG2); spec = build_spec(CFG2, inp)
ks = sober_amd.KernelSpec(spec.kind, spec.lengthscale, spec.outputscale, spec.X_obs, spec.S_cache, spec.noise, spec.mean_const, spec.alpha)
sober_amd.setting_parameters(device=dev, dtype=torch.double)
Xc = t(inp["X_cand"]).to(dev); Xn = t(inp["X_nys"]).to(dev); mu0 = t(inp["mu0"]).to(dev)
kern = sober_amd.Kernel(ks, CFG2["mode"])
obj = lambda X: (X ** 2).sum(1)
def rec():
    mu = mu0.clone(); torch.manual_seed(SEED_CALL)
    return sober_amd.recombination(Xc, Xn, CFG2["b"], kern, dev, torch.double, init_weights=mu, calc_obj=obj)
for _ in range(3): rec()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(20): rec()
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); rows = []
for (fn, line, name), (cc, nc, tt, ct, callers) in st.stats.items():
    rows.append((ct / 20 * 1e3, tt / 20 * 1e3, nc / 20, os.path.basename(fn), line, name))
rows.sort(reverse=True)
print("cum ms/step  own ms/step  calls/step  where")
for r in rows[:45]:
    print("%9.3f %9.3f %8.1f  %s:%d %s" % r)
