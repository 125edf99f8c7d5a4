"""Timings of the BASELINE.json configurations (and the KMeans Nystrom subsample) on one MI355X.
Not the bench line (bench.py is): this fills the table in DESIGN.md.  Run on the GPU box."""
import json, os, sys, time, warnings
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sober_amd
from tests.golden.synth import synth, build_spec, SEED_CALL
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
CFG = {
    "cfg1": dict(kind="rbf", mode="predictive_covariance", N=2000, M=100, d=2, b=10, n_obs=30, seed=0, ard=True),
    "cfg2": dict(kind="rbf", mode="predictive_covariance", N=100000, M=500, d=10, b=100, n_obs=200, seed=0),
    "cfg3": dict(kind="matern52", mode="predictive_covariance", N=50000, M=500, d=6, b=200, n_obs=200, seed=0),
    "cfg4_1gpu": dict(kind="rbf", mode="predictive_covariance", N=1000000, M=500, d=20, b=100, n_obs=200, seed=0),
    "cfg5": dict(kind="tanimoto", mode="weighted_predictive_covariance", N=250000, M=500, d=2048, b=100, n_obs=200,
                 seed=10, bit_p=0.04, mean_const=0.3),
}
out = {}
ONLY = [a for a in sys.argv[1:] if a in CFG]
for name, case in CFG.items():
    if ONLY and name not in ONLY:
        continue
    if name == "cfg5":                      # 250k x 2048 FP64 = 4 GB on the host: build on the device
        g = torch.Generator(device=dev); g.manual_seed(10)
        X = (torch.rand(case["N"], case["d"], device=dev, generator=g) < 0.04).to(torch.float64)
        small = dict(case, N=case["M"] + 1000)
        inp = synth(small); spec = build_spec(small, inp)
        Xn = X[torch.randperm(case["N"], device=dev, generator=g)[:case["M"]]].clone()
        mu0 = torch.rand(case["N"], device=dev, generator=g, dtype=torch.float64); mu0 /= mu0.sum()
    else:
        inp = synth(case); spec = build_spec(case, inp)
        X, Xn, mu0 = t(inp["X_cand"]).to(dev), t(inp["X_nys"]).to(dev), t(inp["mu0"]).to(dev)
    ks = sober_amd.KernelSpec(spec.kind, spec.lengthscale, spec.outputscale, spec.X_obs, spec.S_cache, spec.noise,
                              spec.mean_const, spec.alpha)
    kern = sober_amd.Kernel(ks, case["mode"])
    mu = mu0.clone()
    def step():
        mu.copy_(mu0); torch.manual_seed(SEED_CALL)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return sober_amd.recombination(X, Xn, case["b"], kern, init_weights=mu)
    step(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); idx, w = step(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    out[name] = dict(ms=float(np.median(ts) * 1e3), cand_per_s=case["N"] / float(np.median(ts)), n_sel=int(idx.numel()),
                     sum_w=float(w.sum()))
    print(name, out[name], flush=True)
    del X
if ONLY:
    sys.exit(0)
# BASQ quadrature (row f4) at the cfg-2 shape: g-space kernel, per-candidate posterior correction, matrix in HBM
case = dict(CFG["cfg2"], mean_const=0.15)
inp = synth(case); spec = build_spec(case, inp)
rng = np.random.default_rng(5)
alpha = (spec.S_cache @ spec.S_cache.T) @ t(np.log(np.exp(-2.0 * ((inp["X_obs"] - 0.5) ** 2).sum(1)) + 1.0))
ks = sober_amd.KernelSpec(spec.kind, spec.lengthscale, spec.outputscale, spec.X_obs, spec.S_cache, spec.noise, 0.15, alpha)
model = sober_amd.ScaleMmlt(ks, beta=-1.0)
X = t(inp["X_cand"]).to(dev)
def qstep():
    torch.manual_seed(SEED_CALL)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return sober_amd.basq_quadrature(X, case["M"], case["b"], model)
qstep(); torch.cuda.synchronize()
ts = []
for _ in range(5):
    t0 = time.perf_counter(); r = qstep(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
torch.cuda.synchronize(); t0 = time.perf_counter()
Kmat = model.gspace_kernel.materialise(X, X[:case["M"]]); torch.cuda.synchronize()
t_mat = time.perf_counter() - t0
out["basq_quadrature_cfg2_shape"] = dict(ms=float(np.median(ts) * 1e3), materialise_ms=t_mat * 1e3, ELML=r[0], AVLML=r[1],
                                         n_sel=int(r[3].numel()))
print("basq", out["basq_quadrature_cfg2_shape"], flush=True)
del X, Kmat
# KMeans Nystrom subsample (SOBER/_weights.py:100-126): reference 45 s at this size on 8 cores
rng = np.random.default_rng(0)
Xk = t(rng.random((100000, 10))).to(dev)
sober_amd.KMeans(Xk, 500); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): cl, c = sober_amd.KMeans(Xk, 500)
torch.cuda.synchronize()
out["kmeans_N100k_K500_d10"] = dict(ms=(time.perf_counter() - t0) / 3 * 1e3)
print("kmeans", out["kmeans_N100k_K500_d10"])
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/config_timings.json", "w"), indent=1)
