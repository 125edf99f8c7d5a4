"""A pool beyond the BASELINE sizes on one GPU: N_rec = 8M x d = 20 (1.3 GB of candidates; 288 GB of HBM are there for it) --
does everything index and size correctly (int32 lists, 64-bit offsets), and what does a step cost?  Checked against the
invariants of the result (batch points, positive weights that sum to the pool's, indices inside the pool, ascending)."""
import os, sys, time, warnings
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sober_amd
from tests.golden.synth import SEED_CALL, build_spec, synth
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
N = int(os.environ.get("BIG_N", "8000000"))
cfg = dict(kind="rbf", mode="predictive_covariance", N=200000, M=500, d=20, b=100, n_obs=200, seed=0)
inp = synth(cfg); spec = build_spec(cfg, inp)
ks = sober_amd.KernelSpec(spec.kind, spec.lengthscale, spec.outputscale, spec.X_obs, spec.S_cache, spec.noise, spec.mean_const, spec.alpha)
kernel = sober_amd.Kernel(ks, cfg["mode"])
sober_amd.setting_parameters(device=dev, dtype=torch.double)
g = torch.Generator(device=dev).manual_seed(1)
X_cand = torch.rand(N, cfg["d"], generator=g, dtype=torch.float64, device=dev)
mu0 = torch.rand(N, generator=g, dtype=torch.float64, device=dev); mu0 /= mu0.sum()
X_nys = X_cand[torch.randperm(N, generator=g, device=dev)[:cfg["M"]]].clone()
def step():
    mu = mu0.clone(); torch.manual_seed(SEED_CALL)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return sober_amd.recombination(X_cand, X_nys, cfg["b"], kernel, dev, torch.double, init_weights=mu)
idx, w = step(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): idx, w = step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 3 * 1e3
i = idx.cpu().numpy(); ww = w.cpu().numpy()
ok = len(i) <= cfg["b"] and len(i) > 0 and (ww > 0).all() and abs(ww.sum() - 1.0) < 1e-9 and i.min() >= 0 and i.max() < N and (np.diff(i) > 0).all()
print("N_rec %d x %d: %.2f ms per step (%.0f M candidates/s), %d points, weights sum %.12f, largest index %d: %s"
      % (N, cfg["d"], ms, N / ms / 1e3, len(i), ww.sum(), i.max(), "ok" if ok else "INVARIANT BROKEN"))
t0 = time.perf_counter()
cl, c = sober_amd.KMeans(X_cand, 500); torch.cuda.synchronize()
t0 = time.perf_counter()
cl, c = sober_amd.KMeans(X_cand, 500); torch.cuda.synchronize()
print("KMeans of the same pool, K = 500: %.2f ms; clusters %d..%d" % ((time.perf_counter() - t0) * 1e3, int(cl.min()), int(cl.max())))
