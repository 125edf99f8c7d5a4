"""SURVEY.md 8(f) rows at BASELINE configuration 2's shapes (N = 100k candidates, d = 10, n_obs = 200, batch 100): device
time against the oracle (the torch-CPU restatement of the reference) on this box's host cores, same inputs.
  f1  pi(x) over the pool: GP mean / variance + LFI weight        (SOBER/_gp.py:212-238, _pi.py:20-38)
  f2  WKDE prior: pdf over the pool, sampling                     (SOBER/_wkde.py:109-145, 221-248)
  f3  recombination with calc_obj (acquisition-guided branch)     (SOBER/_rchq.py:67-69, 79-106)
  f4  BASQ g-space kernel: resident matrix + quadrature           (SOBER/BASQ/_scale_mmlt.py:256-275, _basq.py:59-81)
Usage: python tests/tools/next_rows_time.py   (the oracle is test infrastructure: this script is measurement, like bench.py's
cpu_baseline leg)"""
import os, sys, time, warnings
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sober_amd
from oracle import sober_oracle as O
from tests.golden.synth import SEED_CALL, build_spec, synth

warnings.simplefilter("ignore")
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
CFG2 = dict(kind="rbf", mode="predictive_covariance", N=100000, M=500, d=10, b=100, n_obs=200, seed=0)
inp = synth(CFG2); spec = build_spec(CFG2, inp)
ks = sober_amd.KernelSpec(spec.kind, spec.lengthscale, spec.outputscale, spec.X_obs, spec.S_cache, spec.noise,
                          spec.mean_const, spec.alpha)
sober_amd.setting_parameters(device=dev, dtype=torch.double)
Xc_h = t(inp["X_cand"]); Xc = Xc_h.to(dev); Xn = t(inp["X_nys"]).to(dev); mu0 = t(inp["mu0"]).to(dev)


def gpu_ms(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


def cpu_ms(fn, reps=1):
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    return (time.perf_counter() - t0) / reps * 1e3, out


print("host threads", torch.get_num_threads())
# f1
pi = sober_amd.PI(ks)
g, w = gpu_ms(lambda: pi(Xc))
c, w_ref = cpu_ms(lambda: O.PI(spec)(Xc_h))
big = w_ref > 1e-8 * w_ref.max()          # (below that 0.5 (1 + erf) is cancellation noise in the reference itself)
print("f1 pi(x) over 100k candidates            : device %8.3f ms | oracle %9.1f ms | max abs diff %.1e, max rel diff %.1e over the %d weights above 1e-8 of the largest"
      % (g, c, float((w.cpu() - w_ref).abs().max()), float(((w.cpu() - w_ref).abs() / w_ref.abs())[big].max()), int(big.sum())))
# f2
rng = np.random.default_rng(5)
Xk = rng.random((20000, 10)); Wk = rng.random(20000)
torch.manual_seed(11)
kde = sober_amd.WeightedKernelDensityEstimation(t(Xk), t(Wk), 10, bounds=None, n_kde=4096).to(dev)
g, pdf = gpu_ms(lambda: kde.pdf(Xc))
c, pdf_ref = cpu_ms(lambda: O.wkde_pdf(kde.Xobs.cpu(), kde.weights.cpu(), kde.covariance.cpu(), Xc_h[:20000]))
print("f2 WKDE pdf, 4096 components             : device %8.3f ms per 100k points | oracle %9.1f ms per 20k points | max rel diff %.1e"
      % (g, c, float(((pdf[:20000].cpu() - pdf_ref).abs() / pdf_ref.abs().clamp_min(1e-300)).max())))
g, smp = gpu_ms(lambda: kde.sample(100000))
torch.manual_seed(5)
c, _ = cpu_ms(lambda: O.wkde_sample(kde.Xobs.cpu(), kde.weights.cpu(), kde.covariance.cpu(), 100000, None))
print("f2 WKDE sample, 100k draws               : device %8.3f ms | oracle %9.1f ms" % (g, c))
# f3
obj = lambda X: (X ** 2).sum(1)
kern = sober_amd.Kernel(ks, CFG2["mode"])


def rec(co):
    mu = mu0.clone()
    torch.manual_seed(SEED_CALL)
    return sober_amd.recombination(Xc, Xn, CFG2["b"], kern, dev, torch.double, init_weights=mu, calc_obj=co)


g0, _ = gpu_ms(lambda: rec(None), reps=10)
g1, (idx, ww) = gpu_ms(lambda: rec(obj), reps=10)
print("f3 recombination cfg-2 with calc_obj     : device %8.3f ms (without: %.3f ms), %d points" % (g1, g0, len(idx)))
# f4
z = np.load(os.path.join(ROOT, "tests", "golden", "basq.npz"))
bspec = sober_amd.KernelSpec(str(z["a_kind"]), t(z["a_ls"]), 1.3, t(z["a_X_obs"]), t(z["a_S_cache"]), 1e-3, 0.15, t(z["a_alpha"]))
ospec = O.GPSpec(str(z["a_kind"]), t(z["a_ls"]), 1.3, t(z["a_X_obs"]), t(z["a_S_cache"]), 1e-3, 0.15, t(z["a_alpha"]))
model = sober_amd.ScaleMmlt(bspec, beta=-3.25)
d4 = z["a_X_cand"].shape[1]
Xb_h = t(np.random.default_rng(1).random((50000, d4))); Xb = Xb_h.to(dev)


def basq():
    torch.manual_seed(SEED_CALL)
    return sober_amd.basq_quadrature(Xb, 100, 20, model)


g, out = gpu_ms(basq, reps=5)
torch.manual_seed(SEED_CALL)
c, out_ref = cpu_ms(lambda: O.basq_quadrature(Xb_h, 100, 20, ospec, -3.25))
print("f4 BASQ quadrature, 50k candidates, b=20 : device %8.3f ms | oracle %9.1f ms | same points %s"
      % (g, c, bool(np.array_equal(out[3].cpu().numpy(), out_ref[0].numpy()))))
